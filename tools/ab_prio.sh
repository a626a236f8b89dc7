#!/bin/bash
# A/B of the step kernel's priority-rotation modes (AWSEM_PRIO) on the bench workload: throughput from the product library (150 launches, both legs)
# and workgroup residency from the -DAWSEM_PROF build.  Usage (GPU box): bash tools/ab_prio.sh 0 1 2 ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for mode in "$@"; do
  for leg in "" "--folded"; do
    AWSEM_PRIO=$mode timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps 150 --warmup 5 $leg > /tmp/b.json 2> /tmp/b.err || { cat /tmp/b.err; exit 2; }
    python -c "import json; d=json.load(open('/tmp/b.json')); print('[prio $mode %-8s] value %.3fM steps/s  kernel_ms %.4f' % ('${leg:-extended}', d['value']/1e6, d['roofline']['kernel_ms']))"
    AWSEM_PRIO=$mode AWSEM_HIP_LIBRARY=$R/scratch/libawsem_PROF.so AWSEM_PROF_DUMP=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps 20 $leg 2>&1 | grep "last launch" | cut -c1-400 || exit 3
  done
done
