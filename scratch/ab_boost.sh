#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for leg in "" "--folded"; do for b in 0 2 4 8 0 4; do
  AWSEM_HIP_LIBRARY=$R/scratch/libawsem_BOOST.so AWSEM_PACE_BOOST=$b timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps 150 --warmup 5 $leg > /tmp/b.json 2> /tmp/b.err || { cat /tmp/b.err; exit 2; }
  python -c "import json; d=json.load(open('/tmp/b.json')); print('[boost $b %-8s] value %.3fM steps/s  kernel_ms %.4f' % ('${leg:-extended}', d['value']/1e6, d['roofline']['kernel_ms']))"
done; done
