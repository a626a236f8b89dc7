#!/bin/bash
# pacing (AWSEM_PRIO=3, scratch/libawsem_PACE.so) at several gains against the rotation of the same build
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
run() { # label, prio, gain, leg
  AWSEM_HIP_LIBRARY=$R/scratch/libawsem_PACE.so AWSEM_PRIO=$2 AWSEM_PACE_GAIN=$3 timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps 150 --warmup 5 $4 $AB_ARGS > /tmp/b.json 2> /tmp/b.err || { cat /tmp/b.err; exit 2; }
  python -c "import json; d=json.load(open('/tmp/b.json')); print('[%-14s %-8s] value %.3fM steps/s  kernel_ms %.4f' % ('$1', '${4:-extended}', d['value']/1e6, d['roofline']['kernel_ms']))"
}
for leg in "" "--folded"; do
  run rotation 1 0 $leg || exit 2
  for g in ${GAINS:-5 10 20 40}; do run "pace gain $g" 3 $g $leg || exit 2; done
  run rotation 1 0 $leg || exit 2
done
