#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
export AWSEM_HIP_LIBRARY=$R/scratch/libawsem_PACE.so AWSEM_PLACE_EVERY=16
for prio in 1 3 1 3; do
  AWSEM_PRIO=$prio timeout -k 10 300 python bench.py --no-cpu-baseline > /tmp/b.json 2>/tmp/b.err || { cat /tmp/b.err; exit 2; }
  python -c "import json; d=json.load(open('/tmp/b.json')); print('prio $prio 1r69: value %.3fM full %.3fM compact %.3fM' % (d['value']/1e6, d['value_full_schedule']/1e6, d['value_compact_300K']/1e6))"
done
for w in t0958 2xov 2xov-remd; do for prio in 1 3; do for leg in "" "--folded"; do
  [ "$w" != t0958 ] && [ -n "$leg" ] && continue
  AWSEM_PRIO=$prio timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps 100 --workload $w $leg > /tmp/b.json 2>/tmp/b.err || { cat /tmp/b.err; exit 2; }
  python -c "import json; d=json.load(open('/tmp/b.json')); print('prio $prio $w $leg: value %.3fM' % (d['value']/1e6))"
done; done; done
