#!/bin/bash
# whole-schedule / headline / compact throughput for a few (A, B) of the skin rule  skin = clamp(A + B T, 0.07, 0.25)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for ab in "-0.0125 0.000375" "-0.03 0.000375" "-0.05 0.000375" "-0.07 0.0004" "-0.05 0.0004" "0.0 0.00033" "-0.0125 0.000375"; do
  set -- $ab
  AWSEM_SKIN_A=$1 AWSEM_SKIN_B=$2 timeout -k 10 300 python bench.py --no-cpu-baseline > /tmp/b.json 2>/dev/null || exit 3
  python -c "import json; d=json.load(open('/tmp/b.json')); print('A $1 B $2: value %.3fM full %.3fM compact %.3fM' % (d['value']/1e6, d['value_full_schedule']/1e6, d['value_compact_300K']/1e6))"
done
