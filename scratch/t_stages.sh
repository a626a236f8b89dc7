#!/bin/bash
# value against the stages per launch (50 steps each), both states
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for leg in "--folded" ""; do
for st in 1 2 4 8; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps $((160/st)) --warmup 5 --stages $st $leg > /tmp/b.json 2>/tmp/b.err || { cat /tmp/b.err; exit 2; }
  python -c "import json; d=json.load(open('/tmp/b.json')); print('[stages $st ${leg:-extended}] value %.3fM steps/s  kernel_ms %.4f' % (d['value']/1e6, d['roofline']['kernel_ms']))"
done
done
