#!/bin/bash
# value against the launch length (steps per launch), both states
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for leg in "--folded" ""; do
for ms in 50 100 200 400; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps $((7500/ms)) --warmup 5 --md-steps $ms $leg > /tmp/b.json 2>/tmp/b.err || { cat /tmp/b.err; exit 2; }
  python -c "import json; d=json.load(open('/tmp/b.json')); print('[md-steps $ms ${leg:-extended}] value %.3fM steps/s  kernel_ms %.4f' % (d['value']/1e6, d['roofline']['kernel_ms']))"
done
done
