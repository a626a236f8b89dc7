#!/bin/bash
set -o pipefail
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for w in ${WL:-1r69 1r69-fraglib t0958 2xov}; do
  echo "== $w"
  AWSEM_INFO=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --workload $w --steps ${ST:-20} --warmup 3 2>&1 >/tmp/b.json | grep "step kernel" | head -3
  python -c "import json; d=json.load(open('/tmp/b.json')); print('   value %.3fM steps/s  kernel_ms %.4f' % (d['value']/1e6, d['roofline']['kernel_ms']))"
  AWSEM_STEP_LEAN=0 timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --workload $w --steps ${ST:-20} --warmup 3 2>/dev/null >/tmp/b.json
  python -c "import json; d=json.load(open('/tmp/b.json')); print('   (lean off) value %.3fM steps/s  kernel_ms %.4f' % (d['value']/1e6, d['roofline']['kernel_ms']))"
done
