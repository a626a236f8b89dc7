#!/bin/bash
# A/B of an environment toggle on the bench workload, 150 launches each.  Usage (GPU box): bash tools/ab_env.sh VAR=VALUE [bench args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
  for setting in "" "$1"; do
    env $setting timeout -k 10 300 python bench.py --no-cpu-baseline --steps 150 --warmup 5 ${@:2} > /tmp/b.json 2> /tmp/b.err || { cat /tmp/b.err; exit 2; }
    python -c "import json; d=json.load(open('/tmp/b.json')); print('[%s] value %.3fM steps/s  kernel_ms %.4f' % ('$setting' or 'default', d['value']/1e6, d['roofline']['kernel_ms']))"
  done
done
