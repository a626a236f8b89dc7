#!/bin/bash
# Bench (150 launches) for each value of an environment variable.  Usage (GPU box): bash tools/sweep_env.sh VAR v1 v2 ... [-- bench args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
var=$1; shift
vals=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done; [ "$1" = "--" ] && shift
for v in "${vals[@]}"; do
  env $var=$v timeout -k 10 300 python bench.py --no-cpu-baseline --steps ${AB_STEPS:-150} --warmup 5 "$@" > /tmp/b.json 2> /tmp/b.err || { cat /tmp/b.err; exit 2; }
  python -c "import json; d=json.load(open('/tmp/b.json')); print('[$var=$v] value %.3fM steps/s  kernel_ms %.4f' % (d['value']/1e6, d['roofline']['kernel_ms']))"
done
