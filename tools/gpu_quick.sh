#!/bin/bash
# Quick GPU check: parity tests + one bench line (no profiles).  Usage: bash tools/gpu_quick.sh <tag>
set -o pipefail
tag=${1:-q}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $O/gpu_tests_$tag.log 2>&1 || { tail -40 $O/gpu_tests_$tag.log; exit 1; }
tail -3 $O/gpu_tests_$tag.log
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_$tag.json 2> $O/bench_$tag.err || { cat $O/bench_$tag.err; exit 2; }
python -c "import json,sys; d=json.load(open('$O/bench_$tag.json')); print('value %.3fM steps/s  kernel_ms %.4f' % (d['value']/1e6, d['roofline']['kernel_ms']))"
