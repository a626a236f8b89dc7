#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
timeout -k 10 300 python -u -m pytest tests/test_gpu_remd_device.py -x -q --timeout 120 --timeout-method thread > $O/remd_fused_tests.log 2>&1 || { tail -40 $O/remd_fused_tests.log; exit 1; }
tail -3 $O/remd_fused_tests.log
timeout -k 10 300 python bench.py --workload 2xov-remd --no-cpu-baseline > $O/bench_remd_fused.json 2> $O/bench_remd_fused.err || { tail -20 $O/bench_remd_fused.err; exit 2; }
python -c "import json; d=json.load(open('$O/bench_remd_fused.json')); print(d['value'], d.get('remd_exchange_us'), d.get('config'))"
bash tools/gpu_quick.sh remdlazy
