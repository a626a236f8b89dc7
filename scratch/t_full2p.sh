#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
export AWSEM_STEP_LEAN=0
for lib in "" $R/.variants/libawsem_FULL2P.so "" $R/.variants/libawsem_FULL2P.so; do
if [ -n "$lib" ]; then export AWSEM_HIP_LIBRARY=$lib; else unset AWSEM_HIP_LIBRARY; fi
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_f2p.json 2> $O/bench_f2p.err || { tail -5 $O/bench_f2p.err; exit 2; }
python -c "import json; d=json.load(open('$O/bench_f2p.json')); print('lib [$lib] full kernel: 1r69 %.2f full %.2f compact %.2f' % (d['value']/1e6, d['value_full_schedule']/1e6, d['value_compact_300K']/1e6))"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --precision fp64 > $O/bench_f2p64.json 2> $O/bench_f2p64.err || { tail -5 $O/bench_f2p64.err; exit 2; }
python -c "import json; d=json.load(open('$O/bench_f2p64.json')); print('   fp64 %.2f' % (d['value']/1e6))"
done
if [ -n "$AWSEM_HIP_LIBRARY" ]; then timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread 2>&1 | tail -2; fi
