#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
bash tools/gpu_quick.sh ex2 || exit 1
for w in 2xov t0958; do
timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_ex2_$w.json 2> $O/bench_ex2_$w.err || { tail -20 $O/bench_ex2_$w.err; exit 2; }
python -c "import json; d=json.load(open('$O/bench_ex2_$w.json')); print('$w', d['value'], d['roofline']['kernel_ms'])"
done
python -c "import json; d=json.load(open('$O/bench_ex2.json')); print('1r69', d['value'], d.get('value_full_schedule'), d.get('value_compact_300K'))"
