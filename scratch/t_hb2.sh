#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
bash tools/gpu_quick.sh fm3 || exit 1
for w in 2xov t0958; do
timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline > $O/bench_fm3_$w.json 2> $O/bench_fm3_$w.err || { tail -20 $O/bench_fm3_$w.err; exit 2; }
python -c "import json; d=json.load(open('$O/bench_fm3_$w.json')); print('$w', d['value'], d['roofline']['kernel_ms'])"
done
python -c "import json; d=json.load(open('$O/bench_fm3.json')); print('1r69', d['value'], d.get('value_full_schedule'), d.get('value_compact_300K'))"
