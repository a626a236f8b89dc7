#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for w in synthetic synthetic-hb-20k; do
for s in 0.16 0.2 0.24 0.28 0.32 0.4; do
AWSEM_SKIN=$s timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline > $O/bench_skin_${w}_$s.json 2> $O/bench_skin_${w}_$s.err || { tail -20 $O/bench_skin_${w}_$s.err; exit 2; }
python -c "import json; d=json.load(open('$O/bench_skin_${w}_$s.json')); print('$w skin $s', d['value'], d['ms_per_step'])"
done
done
