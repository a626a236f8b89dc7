#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
for f in 1 0 1 0; do
AWSEM_REMD_FUSE=$f timeout -k 10 200 python bench.py --workload 2xov-remd --no-cpu-baseline > $O/bench_remd_f$f.json 2> $O/bench_remd_f$f.err || { tail -20 $O/bench_remd_f$f.err; exit 2; }
python -c "import json; d=json.load(open('$O/bench_remd_f$f.json')); print('fuse $f', d['value'], d.get('remd_exchange_us'))"
done
