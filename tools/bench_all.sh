#!/bin/bash
# Every bench line the DESIGN tables cite, one after the other.  Usage (GPU box): bash tools/bench_all.sh <tag>  ->  gpurun_out/<tag>_bench*.json
set -o pipefail
tag=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 400 python bench.py > $O/${tag}_bench.json 2> $O/${tag}_bench.err || { cat $O/${tag}_bench.err; exit 2; }
python scratch/show_bench.py $O/${tag}_bench.json
for w in 1r69-fraglib t0958 2xov 2xov-remd synthetic synthetic-hb synthetic-hb-20k; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w > $O/${tag}_bench_$w.json 2> $O/${tag}_bench_$w.err || { cat $O/${tag}_bench_$w.err; exit 3; }
  python -c "import json; d=json.load(open('$O/${tag}_bench_$w.json')); print('$w: value %.4g %s  kernel_ms %.4f' % (d['value'], d['unit'], d['roofline']['kernel_ms']), d.get('remd_exchange_us', ''))"
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --precision fp64 > $O/${tag}_bench_fp64.json 2> $O/${tag}_bench_fp64.err || { cat $O/${tag}_bench_fp64.err; exit 4; }
python -c "import json; d=json.load(open('$O/${tag}_bench_fp64.json')); print('fp64: value %.4g' % d['value'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps 150 --warmup 5 > $O/${tag}_bench_150.json 2>/dev/null && python -c "import json; d=json.load(open('$O/${tag}_bench_150.json')); print('150 launches: value %.4g' % d['value'])"
timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps 150 --warmup 5 --folded > $O/${tag}_bench_150_folded.json 2>/dev/null && python -c "import json; d=json.load(open('$O/${tag}_bench_150_folded.json')); print('150 launches, compact: value %.4g' % d['value'])"
