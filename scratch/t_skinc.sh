#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
for c in 0.01444 0.0125 0.0135 0.0155 0.01444; do
export AWSEM_SKIN_C=$c
timeout -k 10 300 python bench.py --no-cpu-baseline > $O/bench_skc_$c.json 2> $O/bench_skc_$c.err || { tail -5 $O/bench_skc_$c.err; exit 2; }
python -c "import json; d=json.load(open('$O/bench_skc_$c.json')); print('c $c 1r69 %.2f full %.2f compact %.2f' % (d['value']/1e6, d['value_full_schedule']/1e6, d['value_compact_300K']/1e6))"
for w in 2xov t0958; do
timeout -k 10 300 python bench.py --workload $w --no-cpu-baseline --no-schedule-legs > $O/bench_skc_${w}_$c.json 2> $O/bench_skc_${w}_$c.err || { tail -5 $O/bench_skc_${w}_$c.err; exit 2; }
python -c "import json; d=json.load(open('$O/bench_skc_${w}_$c.json')); print('c $c $w %.3f' % (d['value']/1e6))"
done
done
