#!/bin/bash
# Cell-list path (BASELINE configs[4]): bench line + rocprofv3 kernel stats of the synthetic 20 008-residue workload.
# Usage (GPU box, repo root): bash tools/prof_synth.sh <tag> [workload]   -> gpurun_out/<tag>_bench_<workload>.json, <tag>_<workload>_kernel_stats.csv
set -o pipefail
tag=$1; w=${2:-synthetic}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout -k 10 300 python bench.py --no-cpu-baseline --workload $w > $O/${tag}_bench_$w.json 2> $O/${tag}_bench_$w.err || { tail -20 $O/${tag}_bench_$w.err; exit 2; }
python -c "import json; d=json.load(open('$O/${tag}_bench_$w.json')); print('$w: %.0f steps/s  %.1f us per step (kernel %.1f us per step)' % (d['value'], 1e6/d['value'], d['roofline']['kernel_ms']*1e3/d['config']['md_steps_per_launch']))"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${tag}_$w -- python3 $R/bench.py --no-cpu-baseline --workload $w > $O/prof_${tag}_$w.log 2>&1 || { tail -20 $O/prof_${tag}_$w.log; exit 3; }
cat $O/prof_${tag}_$w/*/*kernel_stats.csv > $O/${tag}_${w}_kernel_stats.csv
python3 - $O/${tag}_${w}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %6s  avg %8.1f us  total %8.2f ms" % (r["Name"].split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
