#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprofv3 kernel stats and the two HBM-traffic PMC passes.
# Usage (from the repo root on the box): bash tools/gpu_round.sh <tag>
set -o pipefail
tag=${1:-run}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $O/gpu_tests_$tag.log 2>&1 || { tail -30 $O/gpu_tests_$tag.log; exit 1; }
tail -3 $O/gpu_tests_$tag.log
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$tag -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/pmc_fetch_$tag.log 2>&1 || { tail -20 $O/pmc_fetch_$tag.log; exit 4; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$tag -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/pmc_write_$tag.log 2>&1 || { tail -20 $O/pmc_write_$tag.log; exit 5; }
python3 $R/tools/pmc_hbm.py $O/pmc_fetch_$tag $O/pmc_write_$tag $O/${tag}_pmc_hbm.json || exit 6
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- python3 $R/bench.py --no-cpu-baseline > $O/prof_$tag.log 2>&1 || { tail -20 $O/prof_$tag.log; exit 3; }
cat $O/prof_$tag/*/*kernel_stats.csv | head -5 | cut -c1-220
cd $R
timeout -k 10 300 python bench.py > $O/bench_$tag.json 2> $O/bench_$tag.err || { cat $O/bench_$tag.err; exit 2; }
cat $O/bench_$tag.json
timeout -k 10 200 python bench.py --workload synthetic > $O/bench_synth_$tag.json 2> $O/bench_synth_$tag.err || { cat $O/bench_synth_$tag.err; exit 7; }
cat $O/bench_synth_$tag.json
echo done
