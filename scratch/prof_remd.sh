set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_remd -- python3 $R/bench.py --no-cpu-baseline --workload 2xov-remd --steps 40 --warmup 5 > $O/prof_remd.log 2>&1 || { tail -20 $O/prof_remd.log; exit 3; }
cat $O/prof_remd/*/*kernel_stats.csv | cut -c1-200
tail -2 $O/prof_remd.log | cut -c1-600
