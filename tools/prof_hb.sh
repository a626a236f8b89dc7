cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_hb -- python3 $GRAFT_REPO_ROOT/bench.py --workload synthetic-hb --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_hb.log 2>&1 || { tail -20 $GRAFT_REPO_ROOT/gpurun_out/prof_hb.log; exit 3; }
cat $GRAFT_REPO_ROOT/gpurun_out/prof_hb/*/*kernel_stats.csv | cut -d, -f1-5 | cut -c1-150 | head -40
