#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
for w in 1r69 2xov t0958; do
echo "== $w"; AWSEM_INFO=1 AWSEM_STEP_LEAN=0 timeout -k 10 200 python bench.py --workload $w --no-cpu-baseline --no-schedule-legs --steps 2 --warmup 1 2>&1 | grep "step kernel\|threads per replica" | sort | uniq | cut -c1-400
done
