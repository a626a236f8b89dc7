#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; cd $R
bash tools/gpu_quick.sh v6 || exit 1
bash tools/bench_all.sh r03_v6 || exit 2
bash tools/profile_workload.sh r03_v6 1024 50 || exit 3
bash tools/profile_workload.sh r03_v6_folded 1024 50 --folded || exit 4
bash tools/profile_workload.sh r03_v6_2xov 256 50 --workload 2xov || exit 5
