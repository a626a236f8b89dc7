#!/bin/bash
# The GPU suite under the library's alternative modes (each must pass like the default): bash tools/gpu_modes.sh  ->  gpurun_out/gpu_tests_<mode>.log
cd ${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p gpurun_out
for mode in "AWSEM_STEP_LEAN=0" "AWSEM_STAGES_PER_LAUNCH=16" "AWSEM_QUERY_REUSE=0" "AWSEM_LIST_REUSE=0" "AWSEM_ASYNC_REPORT=0" "AWSEM_FORCE_CELL=1 AWSEM_CELL_FUSED=0" "AWSEM_FORCE_CELL=1 AWSEM_CELL_GRAPH=0" "AWSEM_PRIO=1" "AWSEM_ZERO_COPY_T=0"; do
  tag=$(echo "$mode" | tr ' =' '__')
  env $mode timeout -k 10 600 python -u -m pytest tests -m gpu -q --timeout 300 --timeout-method thread > gpurun_out/gpu_tests_$tag.log 2>&1
  echo "[$mode] $(tail -1 gpurun_out/gpu_tests_$tag.log)"
  grep "^FAILED" gpurun_out/gpu_tests_$tag.log | head -8
done
