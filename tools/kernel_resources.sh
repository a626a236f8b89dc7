#!/bin/bash
# Prints VGPRs / SGPR spills / scratch / occupancy of the kernels named by the pattern (default: the step kernels) from the build remarks.
pat=${1:-awsem_step_kernel}
cd "$(dirname "$0")/../openawsem_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Rpass-analysis=kernel-resource-usage \
  -mllvm -instcombine-max-copied-from-constant-users=100000 ${EXTRA} -o /tmp/libawsem_res.so awsem_hip.hip 2>&1 |
  awk -v pat="$pat" '/Function Name:/ {keep = ($0 ~ pat)} keep && /(Function Name|VGPRs:|Spill|ScratchSize|Occupancy|LDS Size)/ {sub(/.*remark: [^ ]+ /, ""); print}'
