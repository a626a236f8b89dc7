#!/bin/bash
# Usage (here, no GPU needed): bash tools/build_variant.sh <name> [extra hipcc flags, e.g. -DAWSEM_PROF] -> .variants/libawsem_<name>.so
set -e
mkdir -p $(cd $(dirname $0)/.. && pwd)/.variants   # travels to the GPU box (scratch/*.so does not: .gpurunignore); delete variants when done
R=$(cd $(dirname $0)/.. && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -instcombine-max-copied-from-constant-users=100000 \
    "${@:2}" -o $R/.variants/libawsem_$1.so $R/openawsem_amd/csrc/awsem_hip.hip
echo .variants/libawsem_$1.so
