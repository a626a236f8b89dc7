#!/bin/bash
# Phase-cycle profile of the step kernel (-DAWSEM_PROF build in .variants/libawsem_PROF.so, see tools/build_variant.sh); GPU box only.
# The product library is never touched: the loader takes the experiment build from AWSEM_HIP_LIBRARY.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
AWSEM_HIP_LIBRARY=$R/.variants/libawsem_PROF.so AWSEM_PROF_DUMP=1 timeout -k 10 300 python bench.py --no-cpu-baseline ${@:1} 2>&1 | grep "awsem prof\|value" | cut -c1-200
