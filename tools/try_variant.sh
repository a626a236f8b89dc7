#!/bin/bash
# Usage (GPU box): bash tools/try_variant.sh <variant .so under scratch/> <AWSEM_BLOCK> [bench args] -- runs the bench with that library build
# (loaded through AWSEM_HIP_LIBRARY; the product library is never overwritten)
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
AWSEM_HIP_LIBRARY=$R/scratch/$1 AWSEM_BLOCK=$2 timeout -k 10 300 python bench.py --no-cpu-baseline ${@:3} > /tmp/b.json 2> /tmp/b.err || { cat /tmp/b.err; exit 2; }
python -c "import json; d=json.load(open('/tmp/b.json')); print('$1 block $2: value %.3fM steps/s  kernel_ms %.4f' % (d['value']/1e6, d['roofline']['kernel_ms']))"
