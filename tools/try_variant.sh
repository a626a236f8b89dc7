#!/bin/bash
# Usage (GPU box): bash tools/try_variant.sh <variant .so under scratch/> <AWSEM_BLOCK> -- runs the bench with that library build
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp openawsem_amd/csrc/libawsem_hip.so /tmp/libawsem_hip.orig.so
cp scratch/$1 openawsem_amd/csrc/libawsem_hip.so
AWSEM_BLOCK=$2 timeout -k 10 300 python bench.py --no-cpu-baseline ${@:3} > /tmp/b.json 2> /tmp/b.err || { cat /tmp/b.err; cp /tmp/libawsem_hip.orig.so openawsem_amd/csrc/libawsem_hip.so; exit 2; }
python -c "import json; d=json.load(open('/tmp/b.json')); print('$1 block $2: value %.3fM steps/s  kernel_ms %.4f' % (d['value']/1e6, d['roofline']['kernel_ms']))"
cp /tmp/libawsem_hip.orig.so openawsem_amd/csrc/libawsem_hip.so
