#!/bin/bash
# Phase-cycle profile of the step kernel (-DAWSEM_PROF build in scratch/libawsem_PROF.so); GPU box only.
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp openawsem_amd/csrc/libawsem_hip.so /tmp/libawsem_hip.orig.so
cp scratch/libawsem_PROF.so openawsem_amd/csrc/libawsem_hip.so
AWSEM_PROF_DUMP=1 timeout -k 10 300 python bench.py --no-cpu-baseline ${@:1} 2>&1 | grep "awsem prof\|value" | cut -c1-200
cp /tmp/libawsem_hip.orig.so openawsem_amd/csrc/libawsem_hip.so
