#!/bin/bash
# Sweep one environment variable over values with an experiment build.  Usage: bash tools/sweep_lib_env.sh <lib name or ""> VAR v1 v2 ... ; AB_ARGS adds bench args
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
name=$1; var=$2; shift 2
lib=""; [ -n "$name" ] && lib=$R/scratch/libawsem_$name.so
for v in "$@"; do
  for leg in "" "--folded"; do
    env AWSEM_HIP_LIBRARY=$lib $var=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps ${AB_STEPS:-60} --warmup 5 $leg $AB_ARGS > /tmp/b.json 2> /tmp/b.err || { cat /tmp/b.err; exit 2; }
    python -c "import json; d=json.load(open('/tmp/b.json')); print('[%s=%-8s %-8s] value %.3fM steps/s  kernel_ms %.4f' % ('$var', '$v', '${leg:-extended}', d['value']/1e6, d['roofline']['kernel_ms']))"
  done
done
