#!/bin/bash
# A/B of experiment builds (.variants/libawsem_<name>.so, tools/build_variant.sh) against the product library on the bench workload:
# 150 launches each, extended (800 K) and compact (300 K after the anneal) state, two rounds.  Usage (GPU box): bash tools/ab_lib.sh <name> [<name> ...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
  for name in "" "$@"; do
    lib=""; [ -n "$name" ] && lib=$R/.variants/libawsem_$name.so
    for leg in "" "--folded"; do
      AWSEM_HIP_LIBRARY=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps 150 --warmup 5 $leg $AB_ARGS > /tmp/b.json 2> /tmp/b.err || { cat /tmp/b.err; exit 2; }
      python -c "import json; d=json.load(open('/tmp/b.json')); print('[%-10s %-8s] value %.3fM steps/s  kernel_ms %.4f' % ('${name:-product}', '${leg:-extended}', d['value']/1e6, d['roofline']['kernel_ms']))"
    done
  done
done
