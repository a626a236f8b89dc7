#!/bin/bash
cd ${GRAFT_REPO_ROOT:-$(pwd)}
for v in "" alltab "" alltab; do
  lib=""; [ -n "$v" ] && lib=$PWD/.variants/libawsem_$v.so
  AWSEM_HIP_LIBRARY=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --workload synthetic-hb-20k > /tmp/b.json 2>/tmp/b.err || { cat /tmp/b.err; exit 2; }
  python -c "import json; d=json.load(open('/tmp/b.json')); print('[${v:-product}] %.0f steps/s, %.1f us per step' % (d['value'], 1e6/d['value']))"
done
