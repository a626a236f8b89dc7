#!/bin/bash
# Workgroup residency of the step kernel's last launch (-DAWSEM_WALL build, tools/build_variant.sh WALL -DAWSEM_WALL) for the priority modes given.
# Usage (GPU box): bash tools/wall_prio.sh [--folded] mode ...   -> gpurun_out/wg_stamps_<leg>_prio<mode>.txt and a summary line each
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
leg=""; tag=extended
if [ "$1" == "--folded" ]; then leg="--folded"; tag=folded; shift; fi
for mode in "$@"; do
  AWSEM_PRIO=$mode AWSEM_HIP_LIBRARY=$R/.variants/libawsem_WALL.so AWSEM_PROF_DUMP=$R/gpurun_out/wg_stamps_${tag}_prio$mode.txt timeout -k 10 300 python bench.py --no-cpu-baseline --no-schedule-legs --steps 20 $leg > /tmp/w.json 2> /tmp/w.err || { cat /tmp/w.err; exit 3; }
  python -c "import json; d=json.load(open('/tmp/w.json')); print('[prio $mode $tag] value %.3fM steps/s  kernel_ms %.4f' % (d['value']/1e6, d['roofline']['kernel_ms']))"
  grep "last launch" /tmp/w.err | cut -c1-300
done
