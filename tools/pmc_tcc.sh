#!/bin/bash
# L2 (TCC) hit / miss counters of the step kernel on the bench workload.  Usage (GPU box): bash tools/pmc_tcc.sh <tag>
set -o pipefail
tag=${1:-tcc}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $O/pmc_tcc_$tag -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 ${@:2} > $O/pmc_tcc_$tag.log 2>&1 || { tail -5 $O/pmc_tcc_$tag.log; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/pmc_tcc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "awsem_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(f"{k:16s} {sum(acc[k]) / len(acc[k]) / (1024 * 50):12.1f}  per replica-step ({len(acc[k])} dispatches)")
PY
