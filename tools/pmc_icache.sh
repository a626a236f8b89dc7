#!/bin/bash
# Instruction-cache counters of the step kernel on the bench workload; prints per replica-step.  GPU box, repo root.
set -o pipefail
tag=${1:-ic}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH --output-format csv -d $O/pmc_ic_$tag -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 $PMC_ARGS > $O/pmc_ic_$tag.log 2>&1 || { tail -5 $O/pmc_ic_$tag.log; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/pmc_ic_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "awsem_step_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(f"{k:28s} {sum(acc[k]) / len(acc[k]) / (1024 * 50):12.1f}  per replica-step ({len(acc[k])} dispatches)")
PY
