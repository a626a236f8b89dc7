#!/bin/bash
# SQ counters of the step kernel on the bench workload (two passes of <= 8 SQ counters each); prints per replica-step.
# Usage (GPU box, repo root): bash tools/pmc_sq.sh <tag>
set -o pipefail
tag=${1:-sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_sq1_$tag -- python3 $R/bench.py --no-cpu-baseline --no-schedule-legs --steps 3 --warmup 1 $PMC_ARGS > $O/pmc_sq1_$tag.log 2>&1 || { tail -5 $O/pmc_sq1_$tag.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS --output-format csv -d $O/pmc_sq2_$tag -- python3 $R/bench.py --no-cpu-baseline --no-schedule-legs --steps 3 --warmup 1 $PMC_ARGS > $O/pmc_sq2_$tag.log 2>&1 || { tail -5 $O/pmc_sq2_$tag.log; exit 2; }
python3 - <<PY
import csv, glob, collections
for d in ("$O/pmc_sq1_$tag", "$O/pmc_sq2_$tag"):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "awsem_step_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k in sorted(acc):
        print(f"{k:28s} {sum(acc[k]) / len(acc[k]) / (1024 * ${PMC_STEPS:-50}):12.1f}  per replica-step ({len(acc[k])} dispatches)")
PY
