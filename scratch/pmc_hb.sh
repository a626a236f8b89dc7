#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAVES SQ_BUSY_CYCLES --output-format csv -d $O/pmc_hb -- python3 $R/bench.py --no-cpu-baseline --workload synthetic-hb-20k --steps 2 --warmup 1 > $O/pmc_hb.log 2>&1 || { tail -5 $O/pmc_hb.log; exit 5; }
python3 - $O/pmc_hb <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "hb_eval" in k or "pairs" in k or "density" in k:
        print(k, {c: round(sum(v[-20:]) / len(v[-20:])) for c, v in d.items()})
PY
