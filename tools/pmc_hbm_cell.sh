#!/bin/bash
# HBM traffic of the cell-list step: FETCH_SIZE and WRITE_SIZE (separate passes) summed over the kernels of a step, per MD step and per
# bench launch.  Usage (GPU box): bash tools/pmc_hbm_cell.sh <tag> [workload]  ->  gpurun_out/<tag>_<workload>_pmc_hbm.json
set -o pipefail
tag=$1; w=${2:-synthetic}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_${c}_${tag}_$w -- python3 $R/bench.py --no-cpu-baseline --workload $w --steps 3 --warmup 1 > $O/pmc_${c}_${tag}_$w.log 2>&1 || { tail -5 $O/pmc_${c}_${tag}_$w.log; exit 5; }
done
python3 - $O/pmc_FETCH_SIZE_${tag}_$w $O/pmc_WRITE_SIZE_${tag}_$w $O/${tag}_${w}_pmc_hbm.json $w <<'PY'
import csv, glob, json, sys, collections
def collect(d, counter):
    per = collections.defaultdict(float); steps = 0
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter: continue
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if not k.startswith("cell_"): continue
            per[k] += float(r["Counter_Value"])
            if k == "cell_integrate_kernel": steps += 1
    return per, steps
fetch, nf = collect(sys.argv[1], "FETCH_SIZE"); write, nw = collect(sys.argv[2], "WRITE_SIZE")
kib_step = {k: (2 * fetch.get(k, 0) / max(nf, 1) + write.get(k, 0) / max(nw, 1)) for k in sorted(set(fetch) | set(write))}
per_step = sum(kib_step.values()) * 1024.0
rec = {"kernel": "cell-list step (kernel sequence)", "hbm_bytes_per_md_step": per_step, "hbm_bytes_per_launch": per_step * 20, "md_steps_counted": [nf, nw],
       "kib_per_md_step_by_kernel": {k: round(v, 1) for k, v in kib_step.items()},
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, summed over every cell_* dispatch of the run (set-up, warm-up and timed launches alike) and divided by the number of integrator dispatches; FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md)",
       "workload": sys.argv[4], "replicas": 1, "md_steps": 20, "precision": "fp32", "folded": False}
json.dump(rec, open(sys.argv[3], "w"), indent=1)
print(json.dumps(rec))
PY
