#!/bin/bash
# One workload, all the evidence the bench line cites: rocprofv3 kernel stats, SQ counters (two passes), HBM traffic (two passes).
# Usage (GPU box, repo root):  bash tools/profile_workload.sh <tag> <replicas> <md_steps> [bench.py args, e.g. --folded or --workload t0958]
# Output (gpurun_out/): <tag>_kernel_stats.csv, <tag>_kernel_tail.json, <tag>_pmc_sq.txt, <tag>_pmc_hbm.json -- copy into profiles/.
# Every average is taken over the LAST dispatches of awsem_step_kernel only (tools/pmc_tail.py): the timed launches, not an
# untimed anneal before them.  Each rocprofv3 run has its own time limit; the script stops at the first failure.
set -o pipefail
tag=$1; reps=$2; steps=$3; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --no-cpu-baseline --no-schedule-legs"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- $B "$@" > $O/prof_$tag.log 2>&1 || { tail -20 $O/prof_$tag.log; exit 3; }
cat $O/prof_$tag/*/*kernel_stats.csv > $O/${tag}_kernel_stats.csv
python3 $R/tools/pmc_tail.py trace $O/prof_$tag awsem_step_kernel 20 > $O/${tag}_kernel_tail.json || exit 4
cat $O/${tag}_kernel_tail.json
grep -o '"value": [0-9.e+]*\|"kernel_ms": [0-9.e+-]*' $O/prof_$tag.log | tr '\n' ' '; echo
P="--steps 3 --warmup 1"
div=$((reps * steps))
timeout -s KILL 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_sq1_$tag -- $B $P "$@" > $O/pmc_sq1_$tag.log 2>&1 || { tail -5 $O/pmc_sq1_$tag.log; exit 5; }
timeout -s KILL 400 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_TRANS --output-format csv -d $O/pmc_sq2_$tag -- $B $P "$@" > $O/pmc_sq2_$tag.log 2>&1 || { tail -5 $O/pmc_sq2_$tag.log; exit 6; }
{ python3 $R/tools/pmc_tail.py pmc $O/pmc_sq1_$tag awsem_step_kernel 3 $div && python3 $R/tools/pmc_tail.py pmc $O/pmc_sq2_$tag awsem_step_kernel 3 $div; } | sed 's/per unit/per replica-step/' > $O/${tag}_pmc_sq.txt || exit 7
# effective clock of the profiled launches: GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS note); the durations
# are those of the SAME dispatches (kernel trace of the counter pass itself: the first launches of a run differ from the timed ones)
python3 $R/tools/pmc_tail.py trace $O/pmc_sq1_$tag awsem_step_kernel 3 > $O/${tag}_pmc_tail.json || exit 7
python3 - $O/${tag}_pmc_sq.txt $O/${tag}_pmc_tail.json $div <<'PY' >> $O/${tag}_pmc_sq.txt
import json, sys
rows = {l.split()[0]: float(l.split()[1]) for l in open(sys.argv[1]) if l.strip()}
us = json.load(open(sys.argv[2]))["mean_us"]
div = float(sys.argv[3])
if "GRBM_GUI_ACTIVE" in rows:
    ghz = rows["GRBM_GUI_ACTIVE"] * div / 8.0 / (us * 1e3)
    print(f"{'DERIVED_CLOCK_GHZ':28s} {ghz:14.3f}  GRBM_GUI_ACTIVE / 8 XCDs / mean duration of the same dispatches ({us:.1f} us)")
PY
cat $O/${tag}_pmc_sq.txt
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$tag -- $B $P "$@" > $O/pmc_fetch_$tag.log 2>&1 || { tail -5 $O/pmc_fetch_$tag.log; exit 8; }
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$tag -- $B $P "$@" > $O/pmc_write_$tag.log 2>&1 || { tail -5 $O/pmc_write_$tag.log; exit 9; }
python3 - $O/pmc_fetch_$tag $O/pmc_write_$tag $O/${tag}_pmc_hbm.json $reps $steps "$@" <<'PY'
import json, subprocess, sys
fetch_dir, write_dir, out, reps, steps = sys.argv[1:6]
args = sys.argv[6:]
def tail(d, counter):
    for line in subprocess.check_output([sys.executable, __import__("os").environ.get("GRAFT_REPO_ROOT", ".") + "/tools/pmc_tail.py", "pmc", d, "awsem_step_kernel", "3"], text=True).splitlines():
        if line.split()[0] == counter:
            return float(line.split()[1])
    raise SystemExit("missing " + counter)
fetch, write = tail(fetch_dir, "FETCH_SIZE"), tail(write_dir, "WRITE_SIZE")
workload = args[args.index("--workload") + 1] if "--workload" in args else "1r69"
rec = {"kernel": "awsem_step_kernel<float>", "fetch_size_kib_raw": fetch, "write_size_kib": write, "hbm_bytes_per_launch": (2 * fetch + write) * 1024.0,
       "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, mean over the last 3 dispatches; FETCH_SIZE x 2 (gfx950 correction, MI355X_MICROARCH.md)",
       "workload": workload, "replicas": int(reps), "md_steps": int(steps), "precision": "fp32", "folded": "--folded" in args, "bench_args": args}
json.dump(rec, open(out, "w"), indent=1)
print(json.dumps(rec))
PY
