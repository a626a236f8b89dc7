#!/usr/bin/env python3
"""Mean of each counter (or of the kernel duration) over the LAST n dispatches of one kernel in a rocprofv3 CSV output directory.
The bench legs that matter are the timed launches at the end of a run; earlier dispatches of the same kernel (an untimed
anneal, warm-up launches of another length) must not be averaged in.

  python tools/pmc_tail.py pmc   <dir> <kernel-substring> <n> [divisor]   -> "COUNTER  mean/divisor" lines
  python tools/pmc_tail.py trace <dir> <kernel-substring> <n>             -> JSON: mean / min / max duration (us) of those dispatches
"""
import collections
import csv
import glob
import json
import sys


def main():
    mode, d, kernel, n = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    if mode == "pmc":
        div = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0
        per = collections.defaultdict(dict)           # counter -> dispatch id -> value
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel in r["Kernel_Name"]:
                    per[r["Counter_Name"]][int(r["Dispatch_Id"])] = per[r["Counter_Name"]].get(int(r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
        if not per:
            raise SystemExit(f"no counter rows for {kernel} under {d}")
        for c in sorted(per):
            ids = sorted(per[c])[-n:]
            print(f"{c:28s} {sum(per[c][i] for i in ids) / len(ids) / div:14.1f}  per unit ({len(ids)} of {len(per[c])} dispatches, divisor {div:g})")
    else:
        rows = []
        for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel in r["Kernel_Name"]:
                    rows.append((int(r["Dispatch_Id"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        if not rows:
            raise SystemExit(f"no dispatches of {kernel} under {d}")
        us = [t for _, t in sorted(rows)[-n:]]
        print(json.dumps({"kernel": kernel, "dispatches_averaged": len(us), "dispatches_total": len(rows), "mean_us": sum(us) / len(us),
                          "min_us": min(us), "max_us": max(us), "source": "rocprofv3 --kernel-trace, last dispatches of the run (the timed launches)"}))


if __name__ == "__main__":
    main()
