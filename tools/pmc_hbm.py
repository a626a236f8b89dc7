#!/usr/bin/env python3
"""HBM traffic of the step kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE, collected separately with
--output-format csv):  python tools/pmc_hbm.py <fetch_dir> <write_dir> <out.json> [workload replicas md_steps precision]
FETCH_SIZE is doubled (gfx950 counts 128-B read requests as 64 B: MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact."""
import csv
import glob
import json
import sys


def mean_kib(d, counter, kernel="awsem_step_kernel"):
    vals = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and kernel in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel} under {d}")
    return sum(vals) / len(vals), len(vals)


def main():
    fetch, nf = mean_kib(sys.argv[1], "FETCH_SIZE")
    write, nw = mean_kib(sys.argv[2], "WRITE_SIZE")
    extra = sys.argv[4:8] + [None] * 4
    rec = {"kernel": "awsem_step_kernel<float>", "fetch_size_kib_raw": fetch, "write_size_kib": write, "dispatches": [nf, nw],
           "hbm_bytes_per_launch": (2 * fetch + write) * 1024.0,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x 2 (gfx950 correction)",
           "workload": extra[0] or "1r69", "replicas": int(extra[1] or 1024), "md_steps": int(extra[2] or 50), "precision": extra[3] or "fp32"}
    json.dump(rec, open(sys.argv[3], "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
