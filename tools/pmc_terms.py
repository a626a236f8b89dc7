#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run of tools/term_profile.py: per step-kernel dispatch (= per term configuration, in
the order term_profile.py prints them) the counters divided by replica-steps."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
replica_steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1024 * 50
rows = []
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
disp = collections.OrderedDict()
for r in sorted(rows, key=lambda r: int(r["Dispatch_Id"])):
    if "awsem_step_kernel" not in r["Kernel_Name"]:
        continue
    disp.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({k for v in disp.values() for k in v})
print("dispatch " + " ".join(f"{n:>22s}" for n in names))
for i, (k, v) in enumerate(disp.items()):
    print(f"{i:8d} " + " ".join(f"{v.get(n, 0) / replica_steps:22.1f}" for n in names))
