#!/usr/bin/env python3
"""Summarise rocprofv3 result databases (rocpd sqlite, the default output of ROCm 7.2's rocprofv3):
  python tools/rocpd_summary.py stats <results.db>            -> the --stats kernel table as CSV
  python tools/rocpd_summary.py pmc <results.db> [kernel-substring]  -> per kernel: mean counter value per dispatch
"""
import sqlite3
import sys


def main():
    mode, path = sys.argv[1], sys.argv[2]
    cur = sqlite3.connect(path).cursor()
    if mode == "stats":
        print('"Name","Calls","TotalDurationUs","AverageUs","Percentage"')
        for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
            print(f'"{name}",{calls},{total:.3f},{avg:.3f},{pct:.3f}')
    else:
        like = sys.argv[3] if len(sys.argv) > 3 else ""
        q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value) from counters_collection "
             "where kernel_name like ? group by kernel_name, counter_name")
        print("kernel | counter | dispatches | mean | min | max")
        for row in cur.execute(q, (f"%{like}%",)):
            print(" | ".join(str(x) if not isinstance(x, float) else f"{x:.4f}" for x in row))


if __name__ == "__main__":
    main()
