#!/usr/bin/env python3
"""Per-CU view of the workgroup stamps written by a -DAWSEM_WALL build (AWSEM_PROF_DUMP=<file>): python tools/wall_analyze.py <file> ..."""
import collections
import sys

import numpy as np

for path in sys.argv[1:]:
    rows = [l.split() for l in open(path)]
    en = np.array([int(r[2]) for r in rows]) * 0.01
    hw = np.array([int(r[3], 16) for r in rows])
    key = ((hw >> 32) << 16) | ((hw & 0xffffffff) >> 8 & 0xff)
    d = collections.defaultdict(list)
    for k, e in zip(key, en):
        d[k].append(e)
    last = np.array([max(v) for v in d.values()]); first = np.array([min(v) for v in d.values()]); mean = np.array([np.mean(v) for v in d.values()])
    print(f"{path}: span {en.max():.0f} us, {len(d)} CUs; workgroup end mean {en.mean():.0f} sd {en.std():.0f}; per CU: mean end {mean.mean():.0f} sd {mean.std():.0f}, "
          f"last end {last.mean():.0f} sd {last.std():.0f}, first..last end median {np.median(last - first):.0f}")
