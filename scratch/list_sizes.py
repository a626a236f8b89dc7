"""Pair-list sizes of the compact 1r69 ensemble (after the bench's untimed anneal 800 -> 300 K): entries per list and replica."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openawsem_amd import _capi, workloads
R = 256
oa = workloads.build_1r69()
ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, R)
oa.system.upload(ctx)
ctx.set_positions(oa.pdb.positions)
ctx.set_velocities_to_temperature(np.full(R, 800.0), 2026)
for i in range(400):
    ctx.step(250, np.full(R, 800.0 - 500.0 * i / 400), 0.002, 1.0, 7)
kinds = {"excl": (_capi.LIST_EXCL, 1), "contact": (_capi.LIST_CONTACT, 1), "beta": (_capi.LIST_BETA, 3), "pap": (_capi.LIST_PAP, 2)}
for skin in (0.0, 0.10):
    for name, (kind, n) in kinds.items():
        for t in range(n):
            sizes = [len(ctx.neighbors(kind, t, r, skin)) for r in range(0, 64, 8)]
            print(f"skin {skin:.2f} {name}[{t}]: mean {np.mean(sizes):.0f} entries (min {min(sizes)}, max {max(sizes)})")
