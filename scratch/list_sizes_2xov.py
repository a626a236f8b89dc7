"""Pair-list sizes of the 2xov bench state (start structure, a few launches at the bench's temperature): entries per list and replica."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openawsem_amd import _capi, workloads
R = 16
oa = workloads.build_2xov()
ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, R)
oa.system.upload(ctx)
ctx.set_positions(oa.pdb.positions)
ctx.set_velocities_to_temperature(np.full(R, 800.0), 2026)
for i in range(10):
    ctx.step(50, np.full(R, 800.0), 0.002, 1.0, 7)
kinds = {"excl": (_capi.LIST_EXCL, 1), "contact": (_capi.LIST_CONTACT, 1), "beta": (_capi.LIST_BETA, 3), "pap": (_capi.LIST_PAP, 2)}
for skin in (0.0, 0.10, 0.15, 0.25):
    for name, (kind, n) in kinds.items():
        for t in range(n):
            sizes = [len(ctx.neighbors(kind, t, r, skin)) for r in range(0, 16, 4)]
            print(f"skin {skin:.2f} {name}[{t}]: mean {np.mean(sizes):.0f} entries (min {min(sizes)}, max {max(sizes)})")
