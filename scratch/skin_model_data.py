"""Calibration data for a skin rule that knows how compact the chains are: for a state (schedule slope, T0) print the mean number of
listed pairs per replica at a reference skin (neighbour queries on 8 replicas) -- the throughput per skin comes from scratch/skin_midT.py.
Usage: python scratch/skin_model_data.py <steps per stage> <T0>"""
import sys
import numpy as np
sys.path.insert(0, ".")
from openawsem_amd import _capi, workloads
per_stage, T0 = int(sys.argv[1]), float(sys.argv[2])
R = 1024
oa = workloads.build_1r69()
ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=R, device=0, precision=0)
oa.system.upload(ctx)
ctx.set_positions(oa.pdb.positions, replica=-1)
ctx.set_velocities_to_temperature(np.full(R, 800.0), 2026)
for i in range(int(round((800.0 - T0) / 1.5))):
    ctx.step(per_stage, np.full(R, 800.0 - 1.5 * i), 0.002, 1.0, 2026)
ctx.synchronize()
tot = []
for r in range(0, R, R // 8):
    n = len(ctx.neighbors(_capi.LIST_CONTACT, 0, r, 0.1)) + len(ctx.neighbors(_capi.LIST_EXCL, 0, r, 0.1))
    n += sum(len(ctx.neighbors(_capi.LIST_BETA, k, r, 0.1)) for k in range(3)) + sum(len(ctx.neighbors(_capi.LIST_PAP, k, r, 0.1)) for k in range(2))
    tot.append(n)
print(f"per_stage {per_stage} T0 {T0:.0f}: listed pairs at skin 0.1: mean {np.mean(tot):.0f} (min {min(tot)}, max {max(tot)})", flush=True)
