"""Best fixed skin at a mid-schedule temperature: anneal 800 -> T0 along the default schedule's slope, then time launches at T0.
Usage: AWSEM_SKIN=<nm> python scratch/skin_midT.py T0     (the anneal itself runs with that skin too: same trajectories, bit for bit)"""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
from openawsem_amd import _capi, workloads
T0 = float(sys.argv[1])
R = 1024
oa = workloads.build_1r69()
ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=R, device=0, precision=0)
oa.system.upload(ctx)
ctx.set_positions(oa.pdb.positions, replica=-1)
ctx.set_velocities_to_temperature(np.full(R, 800.0), 2026)
n = int(round((800.0 - T0) / 1.5))
for i in range(n):
    ctx.step(int(os.environ.get("PER_STAGE", "250")), np.full(R, 800.0 - 1.5 * i), 0.002, 1.0, 2026)     # default 5x the default stage length: closer to equilibrium at T0
for _ in range(5):
    ctx.step(50, np.full(R, T0), 0.002, 1.0, 2026)
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(100):
    ctx.step(50, np.full(R, T0), 0.002, 1.0, 2026)
ctx.synchronize()
dt = time.perf_counter() - t0
print(f"T0 {T0:.0f} K skin {os.environ.get('AWSEM_SKIN', 'rule')}: {R * 50 * 100 / dt / 1e6:.3f} M steps/s", flush=True)
