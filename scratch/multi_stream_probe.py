"""Would independent streams hide the launch tail?  K contexts of 1024/K replicas each (own stream, AWSEM_BLOCK=256), launches issued
round-robin without a common barrier, compact state at 300 K.  GPU box only."""
import os, sys, time
import numpy as np
sys.path.insert(0, ".")
os.environ["AWSEM_BLOCK"] = "256"
from openawsem_amd import _capi, workloads

oa = workloads.build_1r69()
api = _capi.load_hip_library()
seed = 2026

def run(K, n=150):
    R = 1024 // K
    ctxs = []
    for k in range(K):
        c = _capi.Context(api, oa.natoms, n_replicas=R, device=0, precision=0)
        oa.system.upload(c)
        c.set_positions(oa.pdb.positions, replica=-1)
        c.set_velocities_to_temperature(np.full(R, 800.0), seed + k)
        ctxs.append(c)
    for i in range(400):
        for k, c in enumerate(ctxs):
            c.step(250, np.full(R, 800.0 - 500.0 * i / 400), 0.002, 1.0, seed + k)
    T = np.full(R, 300.0)
    for _ in range(5):
        for k, c in enumerate(ctxs):
            c.step(50, T, 0.002, 1.0, seed + k)
    for c in ctxs:
        c.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for k, c in enumerate(ctxs):
            c.step(50, T, 0.002, 1.0, seed + k)
    for c in ctxs:
        c.synchronize()
    dt = time.perf_counter() - t0
    print(f"[{K} streams x {R} replicas, prio {os.environ.get('AWSEM_PRIO', 'default')}] {1024 * 50 * n / dt / 1e6:.3f} M steps/s", flush=True)
    for c in ctxs:
        c.close()

for prio in ("3", "1"):
    os.environ["AWSEM_PRIO"] = prio
    for K in (1, 2, 4):
        run(K)
