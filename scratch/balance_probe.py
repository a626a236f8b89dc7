"""How much of the compact state's launch tail is the placement of replicas on CUs?  Blocks b, b+256, b+512, b+768 share a CU
(tools/wall_analyze.py); permute the replicas so that each CU's four have a balanced pair count, re-time.  GPU box only."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from openawsem_amd import _capi, workloads

R = 1024
oa = workloads.build_1r69()
api = _capi.load_hip_library()
ctx = _capi.Context(api, oa.natoms, n_replicas=R, device=0, precision=0)
oa.system.upload(ctx)
ctx.set_positions(oa.pdb.positions, replica=-1)
seed = 2026
ctx.set_velocities_to_temperature(np.full(R, 800.0), seed)
for i in range(400):
    ctx.step(250, np.full(R, 800.0 - 500.0 * i / 400), 0.002, 1.0, seed)
ctx.synchronize()

def timed(tag, n=20):
    for _ in range(3):
        ctx.step(50, np.full(R, 300.0), 0.002, 1.0, seed)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.step(50, np.full(R, 300.0), 0.002, 1.0, seed)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    print(f"[{tag}] {R * 50 * n / dt / 1e6:.3f} M steps/s  kernel_ms {ctx.last_kernel_ms():.4f}", flush=True)

def cost():
    x, _ = ctx.get_state(True, False)
    x = np.asarray(x, dtype=np.float32).reshape(R, -1, 3)
    c = np.empty(R)
    for r in range(R):
        d2 = ((x[r, :, None, :] - x[r, None, :, :]) ** 2).sum(-1)
        c[r] = (d2 < 1.0).sum()
    return c

def apply(perm):
    x, v = ctx.get_state(True, True)
    x = np.asarray(x).reshape(R, -1, 3); v = np.asarray(v).reshape(R, -1, 3)
    ctx.set_state(np.ascontiguousarray(x[perm]), np.ascontiguousarray(v[perm]))

timed("as annealed")
c = cost()
print("pair-count cost: mean %.0f sd %.0f min %.0f max %.0f" % (c.mean(), c.std(), c.min(), c.max()))
rng = np.random.default_rng(1)
apply(rng.permutation(R)); timed("random permutation")
c = cost()
order = np.argsort(-c)            # rank -> replica, heaviest first
perm = np.empty(R, dtype=int)     # block -> replica
for b in range(R):
    row, col = divmod(b, 256)
    perm[b] = order[row * 256 + (255 - col if row & 1 else col)]
apply(perm); timed("snake by pair count")
c = cost(); g = c.reshape(4, 256).sum(0)
print("per-CU cost after balancing: sd/mean %.4f (replica sd/mean %.4f)" % (g.std() / g.mean(), c.std() / c.mean()))
timed("snake, 20 launches later")
order = np.argsort(-cost()); apply(order); timed("sorted descending (anti-balanced: heavy together? no: rows)")
apply(np.argsort(-cost()).reshape(256, 4).T.reshape(-1)); timed("heaviest four share a CU (worst case)")
