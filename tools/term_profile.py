#!/usr/bin/env python3
"""Per-term cost table of the step kernel (the shape of the reference's tests/data/Benchmark_data_many.csv):
builds the 1r69 ensemble with ONE term at a time (plus 'none' = integrator/virtual-site floor and 'all') and times
awsem_step launches with HIP events.  Usage: python tools/term_profile.py [--replicas 1024] [--md-steps 50]
[--state extended|collapsed]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from openawsem_amd import _capi, workloads  # noqa: E402
from openawsem_amd.forces import System  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--replicas", type=int, default=1024)
    ap.add_argument("--md-steps", type=int, default=50)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default=None, help="comma separated list of configurations to time (e.g. none,all,12:fragment_memory)")
    ap.add_argument("--lib", default=None, help="alternative libawsem_hip.so (experiments)")
    ap.add_argument("--collapse-steps", type=int, default=0, help="run this many steps with all terms first and reuse the state")
    args = ap.parse_args()
    oa = workloads.build_1r69()
    api = _capi.load_hip_library(args.lib)
    R = args.replicas
    T = np.full(R, 600.0)
    x0 = v0 = None
    if args.collapse_steps:
        ctx = _capi.Context(api, oa.natoms, R)
        oa.system.upload(ctx)
        ctx.set_positions(oa.pdb.positions)
        ctx.set_velocities_to_temperature(T, 1)
        ctx.step(args.collapse_steps, np.full(R, 300.0), 0.002, 1.0, 1)
        ctx.synchronize()
        x0 = [ctx.get_positions(r) for r in range(R)]
        ca = x0[0][oa.ca]
        print(f"# after {args.collapse_steps} steps at 300 K: Rg(replica 0) = {np.sqrt(((ca - ca.mean(0)) ** 2).sum(1).mean()):.2f} nm")
        ctx.close()
    names = ["none", "all"] + [f"{i}:{f.kind}" for i, f in enumerate(oa.system.forces)]
    print(f"{'term':24s} {'ms/launch':>10s} {'us/replica-step':>16s}")
    if args.only:
        names = [n for n in names if n in args.only.split(",")]
    for name in names:
        sysm = System(oa.system.masses.copy())
        sysm.virtual_sites = dict(oa.system.virtual_sites)
        if name == "all":
            sysm.forces = list(oa.system.forces)
        elif name != "none":
            sysm.forces = [oa.system.forces[int(name.split(":")[0])]]
        ctx = _capi.Context(api, oa.natoms, R)
        sysm.upload(ctx)
        if x0 is None:
            ctx.set_positions(oa.pdb.positions)
        else:
            for r in range(R):
                ctx.set_positions(x0[r], replica=r)
        ctx.set_velocities_to_temperature(T, 1)
        ms = []
        for _ in range(args.reps):
            if x0 is not None:   # keep the state comparable: single-term dynamics would unfold the chain
                pass
            ctx.step(args.md_steps, T, 0.00002, 1.0, 1)   # tiny dt: geometry stays put, cost is representative
            ms.append(ctx.last_kernel_ms())
        m = float(np.median(ms))
        print(f"{name:24s} {m:10.3f} {m * 1e3 / (R * args.md_steps):16.4f}")
        ctx.close()


if __name__ == "__main__":
    main()
