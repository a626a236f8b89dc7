#!/usr/bin/env python3
"""Timing of the cell-list path on the synthetic multi-chain system (BASELINE.json configs[4]):
python tools/cell_profile.py [--chains 122] [--md-steps 20] [--reps 5]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from openawsem_amd import _capi, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chains", type=int, default=122)
    ap.add_argument("--md-steps", type=int, default=20)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--replicas", type=int, default=1)
    args = ap.parse_args()
    t0 = time.time()
    oa = workloads.build_synthetic(n_chains=args.chains)
    print(f"# {oa.nres} residues, {oa.natoms} atoms, built in {time.time() - t0:.1f} s")
    t0 = time.time()
    ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, args.replicas)
    oa.system.upload(ctx)
    print(f"# finalize {time.time() - t0:.1f} s")
    ctx.set_positions(oa.pdb.positions)
    T = np.full(args.replicas, 300.0)
    ctx.set_velocities_to_temperature(T, 1)
    e = ctx.energy_by_group()
    print("# eval (first, builds lists) ms:", ctx.last_kernel_ms(), " E[20..22] =", e[0, 20:23])
    ctx.energy_by_group()
    print("# eval (lists valid) ms:", ctx.last_kernel_ms())
    for kind, name in ((_capi.LIST_CONTACT, "contact"), (_capi.LIST_EXCL, "excl")):
        print(f"# {name} pairs inside cutoff + 0.15 nm:", len(ctx.neighbors(kind, 0, 0, 0.15)))
    ms = []
    for _ in range(args.reps):
        ctx.step(args.md_steps, T, 0.002, 1.0, 1)
        ms.append(ctx.last_kernel_ms())
    m = float(np.median(ms))
    print(f"step launches ms: {['%.2f' % x for x in ms]}  -> {m / args.md_steps * 1e3:.1f} us/step, "
          f"{args.replicas * args.md_steps / (m * 1e-3):.0f} steps/s")
    assert np.isfinite(ctx.energy()).all()


if __name__ == "__main__":
    main()
