"""Per-term parity table: for every golden protein and energy column, the largest deviation of the HIP
path from the reference's golden energies (tests/golden/<p>_energies.csv, OpenMM Reference platform, kJ/mol) over the 12
frames -- fp64 check mode (absolute), fp32 production mode (relative to the column's largest magnitude), both paths -- and
the force error of the fp32 path against the fp64 oracle.  Usage (GPU box): python tools/parity_table.py > profiles/<tag>_parity_table.md"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from openawsem_amd import _capi
from oracle import awsem_oracle
from tests.common import COLUMNS, FORCE_GROUP, PROTEINS, golden_system

print("| protein | residues | column | max abs dE fp64 resident | max abs dE fp64 cell-list | max rel dE fp32 resident | max rel dE fp32 cell-list |")
print("|---|---|---|---|---|---|---|")
force_rows = []
for protein in PROTEINS:
    oa, traj, golden = golden_system(protein)
    e = {}
    for cell in ("0", "1"):
        os.environ["AWSEM_FORCE_CELL"] = cell
        for prec in (1, 0):
            ctx = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=len(traj), precision=prec)
            oa.system.upload(ctx)
            for r in range(len(traj)):
                ctx.set_positions(traj[r], replica=r)
            e[cell, prec] = ctx.energy_by_group()
            if prec == 0:
                cpu = awsem_oracle.make_context(oa.system, n_replicas=2, threads=2)
                for r, f in enumerate((0, 11)):
                    cpu.set_positions(traj[f], replica=r)
                fc = cpu.forces()
                fg = ctx.forces()[[0, 11]]
                force_rows.append((protein, "cell-list" if cell == "1" else "resident", np.linalg.norm(fg - fc) / np.linalg.norm(fc), np.abs(fg - fc).max(), np.abs(fc).max()))
            ctx.close()
    for c in COLUMNS:
        g = golden[c]
        if not np.any(g):
            continue
        col = FORCE_GROUP[c]
        scale = np.abs(g).max()
        print(f"| {protein} | {oa.nres} | {c} | {np.abs(e['0', 1][:, col] - g).max():.1e} | {np.abs(e['1', 1][:, col] - g).max():.1e} | "
              f"{np.abs(e['0', 0][:, col] - g).max() / scale:.1e} | {np.abs(e['1', 0][:, col] - g).max() / scale:.1e} |")
print()
print("| protein | path | fp32 forces vs fp64 oracle: relative L2 error | max abs error (kJ/mol/nm) | largest force component |")
print("|---|---|---|---|---|")
for p, path, rel, mx, big in force_rows:
    print(f"| {p} | {path} | {rel:.1e} | {mx:.1e} | {big:.0f} |")
