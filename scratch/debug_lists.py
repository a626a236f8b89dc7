import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from openawsem_amd import _capi
from oracle import awsem_oracle
from tests.common import COLUMNS, FORCE_GROUP, golden_system, energies_by_column

oa, traj, golden = golden_system("1brs")
for prec in (1, 0):
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, 12, precision=prec)
    oa.system.upload(gpu)
    got = energies_by_column(gpu, traj)
    for c in COLUMNS:
        d = np.abs(got[c] - golden[c])
        print(prec, c, "max", d.max(), "frames>1e-5:", np.flatnonzero(d > 1e-5))
    cpu = _capi.Context(awsem_oracle.load(), oa.natoms, 12, precision=prec)
    oa.system.upload(cpu)
    for r in range(12):
        cpu.set_positions(traj[r], replica=r)
    for kind, cnt in ((0, 1), (1, 1), (2, 3), (3, 2)):
        for idx in range(cnt):
            for r in (8,):
                for skin in (0.0, 0.15):
                    a, b = gpu.neighbors(kind, idx, r, skin), cpu.neighbors(kind, idx, r, skin)
                    sa, sb = set(map(tuple, a)), set(map(tuple, b))
                    print(prec, "kind", kind, idx, "rep", r, "skin", skin, len(a), len(b), "dups", len(a) - len(sa), "missing", sorted(sb - sa)[:5], "extra", sorted(sa - sb)[:5])
    # repeat energy: deterministic?
    e1 = gpu.energy_by_group(); e2 = gpu.energy_by_group()
    print("repeatable:", np.array_equal(e1, e2), np.abs(e1 - e2).max())
