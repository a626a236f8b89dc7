"""The reference pins energies only; forces of the oracle are checked against central finite differences of its
own (golden-pinned) energies, per force group, on golden trajectory frames."""
import numpy as np
import pytest

from oracle import awsem_oracle
from tests.common import FORCE_GROUP, golden_system


def _fd_check(ctx, x0, group_mask, atoms, h=1e-6):
    ctx.set_positions(x0, replica=0)
    f = ctx.forces(group_mask, redistribute=False)[0]
    fd = np.zeros((len(atoms), 3))
    for n, a in enumerate(atoms):
        for k in range(3):
            xp = x0.copy(); xp[a, k] += h
            xm = x0.copy(); xm[a, k] -= h
            ctx.set_positions(xp, replica=0); ep = ctx.energy(group_mask)[0]
            ctx.set_positions(xm, replica=0); em = ctx.energy(group_mask)[0]
            fd[n, k] = -(ep - em) / (2 * h)
    return f[atoms], fd


@pytest.mark.parametrize("protein,frame", [("1ubq", 5), ("1brs", 11)])
def test_oracle_forces_match_finite_differences(protein, frame, oracle_api):
    oa, traj, _ = golden_system(protein)
    ctx = awsem_oracle.make_context(oa.system, n_replicas=1)
    rng = np.random.default_rng(7)
    atoms = rng.choice(oa.natoms, size=40 if protein == "1ubq" else 16, replace=False)
    for name, g in FORCE_GROUP.items():
        if name in ("Membrane", "ER", "TBM_Q"):
            continue
        f, fd = _fd_check(ctx, traj[frame], 1 << g, atoms)
        scale = max(1.0, np.abs(fd).max())
        assert np.abs(f - fd).max() / scale < 2e-6, f"{protein} {name}: {np.abs(f - fd).max():.3e} (scale {scale:.3e})"


def test_oracle_virtual_site_force_redistribution(oracle_api):
    oa, traj, _ = golden_system("1ubq")
    ctx = awsem_oracle.make_context(oa.system, n_replicas=1)
    ctx.set_positions(traj[3], replica=0)
    raw = ctx.forces(redistribute=False)[0]
    red = ctx.forces(redistribute=True)[0]
    virtual = sorted(oa.system.virtual_sites)
    assert np.all(red[virtual] == 0)
    # ThreeParticleAverageSite: parents receive w_k * F_site (the reference's weights sum to 1.00003 / 1.00007 /
    # 1.0 for N / H / C, openAWSEM.py:685-696, so the total force is only conserved to that accuracy)
    expect = raw.copy()
    for s, (p1, p2, p3, w1, w2, w3) in oa.system.virtual_sites.items():
        for p, w in ((p1, w1), (p2, w2), (p3, w3)):
            expect[p] += w * raw[s]
        expect[s] = 0
    assert np.allclose(red, expect, atol=1e-9)
    # all terms are translation invariant: net force vanishes
    assert np.abs(raw.sum(0)).max() < 1e-6
