"""Order parameters and biases (reference biasTerms.py): q_value 146-188, qc_value 191-223, partial_q_value 226-267,
qbias_term 270-302, rg_term 461-486, rg_bias_term 489-528, cylindrical_rg_bias_term 531-571.

Parity unpinned by any reference fixture (the golden CSVs carry no Q / Rg column), so: the factories are checked against
a direct numpy evaluation of the reference's formulas, the oracle's forces against finite differences, and the HIP path
against the oracle (energies, forces, step-locked dynamics)."""
import os

import numpy as np
import pytest

from openawsem_amd import _capi, unit as u
from openawsem_amd.forces import System
from openawsem_amd.functionTerms import basicTerms, biasTerms
from oracle import awsem_oracle
from tests.common import GOLDEN, golden_system

NATIVE = os.path.join(GOLDEN, "1ubq-crystal_structure.pdb")


def _native_ca():
    xyz = []
    with open(NATIVE) as fh:
        for line in fh:
            if line.startswith("ATOM") and line[12:16].strip() == "CA":
                xyz.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    return np.array(xyz) / 10.0


def _numpy_q(oa, x, min_sep=3, a=0.1):
    ca, nat = x[oa.ca], _native_ca()
    n, s, cnt = len(ca), 0.0, 0
    for i in range(n):
        for j in range(n):
            if abs(i - j) >= min_sep:
                r, r0 = np.linalg.norm(ca[i] - ca[j]), np.linalg.norm(nat[i] - nat[j])
                s += np.exp(-(r - r0) ** 2 / (2 * (a * abs(i - j) ** 0.15) ** 2))
                cnt += 1
    return s / cnt


def _numpy_rg(oa, x):
    ca = x[oa.ca]
    d2 = ((ca[:, None] - ca[None]) ** 2).sum(-1)
    return np.sqrt(np.triu(d2, 1).sum() / len(ca) ** 2)


def _bias_system(oa, backbone=False):
    system = System(oa.system.masses.copy())
    system.virtual_sites = dict(oa.system.virtual_sites)
    forces = [biasTerms.q_value(oa, NATIVE),                                                  # group 1
              biasTerms.rg_term(oa),                                                          # group 2
              biasTerms.qc_value(oa, NATIVE),                                                 # group 3
              biasTerms.qbias_term(oa, 0.7, NATIVE, "ALL", forceGroup=4),                     # group 4
              biasTerms.rg_bias_term(oa, k=5 * u.kilocalorie_per_mole, rg0=0.9, forceGroup=27),
              biasTerms.cylindrical_rg_bias_term(oa, k=3 * u.kilocalorie_per_mole, rg0=0.5, atomGroup=list(range(5, 60)), forceGroup=28)]
    if backbone:
        forces += [basicTerms.con_term(oa), basicTerms.chain_term(oa), basicTerms.chi_term(oa), basicTerms.excl_term(oa),
                   basicTerms.rama_term(oa)]
    for f in forces:
        system.addForce(f)
    return system


def test_factories_and_oracle_match_the_reference_formulas():
    oa, traj, _ = golden_system("1ubq")
    system = _bias_system(oa)
    q, rg, qc, qb, rb = system.forces[:5]
    n = oa.nres
    assert q.getNumBonds() == sum(abs(i - j) >= 3 for i in range(n) for j in range(n)) and q.getForceGroup() == 1
    assert qc.getNumBonds() == sum(j - i >= 10 for i in range(n) for j in range(n)) and qc.getForceGroup() == 3
    assert rg.getNumBonds() == n * (n - 1) // 2 and rg.getForceGroup() == 2
    assert abs(qb.k - 100 * 4.184) < 1e-12 and qb.x0 == 0.7 and rb.k == 5 * 4.184
    cpu = awsem_oracle.make_context(system, n_replicas=3)
    for r, f in enumerate((0, 5, 11)):
        cpu.set_positions(traj[f], replica=r)
    e = cpu.energy_by_group()
    for r, f in enumerate((0, 5, 11)):
        Q, Rg = _numpy_q(oa, traj[f]), _numpy_rg(oa, traj[f])
        assert 0.05 < Q < 1.0
        assert abs(e[r, 1] - Q) < 1e-12
        assert abs(e[r, 2] - 10 * Rg) < 1e-10
        assert abs(e[r, 4] - 0.5 * 418.4 * (Q - 0.7) ** 2) < 1e-9
        assert abs(e[r, 27] - 5 * 4.184 * (Rg - 0.9) ** 2) < 1e-9
        ca = traj[f][oa.ca][5:60]
        d2 = ((ca[:, None, :2] - ca[None, :, :2]) ** 2).sum(-1)
        assert abs(e[r, 28] - 3 * 4.184 * (np.sqrt(np.triu(d2, 1).sum() / len(ca) ** 2) - 0.5) ** 2) < 1e-9      # cylindrical: xy only
    # Q of the native CA trace itself is 1: put every CA on its crystal position
    x = traj[0].copy()
    x[oa.ca] = _native_ca()
    cpu.set_positions(x, replica=0)
    assert abs(cpu.energy_by_group()[0, 1] - 1.0) < 1e-5      # PDB prints 3 decimals of an Angstrom
    # partial Q over a residue window
    pq = biasTerms.partial_q_value(oa, NATIVE, startResidueIndex=10, endResidueIndex=40)
    assert pq.getForceGroup() == 4 and all(min(p) >= oa.ca[10] for p in pq.pairs)
    with pytest.raises(ValueError):
        biasTerms.partial_q_value(oa, NATIVE, startResidueIndex=10_000)


def test_tbm_q_term_is_a_harmonic_bias_on_q_from_rnative_dat(tmp_path):
    from openawsem_amd.functionTerms import templateTerms
    oa, traj, _ = golden_system("1ubq")
    nat = _native_ca()
    table = 10 * np.linalg.norm(nat[:, None] - nat[None], axis=-1)
    np.savetxt(tmp_path / "rnative.dat", table, fmt="%.5f")
    table = np.loadtxt(tmp_path / "rnative.dat")
    f = templateTerms.tbm_q_term(oa, k_tbm_q=1000.0, rnative_dat=str(tmp_path / "rnative.dat"), target_q=0.9)
    n = oa.nres
    assert f.getForceGroup() == 26 and f.getNumBonds() == sum(j - i >= 3 for i in range(n) for j in range(n))   # no CA pair within 2 A
    system = System(oa.system.masses.copy())
    system.addForce(f)
    cpu = awsem_oracle.make_context(system, n_replicas=1)
    cpu.set_positions(traj[6], replica=0)
    ca = traj[6][oa.ca]
    i, j = np.nonzero(np.subtract.outer(np.arange(n), np.arange(n)) <= -3)
    q = np.exp(-(np.linalg.norm(ca[i] - ca[j], axis=1) - table[i, j] / 10) ** 2 / (2 * (0.1 * (j - i) ** 0.15) ** 2)).mean()
    assert abs(cpu.energy_by_group()[0, 26] - 1000.0 * (q - 0.9) ** 2) < 1e-6


def test_oracle_forces_are_the_gradient_of_the_collective_variable_energies():
    oa, traj, _ = golden_system("1ubq")
    system = _bias_system(oa)
    cpu = awsem_oracle.make_context(system, n_replicas=1)
    x = traj[7].copy()
    cpu.set_positions(x, replica=0)
    rng = np.random.default_rng(5)
    h = 1e-6
    for group in (1, 2, 3, 4, 27, 28):
        f = cpu.forces(1 << group, redistribute=False)[0]
        assert np.abs(f).max() > 0
        for a in rng.choice(oa.ca, 6, replace=False):
            for k in range(3):
                xp, xm = x.copy(), x.copy()
                xp[a, k] += h
                xm[a, k] -= h
                cpu.set_positions(xp, replica=0)
                ep = cpu.energy(1 << group)[0]
                cpu.set_positions(xm, replica=0)
                em = cpu.energy(1 << group)[0]
                assert abs(-(ep - em) / (2 * h) - f[a, k]) <= 1e-6 * max(1.0, np.abs(f).max()), (group, a, k)
        cpu.set_positions(x, replica=0)


@pytest.mark.gpu
@pytest.mark.parametrize("cell", ["0", "1"])
@pytest.mark.parametrize("precision,etol,ftol", [(0, 2e-5, 1e-3), (1, 1e-9, 1e-7)])
def test_gpu_matches_oracle(precision, etol, ftol, cell, monkeypatch):
    monkeypatch.setenv("AWSEM_FORCE_CELL", cell)      # both paths: LDS-resident replica, and grid-wide kernels with global accumulators
    oa, traj, _ = golden_system("1ubq")
    system = _bias_system(oa)
    frames = [0, 3, 7, 11]
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=len(frames), precision=precision)
    system.upload(gpu)
    cpu = awsem_oracle.make_context(system, n_replicas=len(frames))
    for r, f in enumerate(frames):
        gpu.set_positions(traj[f], replica=r)
        cpu.set_positions(traj[f], replica=r)
    eg, ec = gpu.energy_by_group(), cpu.energy_by_group()
    for group in (1, 2, 3, 4, 27, 28):
        # plain sums go through the 2^-32 energy accumulators: ~10^4 nearly equal bonds round coherently (2e-7 on Q = 1)
        tol = max(etol, 5e-7) if group in (1, 3) else etol
        assert np.abs(eg[:, group] - ec[:, group]).max() <= tol * max(1.0, np.abs(ec[:, group]).max()), group
        fg, fc = gpu.forces(1 << group, redistribute=False), cpu.forces(1 << group, redistribute=False)
        # fp32 forces are accumulated as Q19.12 fixed point (2.4e-4 kJ/mol/nm per addition): the bonds of a bare order
        # parameter each pull with less than that, hence the absolute term
        assert np.abs(fg - fc).max() <= ftol * np.abs(fc).max() + (5e-3 if precision == 0 and cell == "0" else 1e-7 if precision else 1e-5), group   # fp64: Q31.32, 2.3e-10 per addition
    # a second evaluation starts from cleared accumulators
    assert np.array_equal(gpu.energy_by_group(), eg)
    assert np.array_equal(gpu.energy(1 << 4), eg[:, 4])


@pytest.mark.gpu
@pytest.mark.parametrize("cell", ["0", "1"])
def test_biased_dynamics_tracks_the_oracle_and_pulls_q_to_its_target(cell, monkeypatch):
    monkeypatch.setenv("AWSEM_FORCE_CELL", cell)
    oa, traj, _ = golden_system("1ubq")
    system = _bias_system(oa, backbone=True)
    R = 2
    T = np.array([300.0, 300.0])
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=R, precision=1)
    system.upload(gpu)
    cpu = awsem_oracle.make_context(system, n_replicas=R)
    for ctx in (gpu, cpu):
        for r in range(R):
            ctx.set_positions(traj[11 - r], replica=r)
        ctx.set_velocities_to_temperature(T, 4)
        ctx.step(10, T, 0.002, 1.0, 8)
    for r in range(R):
        assert np.abs(gpu.get_positions(r) - cpu.get_positions(r)).max() <= 1e-10
    # fp32, many replicas at 600 K, backbone terms only: with a stiff umbrella at Q = 0.7 the chains climb from the
    # frame's Q = 0.28 towards the target, without it they melt
    R = 8
    T = np.full(R, 600.0)
    q = {}
    for k in (20000.0, 0.0):
        system.forces[3].k = k
        gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=R, precision=0)
        system.upload(gpu)
        gpu.set_positions(traj[11])
        gpu.set_velocities_to_temperature(T, 11)
        gpu.step(3000, T, 0.002, 1.0, 12)
        q[k] = gpu.energy(1 << 1)
    assert q[20000.0].min() > 0.45 and q[0.0].max() < 0.35, q
