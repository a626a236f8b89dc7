"""Functional form of the lambda / gamma tables of the hydrogen-bond terms (include/awsem_hip.h: awsem_add_hbond_beta_fn /
awsem_add_hbond_pap_fn): the reference builds five dense nres x nres tables (hydrogenBondTerms.py:441,460,513-519,577-583,
649-658,728-736 -- its own limit on system size); the engine can evaluate an entry from the pair's residue data instead.

The dense builders are the checker (they equal the reference's own double loops: tests/test_host_tables.py).  CPU: the
functional form reproduces every entry of every dense table, and the oracle reproduces the reference's golden Beta / Pap
columns through it.  GPU: golden columns through the functional path (fp64, both kernel paths), dense and functional handles
agree, the pair lists still equal the brute-force sets."""
import numpy as np
import pytest

from openawsem_amd import _capi
from openawsem_amd.forces import System
from openawsem_amd.functionTerms import hydrogenBondTerms as hb
from oracle import awsem_oracle
from tests.common import FORCE_GROUP, energies_by_column, golden_system


def _ssweight_file(tmp_path, nres, seed=4):
    rng = np.random.default_rng(seed)
    w = np.zeros((nres, 2))
    w[np.arange(nres), rng.integers(0, 2, nres)] = 1.0       # helix (column 0) or strand (column 1), as stride2ssweight writes it
    w[rng.random(nres) < 0.2] = 0.0
    path = tmp_path / "ssweight"
    np.savetxt(path, w, fmt="%.1f")
    return str(path)


def _hb_terms(oa, ssweight):
    return [hb.beta_term_1(oa), hb.beta_term_2(oa), hb.beta_term_3(oa), hb.pap_term_1(oa, ssweightFileName=ssweight),
            hb.pap_term_2(oa, ssweightFileName=ssweight)]


def _hb_system(oa, terms):
    system = System(oa.system.masses.copy())
    system.virtual_sites = dict(oa.system.virtual_sites)
    for f in oa.system.forces:
        if f.getForceGroup() == 20:        # the excluded-volume term drives the cell-list path's rebuilds
            system.addForce(f)
    for f in terms:
        system.addForce(f)
    return system


@pytest.mark.parametrize("protein", ["1ubq", "1brs"])
def test_functional_form_reproduces_every_dense_entry(protein, tmp_path, oracle_api, capsys):
    oa, _, _ = golden_system(protein)
    for f in _hb_terms(oa, _ssweight_file(tmp_path, oa.nres)):
        dense = f.fn.dense()
        got = awsem_oracle.hb_fn_table(f.fn)
        assert np.count_nonzero(dense) > oa.nres
        assert np.abs(got - dense).max() <= 4e-16 * np.abs(dense).max(), f.fn.kind     # same operations in the same order
        assert np.array_equal(got != 0, dense != 0), f.fn.kind                          # the pair lists hold exactly the non-zero entries
    # the P-AP tables see the helix flags and the chain boundary
    g = _hb_terms(oa, _ssweight_file(tmp_path, oa.nres))[3].fn
    assert g.res_info[:, 3].any() and set(np.unique(g.dense())) >= {0.0, 0.4, 1.0}


def test_the_reference_goldens_through_the_functional_path(oracle_api, monkeypatch):
    monkeypatch.setenv("AWSEM_HB_TABLES", "functional")
    for protein in ("1brs", "2lyz"):
        oa, traj, golden = golden_system(protein)
        ctx = awsem_oracle.make_context(oa.system, n_replicas=4, threads=4)
        got = energies_by_column(ctx, traj)
        for col in ("Beta", "Pap"):
            assert np.abs(golden[col]).max() > 1.0
            assert np.allclose(got[col], golden[col], atol=1e-5, rtol=0), (protein, col)


def test_small_systems_upload_dense_large_ones_functional(monkeypatch):
    oa, _, _ = golden_system("1ubq")
    fn = hb.beta_term_2(oa).fn
    monkeypatch.delenv("AWSEM_HB_TABLES", raising=False)
    assert not fn.wants_functional()
    big = type(fn)(fn.kind, np.zeros((5000, 4), dtype=np.int32))
    assert big.wants_functional()
    monkeypatch.setenv("AWSEM_HB_TABLES", "functional")
    assert fn.wants_functional()


@pytest.mark.gpu
@pytest.mark.parametrize("cell", ["0", "1"])
def test_gpu_goldens_through_the_functional_path(cell, monkeypatch):
    monkeypatch.setenv("AWSEM_FORCE_CELL", cell)
    monkeypatch.setenv("AWSEM_HB_TABLES", "functional")
    oa, traj, golden = golden_system("1brs")
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=4, precision=1)
    oa.system.upload(gpu)
    got = energies_by_column(gpu, traj)
    for col in ("Beta", "Pap"):
        assert np.allclose(got[col], golden[col], atol=1e-5, rtol=0), col
    gpu.close()


@pytest.mark.gpu
@pytest.mark.parametrize("precision,etol,ftol", [(0, 1e-4, 1e-3), (1, 1e-9, 1e-7)])
@pytest.mark.parametrize("cell", ["0", "1"])
def test_gpu_functional_and_dense_handles_agree_and_match_the_oracle(cell, precision, etol, ftol, tmp_path, monkeypatch, capsys):
    monkeypatch.setenv("AWSEM_FORCE_CELL", cell)
    oa, traj, _ = golden_system("1brs")
    system = _hb_system(oa, _hb_terms(oa, _ssweight_file(tmp_path, oa.nres)))
    frames = [traj[k] for k in (0, 6, 11)]
    out = {}
    for mode in ("dense", "functional"):
        monkeypatch.setenv("AWSEM_HB_TABLES", mode)
        gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=len(frames), precision=precision)
        system.upload(gpu)
        for r, x in enumerate(frames):
            gpu.set_positions(x, replica=r)
        lists = [gpu.neighbors(_capi.LIST_BETA, t, replica=1, skin=0.1) for t in range(3)] + [gpu.neighbors(_capi.LIST_PAP, t, replica=1, skin=0.1) for t in range(2)]
        out[mode] = (gpu.energy_by_group(), gpu.forces(redistribute=False), lists)
        gpu.close()
    monkeypatch.setenv("AWSEM_HB_TABLES", "dense")
    cpu = awsem_oracle.make_context(system, n_replicas=len(frames), threads=len(frames))
    for r, x in enumerate(frames):
        cpu.set_positions(x, replica=r)
    ec, fc = cpu.energy_by_group(), cpu.forces(redistribute=False)
    for mode in ("dense", "functional"):
        eg, fg, _ = out[mode]
        for g in (FORCE_GROUP["Beta"], FORCE_GROUP["Pap"]):
            assert np.abs(ec[:, g]).min() > 1.0
            assert (np.abs(eg[:, g] - ec[:, g]) <= etol * np.abs(ec[:, g]) + 2e-7).all(), (mode, g)
        assert np.abs(fg - fc).max() <= ftol * np.abs(fc).max() + (5e-3 if precision == 0 and cell == "0" else 1e-7 if precision else 1e-5), mode
    for a, b in zip(out["dense"][2], out["functional"][2]):
        assert len(a) > 0 and np.array_equal(a, b)
    if precision == 1:      # fp64: the two forms give the same coefficients to the last bits
        assert np.abs(out["dense"][0] - out["functional"][0]).max() < 1e-9


@pytest.mark.gpu
def test_functional_tables_drive_dynamics_like_the_oracle(tmp_path, monkeypatch, capsys):
    monkeypatch.setenv("AWSEM_HB_TABLES", "functional")
    oa, traj, _ = golden_system("1ubq")
    system = _hb_system(oa, _hb_terms(oa, _ssweight_file(tmp_path, oa.nres)))
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=2, precision=1)
    system.upload(gpu)
    cpu = awsem_oracle.make_context(system, n_replicas=2)
    T = [300.0, 500.0]
    for c in (gpu, cpu):
        c.set_positions(traj[3])
        c.set_velocities_to_temperature(T, 3)
        c.step(30, T, 0.002, 1.0, 17)
    gpu.synchronize()
    for r in range(2):
        assert np.abs(gpu.get_positions(r) - cpu.get_positions(r)).max() < 1e-8
    gpu.close()
