"""GPU parity on the BASELINE.json workloads (configs 0-3): 1r69 (single memory and the 1100-memory fragment library),
T0958 (all contact/burial/H-bond terms + helical + rama_ssweight) and the 2xov membrane protein (membrane_term +
z-dependent contact).  None of these has fp64 golden energies in the reference (its goldens are the five tests/data
proteins, covered by test_gpu_parity.py), so the check is GPU vs the fp64 CPU oracle on the same descriptors:
per-group energies <= 1e-4 relative, forces <= 1e-3 relative in fp32; 1e-9 / 1e-7 in the fp64 check mode.
The membrane term and the z-dependent contact variant are "parity unpinned" (no reference fixture at all).
"""
import os

import numpy as np
import pytest

from openawsem_amd import _capi, workloads
from oracle import awsem_oracle
from tests.common import GOLDEN

pytestmark = pytest.mark.gpu

BUILDERS = {
    "1r69": workloads.build_1r69,
    "1r69-fraglib": lambda: workloads.build_1r69(fragment_library=True),
    "T0958": workloads.build_t0958,
    "2xov": workloads.build_2xov,
}
_cache = {}


def _system(name):
    if name not in _cache:
        _cache[name] = BUILDERS[name]()
    return _cache[name]


def _relaxed_frames(oa, name):
    """Start structure plus two snapshots (400 and 3000 steps) of a 500 K Langevin run of the fp64 CPU oracle, committed as
    tests/golden/workload_frames_<name>.npy by tools/make_workload_frames.py: compact enough to populate the pair terms, and
    independent of the GPU integrator."""
    frames = np.load(os.path.join(GOLDEN, f"workload_frames_{name}.npy"))
    assert frames.shape == (3, oa.natoms, 3) and np.isfinite(frames).all()
    assert np.abs(frames[0] - np.asarray(oa.pdb.positions)).max() < 1e-12
    return list(frames)


@pytest.mark.parametrize("name", list(BUILDERS))
@pytest.mark.parametrize("precision,etol,ftol", [(0, 1e-4, 1e-3), (1, 1e-9, 1e-7)])
def test_energies_and_forces_match_oracle(name, precision, etol, ftol):
    oa = _system(name)
    frames = _relaxed_frames(oa, name)
    R = len(frames)
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=R, precision=precision)
    oa.system.upload(gpu)
    cpu = awsem_oracle.make_context(oa.system, n_replicas=R, threads=R)
    for r, x in enumerate(frames):
        gpu.set_positions(x, replica=r)
        cpu.set_positions(x, replica=r)
    eg, ec = gpu.energy_by_group(), cpu.energy_by_group()
    scale = np.maximum(np.abs(ec), 1.0)
    # 1e-8 kJ/mol absolute: the accumulators are Q31.32 (2.3e-10 per addition), and a group may sum ten thousand terms of both signs
    worst = ((np.abs(eg - ec) - 1e-8) / scale).max()
    assert worst <= etol, f"{name}: worst group energy error {worst:.3e}"
    assert np.abs(ec).sum() > 0
    fg, fc = gpu.forces(redistribute=True), cpu.forces(redistribute=True)
    for r in range(R):
        err, ref = np.linalg.norm(fg[r] - fc[r]), np.linalg.norm(fc[r])
        # fragment-memory force is piecewise constant in r: allow a handful of bonds on a bin edge in fp32
        allowed = ftol * ref * (3.0 if precision == 0 else 1.0) + 1e-4
        assert err <= allowed, f"{name} frame {r}: |dF| = {err:.3e} vs |F| = {ref:.3e}"


def test_fragment_library_has_the_bond_count_the_reference_reports():
    """examples/1r69/energy.dat:4 of the reference prints 'Number of atom pairs ... 1272' for frags.mem."""
    oa = _system("1r69-fraglib")
    fm = [f for f in oa.system.forces if f.kind == "fragment_memory"][0]
    assert fm.getNumBonds() == 1272


def test_2xov_annealing_ensemble_runs_and_stays_finite():
    oa = _system("2xov")
    R = 64
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=R, precision=0)
    oa.system.upload(gpu)
    gpu.set_positions(oa.pdb.positions)
    T = np.linspace(300.0, 600.0, R)
    gpu.set_velocities_to_temperature(T, 9)
    e0 = gpu.energy()
    gpu.step(200, T, 0.002, 1.0, 9)
    e1 = gpu.energy()
    assert np.isfinite(e1).all() and not np.array_equal(e0, e1)
    # kinetic temperature of the hot half above that of the cold half (thermostat at per-replica T)
    m = oa.system.masses
    ke = np.array([0.5 * (m[:, None] * gpu.get_velocities(r) ** 2).sum() for r in range(R)])
    assert ke[R // 2:].mean() > ke[:R // 2].mean()
