"""External-field terms beyond the membrane slab: pulling_term (reference biasTerms.py:574-605) and
membrane_preassigned_side_term (membraneTerms.py:177-215), both CustomExternalForce in the reference.

Parity unpinned by any reference fixture; both are closed forms, so the oracle is checked against numpy and the HIP kernels
(both paths, both precisions) against the oracle, including the redistribution of the pull on a residue's virtual sites."""
import numpy as np
import pytest

from openawsem_amd import _capi, unit as u
from openawsem_amd.forces import System
from openawsem_amd.functionTerms import biasTerms, membraneTerms
from oracle import awsem_oracle
from tests.common import golden_system


def _side_file(tmp_path, oa):
    path = tmp_path / "PredictedZimSide"
    path.write_text("".join(("down", "middle", "up")[i % 3] + "\n" for i in range(oa.nres)))
    return str(path)


def _system(oa, tmp_path):
    system = System(oa.system.masses.copy())
    system.virtual_sites = dict(oa.system.virtual_sites)
    system.addForce(biasTerms.pulling_term(oa, k_pulling=3.0, forceDirect="y", appliedToResidue="LAST"))            # group 19
    system.addForce(biasTerms.pulling_term(oa, k_pulling=2.0, forceDirect="x", appliedToResidue=5, forceGroup=18))
    system.addForce(membraneTerms.membrane_preassigned_side_term(oa, k=2 * u.kilocalorie_per_mole,
                                                                 membrane_center=1 * u.angstrom, zimFile=_side_file(tmp_path, oa)))   # group 24
    return system


def test_factories_and_oracle_match_the_closed_forms(tmp_path, oracle_api):
    oa, traj, _ = golden_system("1ubq")
    system = _system(oa, tmp_path)
    last, fifth, side = system.forces
    assert sorted(last.atoms) == [i for i in range(oa.natoms) if oa.resi[i] == oa.nres - 1] and last.getForceGroup() == 19
    assert sorted(fifth.atoms) == [i for i in range(oa.natoms) if oa.resi[i] == 4] and fifth.k == 2.0
    assert side.getNumParticles() == oa.nres and side.k == 2 * 4.184 and abs(side.center - 0.1) < 1e-12
    assert membraneTerms.SideToZ_m(" up\n") == 1.5 and membraneTerms.SideToZ_m("down") == -1.5 and membraneTerms.SideToZ_m("middle") == 0
    with pytest.raises(ValueError):
        biasTerms.pulling_term(oa, forceDirect="w")
    x = traj[4] - traj[4].mean(0)
    cpu = awsem_oracle.make_context(system, n_replicas=1)
    cpu.set_positions(x, replica=0)
    e = cpu.energy_by_group()[0]
    assert abs(e[19] - (-3.0 * x[last.atoms, 1].sum())) < 1e-10
    assert abs(e[18] - (-2.0 * x[fifth.atoms, 0].sum())) < 1e-10
    zm = np.array(side.param)
    assert abs(e[24] - 2 * 4.184 * np.abs(x[side.atoms, 2] - 0.1 - zm).sum()) < 1e-9
    f = cpu.forces(redistribute=False)[0]
    want = np.zeros_like(f)
    want[last.atoms, 1] += 3.0
    want[fifth.atoms, 0] += 2.0
    want[side.atoms, 2] -= 2 * 4.184 * np.sign(x[side.atoms, 2] - 0.1 - zm)
    assert np.abs(f - want).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("precision,tol", [(0, 1e-5), (1, 1e-9)])
@pytest.mark.parametrize("cell", ["0", "1"])
def test_gpu_matches_oracle(tmp_path, precision, tol, cell, monkeypatch):
    monkeypatch.setenv("AWSEM_FORCE_CELL", cell)
    oa, traj, _ = golden_system("1ubq")
    system = _system(oa, tmp_path)
    frames = [traj[k] - traj[k].mean(0) for k in (0, 6, 11)]
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=len(frames), precision=precision)
    system.upload(gpu)
    cpu = awsem_oracle.make_context(system, n_replicas=len(frames))
    for r, x in enumerate(frames):
        gpu.set_positions(x, replica=r)
        cpu.set_positions(x, replica=r)
    eg, ec = gpu.energy_by_group(), cpu.energy_by_group()
    for g in (18, 19, 24):
        assert np.abs(ec[:, g]).min() > 0.1
        assert (np.abs(eg[:, g] - ec[:, g]) <= tol * np.maximum(np.abs(ec[:, g]), 1.0) + 1e-7).all(), g
    for redistribute in (False, True):      # the pull on a residue's N / C / H sites reaches their parents
        fg, fc = gpu.forces(redistribute=redistribute), cpu.forces(redistribute=redistribute)
        assert np.abs(fc).max() > 1.0
        assert np.abs(fg - fc).max() <= (5e-3 if precision == 0 else 1e-7)
    gpu.close()


@pytest.mark.gpu
def test_pulling_moves_the_residue_along_the_axis():
    oa, traj, _ = golden_system("1ubq")
    def run(k):
        system = System(oa.system.masses.copy())
        system.virtual_sites = dict(oa.system.virtual_sites)
        for f in oa.system.forces:
            if f.getForceGroup() == 20:      # backbone keeps the chain together
                system.addForce(f)
        system.addForce(biasTerms.pulling_term(oa, k_pulling=k, forceDirect="z", appliedToResidue="LAST"))
        gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=8, precision=0)
        system.upload(gpu)
        gpu.set_positions(traj[0])
        gpu.set_velocities_to_temperature(np.full(8, 300.0), 5)
        gpu.step(2000, np.full(8, 300.0), 0.002, 1.0, 5)
        z = np.mean([gpu.get_positions(r)[oa.ca[-1], 2] for r in range(8)])
        gpu.close()
        return z
    assert run(400.0) > run(0.0) + 0.5      # nm, same noise in both runs
