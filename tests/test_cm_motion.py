"""Centre-of-mass motion removal.  The reference builds its System with ForceField.createSystem(topology)
(openAWSEM.py:826,878), whose default removeCMMotion=True adds a CMMotionRemover(frequency 1): at the top of every step
OpenMM subtracts the mass-weighted mean velocity from every massive particle.

PARITY UNPINNED by the reference (its goldens are energies); the oracle restates OpenMM's ReferenceRemoveCMMotionKernel, is
checked here against what that kernel must do (closed forms), and the HIP kernels of both paths against the oracle."""
import numpy as np
import pytest

from openawsem_amd import _capi
from openawsem_amd.forces import System
from oracle import awsem_oracle
from tests.common import golden_system


def _backbone_system(oa, remove):
    system = System(oa.system.masses.copy(), remove_cm_motion=remove)
    system.virtual_sites = dict(oa.system.virtual_sites)
    for f in oa.system.forces:
        if f.getForceGroup() in (20, 21):
            system.addForce(f)
    return system


def _momentum(masses, v):
    return (masses[:, None] * v).sum(0)


def _cm(masses, x):
    return (masses[:, None] * x).sum(0) / masses.sum()


def test_the_python_mirror_switches_it_on():
    oa, _, _ = golden_system("1ubq")
    assert oa.system.remove_cm_motion                        # createSystem's default
    assert not System(oa.system.masses.copy()).remove_cm_motion   # a bare openmm.System has no CMMotionRemover


def test_oracle_removes_the_mean_velocity_at_the_top_of_every_step(oracle_api):
    oa, traj, _ = golden_system("1ubq")
    m = oa.system.masses
    drift = np.array([0.3, -0.2, 0.5])
    out = {}
    for remove in (False, True):
        ctx = awsem_oracle.make_context(_backbone_system(oa, remove), n_replicas=1)
        ctx.set_positions(traj[0], replica=0)
        ctx.set_velocities_to_temperature([300.0], 4)
        v0 = ctx.get_velocities(0)
        v0[m > 0] += drift                                    # a net drift on top of the thermal velocities
        ctx.set_velocities(v0, replica=0)
        x0 = ctx.get_positions(0)
        ctx.step(1, [300.0], 0.002, 0.0, 9)                   # friction 0: no noise, v' = (v - v_cm) + dt F / m
        v1, x1 = ctx.get_velocities(0), ctx.get_positions(0)
        out[remove] = (v0, v1, x0, x1)
    (v0, v1_off, x0, _), (_, v1_on, _, x1_on) = out[False], out[True]
    vcm = _momentum(m, v0) / m.sum()
    assert np.abs(vcm - drift).max() < 0.15                # the thermal part of the mean velocity: sqrt(kT/M) = 0.034 nm/ps
    # same forces in both runs, so the two velocity sets differ by exactly the removed mean velocity on the massive particles
    assert np.abs((v1_off - v1_on)[m > 0] - vcm).max() < 1e-12 and not (v1_off - v1_on)[m == 0].any()
    # internal forces cancel (to the 3e-5 by which the reference's virtual-site weights miss 1): no momentum is left
    assert np.abs(_momentum(m, v1_on)).max() < 1e-2 * np.abs(_momentum(m, v0)).max()
    assert np.abs(_cm(m, x1_on) - _cm(m, x0)).max() < 1e-5


def test_oracle_centre_of_mass_stays_put_under_langevin_noise(oracle_api):
    """With the remover the centre of mass only feels one step's fresh noise at a time; without it the noise integrates into
    free diffusion (D = kT / (M gamma)), two orders of magnitude further over a few ps."""
    oa, traj, _ = golden_system("1ubq")
    m = oa.system.masses
    d = {}
    for remove in (False, True):
        ctx = awsem_oracle.make_context(_backbone_system(oa, remove), n_replicas=8, threads=8)
        ctx.set_positions(traj[0])
        T = np.full(8, 300.0)
        ctx.set_velocities_to_temperature(T, 4)
        x0 = np.array([_cm(m, ctx.get_positions(r)) for r in range(8)])
        ctx.step(600, T, 0.002, 1.0, 9)
        x1 = np.array([_cm(m, ctx.get_positions(r)) for r in range(8)])
        d[remove] = np.sqrt(((x1 - x0) ** 2).sum(1).mean())
    assert d[True] < 2e-3 and d[False] > 10 * d[True]


@pytest.mark.gpu
@pytest.mark.parametrize("cell", ["0", "1"])
def test_gpu_follows_the_oracle_step_for_step(cell, monkeypatch):
    monkeypatch.setenv("AWSEM_FORCE_CELL", cell)
    oa, traj, _ = golden_system("1ubq")
    m = oa.system.masses
    system = _backbone_system(oa, True)
    gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=3, precision=1)
    system.upload(gpu)
    cpu = awsem_oracle.make_context(system, n_replicas=3)
    T = [300.0, 450.0, 600.0]
    for c in (gpu, cpu):
        c.set_positions(traj[0])
        c.set_velocities_to_temperature(T, 3)
        v = c.get_velocities(1)
        v[m > 0] += [0.4, 0.0, -0.3]
        c.set_velocities(v, replica=1)
        c.step(25, T, 0.002, 1.0, 17)
        c.step(16, T, 0.002, 1.0, 17)         # a second launch: the momentum of the incoming velocities is taken again
    gpu.synchronize()
    for r in range(3):
        assert np.abs(gpu.get_positions(r) - cpu.get_positions(r)).max() < 1e-8
        assert np.abs(gpu.get_velocities(r) - cpu.get_velocities(r)).max() < 1e-6
    gpu.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cell", ["0", "1"])
@pytest.mark.parametrize("block", ["256", "1024"])
def test_gpu_total_momentum_vanishes_in_fp32(cell, block, monkeypatch):
    """Production precision, the full 16-term system: without friction a launch must leave (almost) no momentum behind, the
    centre of mass must not move, and without the remover the drift must still be there."""
    monkeypatch.setenv("AWSEM_FORCE_CELL", cell)
    monkeypatch.setenv("AWSEM_BLOCK", block)
    oa, traj, _ = golden_system("1ubq")
    m = oa.system.masses
    res = {}
    for remove in (True, False):
        system = System(oa.system.masses.copy(), remove_cm_motion=remove)
        system.virtual_sites = dict(oa.system.virtual_sites)
        for f in oa.system.forces:
            system.addForce(f)
        gpu = _capi.Context(_capi.load_hip_library(), oa.natoms, n_replicas=2, precision=0)
        system.upload(gpu)
        gpu.set_positions(traj[0])
        gpu.set_velocities_to_temperature([300.0, 300.0], 3)
        for r in range(2):
            v = gpu.get_velocities(r)
            v[m > 0] += [0.5, -0.25, 0.125]
            gpu.set_velocities(v, replica=r)
        x0 = gpu.get_positions(1)
        p0 = _momentum(m, gpu.get_velocities(1))
        gpu.step(50, [300.0, 300.0], 0.002, 0.0, 5)
        gpu.synchronize()
        res[remove] = (np.abs(_momentum(m, gpu.get_velocities(1))).max() / np.abs(p0).max(), np.abs(_cm(m, gpu.get_positions(1)) - _cm(m, x0)).max())
        gpu.close()
    assert res[True][0] < 2e-3 and res[True][1] < 2e-4          # the weights of the virtual sites miss 1 by 3e-5: a trickle of net force remains
    assert res[False][0] > 0.9 and res[False][1] > 0.02         # 50 steps x 2 fs x 0.5 nm/ps
