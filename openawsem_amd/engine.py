"""The handful of OpenMM classes the openawsem drivers use, backed by libawsem_hip.so.

Reference call sites: openawsem/scripts/mm_run.py:33-38,125-175 (Platform, LangevinIntegrator, CustomIntegrator,
Simulation, reporters, minimizeEnergy, step, checkpoints), openawsem/scripts/mm_analyze.py:115-178 and
tests/test_energies.py:105-122 (context.setPositions / getState(getEnergy=True, groups=...)).

Beyond the OpenMM surface, `Simulation(..., n_replicas=R)` runs R independent trajectories of the same system in one
engine handle (one workgroup per replica on the GPU); `Context.getState(replica=r)`, per-replica temperatures and
`Context.energies()` expose the batch.  There is no CPU fallback: creating a Context without a GPU raises.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

from . import _capi, unit as u
from .dcd import DCDWriter
from .forces import Force, System

KB = u.BOLTZMANN_KJ_PER_MOL_K


def _value(x, unit_):
    return x.value_in_unit(unit_) if u.is_quantity(x) else float(x)


# --------------------------------------------------------------------------- platform / integrators
class Platform:
    """Platform.getPlatformByName(name): every OpenMM platform name maps onto the MI355X engine; 'Reference' selects
    its fp64 check mode (the reference's golden energies were produced on OpenMM's Reference platform), everything
    else ('HIP', 'CUDA', 'OpenCL', 'CPU') the fp32 production kernels."""
    _names = ("HIP", "CUDA", "OpenCL", "CPU", "Reference")

    def __init__(self, name):
        self.name = name
        self.properties = {"Threads": "1", "DeviceIndex": "0", "Precision": "double" if name == "Reference" else "single"}

    @staticmethod
    def getPlatformByName(name):
        if name not in Platform._names:
            raise ValueError(f'There is no registered Platform called "{name}"')
        return Platform(name)

    @staticmethod
    def getNumPlatforms():
        return len(Platform._names)

    def getName(self):
        return self.name

    def setPropertyDefaultValue(self, key, value):
        self.properties[key] = str(value)

    def getPropertyDefaultValue(self, key):
        return self.properties[key]

    @property
    def device(self):
        return int(self.properties.get("DeviceIndex", "0"))


class LangevinIntegrator:
    """openmm.LangevinIntegrator(temperature, frictionCoeff, stepSize) -- scripts/mm_run.py:145."""

    def __init__(self, temperature, frictionCoeff, stepSize):
        self._T = _value(temperature, u.kelvin)
        self._friction = _value(frictionCoeff, u.picosecond ** -1)
        self._dt = _value(stepSize, u.picosecond)
        self._seed = 0

    def getTemperature(self):
        return self._T * u.kelvin

    def setTemperature(self, temperature):
        self._T = _value(temperature, u.kelvin)

    def getFriction(self):
        return self._friction / u.picosecond

    def setFriction(self, f):
        self._friction = _value(f, u.picosecond ** -1)

    def getStepSize(self):
        return self._dt * u.picosecond

    def setStepSize(self, dt):
        self._dt = _value(dt, u.picosecond)

    def setRandomNumberSeed(self, seed):
        self._seed = int(seed)

    def getRandomNumberSeed(self):
        return self._seed


class CustomIntegrator:
    """openmm.CustomIntegrator(stepSize) with no computations, as the reference uses it to write the un-minimised and
    minimised structure (scripts/mm_run.py:130-137): stepping advances the step counter only."""

    def __init__(self, stepSize):
        self._dt = _value(stepSize, u.picosecond)

    def getStepSize(self):
        return self._dt * u.picosecond


# --------------------------------------------------------------------------- state / context
class State:
    def __init__(self, energy=None, kinetic=None, positions=None, velocities=None, forces=None, time_ps=0.0):
        self._e, self._ke, self._x, self._v, self._f, self._t = energy, kinetic, positions, velocities, forces, time_ps

    def getPotentialEnergy(self):
        if self._e is None:
            raise RuntimeError("getState was not called with getEnergy=True")
        return self._e * u.kilojoule_per_mole

    def getKineticEnergy(self):
        if self._ke is None:
            raise RuntimeError("getState was not called with getEnergy=True")
        return self._ke * u.kilojoule_per_mole

    def getPositions(self, asNumpy=False):
        if self._x is None:
            raise RuntimeError("getState was not called with getPositions=True")
        return u.Quantity(self._x, u.nanometer)

    def getVelocities(self, asNumpy=False):
        if self._v is None:
            raise RuntimeError("getState was not called with getVelocities=True")
        return u.Quantity(self._v, u.nanometer / u.picosecond)

    def getForces(self, asNumpy=False):
        if self._f is None:
            raise RuntimeError("getState was not called with getForces=True")
        return u.Quantity(self._f, u.kilojoule_per_mole / u.nanometer)

    def getTime(self):
        return self._t * u.picosecond


def _group_mask(groups):
    if groups is None or groups == -1:
        return 0xFFFFFFFF
    if isinstance(groups, (set, frozenset, list, tuple)):
        m = 0
        for g in groups:
            m |= 1 << int(g)
        return m & 0xFFFFFFFF
    return int(groups) & 0xFFFFFFFF


class Context:
    """openmm.Context for a `forces.System`; the arithmetic happens in libawsem_hip.so."""

    def __init__(self, system: System, integrator, platform: Platform | None = None, n_replicas: int = 1):
        if not isinstance(system, System):
            raise TypeError("Context needs the System of an openawsem_amd.OpenMMAWSEMSystem")
        self.system, self.integrator = system, integrator
        self.platform = platform or Platform("HIP")
        self.n_replicas = int(n_replicas)
        api = _capi.load_hip_library()
        self._raw = _capi.Context(api, system.getNumParticles(), self.n_replicas, self.platform.device,
                                  1 if self.platform.properties.get("Precision") == "double" else 0)
        system.upload(self._raw)
        # stages of an annealing loop waiting to go out as one launch (_step / _flush)
        self._pending, self._pending_key = [], None
        self._max_stages = max(1, min(16, int(os.environ.get("AWSEM_STAGES_PER_LAUNCH", "4"))))
        self.temperatures = None        # per-replica override of the integrator temperature
        self._time_ps = 0.0
        self._have_positions = False

    # The annealing loop of scripts/mm_run.py:171-175 changes the temperature and steps 50 times, 400 times over.  Each of those steps
    # used to be a launch; now up to AWSEM_STAGES_PER_LAUNCH (default 4, at most 16) consecutive stages of equal length go out as ONE
    # (awsem_step_stages: same trajectory bit for bit, +4 % throughput).  The stages wait here until the block is full or anything else
    # wants the library handle: every access to `_ctx` sends them off first, so no query can see a state the queued steps have not reached.
    # (A report point is such an access: a run that reports after every stage -- awsem_run's default, reportFrequency = steps / 400 --
    # keeps one launch per stage; the blocks form when --reportFrequency spans several stages.)
    @property
    def _ctx(self):
        self._flush()
        return self._raw

    def _flush(self):
        if self._pending:
            (n, dt, friction, seed), rows = self._pending_key, self._pending
            self._pending, self._pending_key = [], None
            if len(rows) == 1:
                self._raw.step(n, rows[0], dt, friction, seed)
            else:
                self._raw.step_stages(n, np.stack(rows), dt, friction, seed)

    # ---- state setters
    def setPositions(self, positions, replica=-1):
        x = np.asarray(_value(positions, u.nanometer) if u.is_quantity(positions) else positions, dtype=np.float64)
        self._ctx.set_positions(x.reshape(-1, 3), replica)
        self._have_positions = True

    def setVelocities(self, velocities, replica=-1):
        v = np.asarray(_value(velocities, u.nanometer / u.picosecond) if u.is_quantity(velocities) else velocities, dtype=np.float64)
        self._ctx.set_velocities(v.reshape(-1, 3), replica)

    def setVelocitiesToTemperature(self, temperature, randomSeed=None):
        T = _value(temperature, u.kelvin) if not isinstance(temperature, (list, tuple, np.ndarray)) else np.asarray(temperature, float)
        if randomSeed is None:
            randomSeed = int(time.time() * 1e3) & 0x7FFFFFFF
        self._ctx.set_velocities_to_temperature(T, int(randomSeed))

    def computeVirtualSites(self):
        self._ctx.compute_virtual_sites()

    def setParameter(self, name, value):
        raise NotImplementedError(f"global parameter {name!r}: the MI355X AWSEM engine has no run-time expression parameters")

    def setReplicaTemperatures(self, T):
        self.temperatures = None if T is None else np.broadcast_to(np.asarray(T, float), (self.n_replicas,)).copy()

    # ---- queries
    def energies(self, groups=-1):
        """Potential energy of every replica (kJ/mol) for the given force groups."""
        return self._ctx.energy(_group_mask(groups))

    def energiesByGroup(self):
        return self._ctx.energy_by_group()

    def kineticEnergies(self):
        return self._ctx.kinetic_energy()   # reduced on the device: one value per replica

    def getState(self, getPositions=False, getVelocities=False, getForces=False, getEnergy=False, groups=-1,
                 enforcePeriodicBox=False, replica=0):
        e = ke = x = v = f = None
        mask = _group_mask(groups)
        if getEnergy:
            e = float(self.energies(mask)[replica])
            ke = float(self._ctx.kinetic_energy()[replica])
        if getPositions:
            x = self._ctx.get_positions(replica)
        if getVelocities:
            v = self._ctx.get_velocities(replica)
        if getForces:
            f = self._ctx.forces(mask, redistribute=True)[replica]
        return State(e, ke, x, v, f, self._time_ps)

    def getSystem(self):
        return self.system

    def getIntegrator(self):
        return self.integrator

    def getPlatform(self):
        return self.platform

    # ---- dynamics
    def _step(self, n):
        integ = self.integrator
        if isinstance(integ, CustomIntegrator):
            self._ctx.step_count = self._ctx.step_count + n
        elif n > 0:
            T = self.temperatures if self.temperatures is not None else integ._T
            key = (int(n), float(integ._dt), float(integ._friction), int(integ._seed))
            if key != self._pending_key:
                self._flush()
                self._pending_key = key
            self._pending.append(np.broadcast_to(np.asarray(T, dtype=np.float64), (self.n_replicas,)).copy())
            if len(self._pending) >= self._max_stages:
                self._flush()
        self._time_ps += n * _value(integ.getStepSize(), u.picosecond)


# --------------------------------------------------------------------------- minimiser
def minimize(context: Context, tolerance=10.0, maxIterations=0, replica=0):
    """L-BFGS on the massive particles with virtual sites re-derived each evaluation -- the job of OpenMM's
    LocalEnergyMinimizer behind simulation.minimizeEnergy() (scripts/mm_run.py:136,153).  Host-driven: energies and
    redistributed forces come from the GPU engine, the optimiser is scipy's L-BFGS-B.  `tolerance` is the RMS force
    criterion in kJ/mol/nm (OpenMM's default is 10)."""
    from scipy.optimize import minimize as sp_minimize
    tolerance = _value(tolerance, u.kilojoule_per_mole / u.nanometer)
    ctx = context._ctx
    massive = np.flatnonzero(context.system.masses > 0)
    x_full = ctx.get_positions(replica)

    def fun(xm):
        x_full[massive] = xm.reshape(-1, 3)
        ctx.set_positions(x_full, replica)
        ctx.compute_virtual_sites()
        e = float(context.energies()[replica])
        f = ctx.forces(redistribute=True)[replica][massive]
        return e, -f.reshape(-1)

    res = sp_minimize(fun, x_full[massive].reshape(-1), jac=True, method="L-BFGS-B",
                      options={"maxiter": int(maxIterations) if maxIterations else 5000,
                               "gtol": tolerance, "ftol": 0.0, "maxcor": 20})
    x_full[massive] = res.x.reshape(-1, 3)
    ctx.set_positions(x_full, replica)
    ctx.compute_virtual_sites()
    return res


# --------------------------------------------------------------------------- simulation + reporters
class Simulation:
    """openmm.app.Simulation(topology, system, integrator, platform) -- scripts/mm_run.py:126-175."""

    def __init__(self, topology, system, integrator, platform=None, platformProperties=None, n_replicas=1):
        self.topology, self.system, self.integrator = topology, system, integrator
        self.context = Context(system, integrator, platform, n_replicas=n_replicas)
        self.reporters = []
        self.currentStep = 0
        # asynchronous reporting (step): two snapshot slots, one worker thread that writes in fetch order
        import threading
        self._async_reports = os.environ.get("AWSEM_ASYNC_REPORT", "1") != "0"
        self._worker, self._jobs, self._worker_error, self._next_slot = None, None, None, 0
        self._slot_free = [threading.Event() for _ in range(self._NSLOT)]
        for ev in self._slot_free:
            ev.set()

    def minimizeEnergy(self, tolerance=10.0, maxIterations=0):
        """All replicas at once on the device (FIRE, awsem_minimize) on either path; should the library report the device
        minimiser unavailable (AWSEM_ELIMIT), the host-driven L-BFGS of `minimize` takes over, one replica at a time."""
        from ._capi import AwsemError
        self.flushReporters()
        tol = _value(tolerance, u.kilojoule_per_mole / u.nanometer)
        try:
            self.last_minimization = self.context._ctx.minimize(tol, int(maxIterations))
        except AwsemError as err:
            if err.code != -4:   # AWSEM_ELIMIT: not available for this system
                raise
            for r in range(self.context.n_replicas):
                minimize(self.context, tolerance, maxIterations, replica=r)

    def step(self, steps):
        """simulation.step (scripts/mm_run.py:169-175): MD launches between report points.  The built-in reporters are served in two
        halves: at a report point the values they need are FETCHED asynchronously (awsem_snapshot_begin: queued on the engine's stream
        behind the launch, into pinned host memory), and the files are WRITTEN by a worker thread once the values have arrived -- by
        which time this thread has long queued the next launches, so the GPU does not idle while checkpoints and frames are written
        (the reference's OpenMM reporters run synchronously inside step).  Files are byte-identical to synchronous reporting: same
        values, same order.  A reporter without the two halves (a user's own class with report() only) is served synchronously after
        the pending ones have been written; AWSEM_ASYNC_REPORT=0 serves every reporter that way."""
        steps = int(steps)
        end = self.currentStep + steps
        while self.currentStep < end:
            nxt = end
            for rep in self.reporters:
                k = rep._interval - self.currentStep % rep._interval
                nxt = min(nxt, self.currentStep + k)
            n = nxt - self.currentStep
            self._raise_worker_error()
            self.context._step(n)
            self.currentStep = nxt
            due = [rep for rep in self.reporters if self.currentStep % rep._interval == 0]
            if not due:
                continue
            if self._async_reports and all(getattr(rep, "_deferred", False) for rep in due) and not isinstance(self.integrator, CustomIntegrator):
                self._fetch(due)
            else:
                self.flushReporters()
                for rep in due:
                    rep.report(self)

    # ---- the two halves
    _NSLOT = 2

    def _fetch(self, due):
        c = self.context._ctx
        need_e = any(rep._needs().get("energy") for rep in due)
        replicas = {rep._needs().get("positions") for rep in due} - {None}
        with_state = any(rep._needs().get("state") for rep in due)
        if len(replicas) > 1:   # reporters of different replicas at one step: not worth a slot each
            self.flushReporters()
            for rep in due:
                rep.report(self)
            return
        slot = self._next_slot
        self._next_slot = (slot + 1) % self._NSLOT
        self._slot_free[slot].wait()          # its previous snapshot has been written (two stages ago)
        self._raise_worker_error()
        self._slot_free[slot].clear()
        replica = replicas.pop() if replicas else -1
        c.snapshot_begin(slot, 0xFFFFFFFF if need_e else 0, 0, replica, with_state)   # the built-in reporters print replica 0's energy
        job = dict(slot=slot, due=due, need_e=need_e, replica=replica, with_state=with_state, step=self.currentStep,
                   time_ps=self.context._time_ps, step_count=c.step_count)
        if self._worker is None:
            import queue
            import threading
            self._jobs = queue.Queue()
            self._worker = threading.Thread(target=self._write_loop, name="awsem-reporters", daemon=True)
            self._worker.start()
        self._jobs.put(job)

    def _write_loop(self):
        while True:
            job = self._jobs.get()
            try:
                if job is None:
                    return
                if self._worker_error is None:
                    snap = self.context._ctx.snapshot_end(job["slot"], energy=job["need_e"], replica_positions=job["replica"] >= 0,
                                                          state=job["with_state"])
                    snap.update(step=job["step"], time_ps=job["time_ps"], step_count=job["step_count"])
                    for rep in job["due"]:
                        rep._write(self, snap)
            except BaseException as err:   # handed to the thread that drives the simulation
                self._worker_error = err
            finally:
                if job is not None:
                    self._slot_free[job["slot"]].set()
                self._jobs.task_done()

    def _raise_worker_error(self):
        if self._worker_error is not None:
            err, self._worker_error = self._worker_error, None
            raise err

    def flushReporters(self):
        """Wait until every report fetched so far is on disk (called by everything that reads or changes the state; call it yourself
        before reading the output files of a run that is still alive)."""
        self.context._flush()
        if self._worker is not None:
            self._jobs.join()
        self._raise_worker_error()

    def __del__(self):
        try:
            if self._worker is not None:
                self._jobs.join()
                self._jobs.put(None)
        except Exception:
            pass

    # own checkpoint format (OpenMM's is a platform-specific binary blob): a magic line, then four arrays in numpy's .npy framing, one
    # after the other.  (A .npz archive costs a CRC-32 pass over the 18 MB of a 1024-replica state per checkpoint -- a quarter of the
    # wall time of a default annealing run; archives written by earlier versions still load.)
    _CKP_MAGIC = b"AWSEMCKP1\n"

    def saveCheckpoint(self, file, _snap=None):
        c = self.context._ctx
        if _snap is None:
            self.flushReporters()
            x, v = c.get_state()
            arrays = (x, v, np.array([self.currentStep, c.step_count], dtype=np.int64), np.array([self.context._time_ps]))
        else:   # the write half of CheckpointReporter: everything was captured when the snapshot was fetched
            x, v = _snap["state"]
            arrays = (x, v, np.array([_snap["step"], _snap["step_count"]], dtype=np.int64), np.array([_snap["time_ps"]]))

        def write(fh):
            fh.write(self._CKP_MAGIC)
            for a in arrays:
                np.save(fh, a, allow_pickle=False)
        if isinstance(file, str):
            with open(file, "wb") as fh:
                write(fh)
        else:
            write(file)

    def loadCheckpoint(self, file):
        self.flushReporters()
        fh = open(file, "rb") if isinstance(file, str) else file
        try:
            head = fh.read(len(self._CKP_MAGIC))
            if head == self._CKP_MAGIC:
                x, v, step, t = (np.load(fh, allow_pickle=False) for _ in range(4))
            else:
                fh.seek(0)
                with np.load(fh, allow_pickle=False) as z:
                    x, v, step, t = z["positions"], z["velocities"], z["step"], z["time_ps"]
        finally:
            if isinstance(file, str):
                fh.close()
        if x.shape != (self.context.n_replicas, self.system.getNumParticles(), 3):
            raise ValueError("checkpoint does not match this system / replica count")
        self.context._ctx.set_state(x, v)
        self.currentStep = int(step[0])
        self.context._ctx.step_count = int(step[1])
        self.context._time_ps = float(t[0])

    saveState, loadState = saveCheckpoint, loadCheckpoint


class _Reporter:
    def __init__(self, interval):
        self._interval = max(1, int(interval))


class StateDataReporter(_Reporter):
    """step / potentialEnergy / temperature columns in OpenMM's CSV layout (scripts/mm_run.py:157-158); reports
    replica 0 (plus the ensemble mean when there are several replicas)."""

    def __init__(self, file, reportInterval, step=False, time=False, potentialEnergy=False, kineticEnergy=False,
                 totalEnergy=False, temperature=False, separator=",", append=False, **_ignored):
        super().__init__(reportInterval)
        self._own = isinstance(file, str)
        self._out = open(file, "a" if append else "w") if self._own else file
        self._cols = dict(step=step, time=time, potentialEnergy=potentialEnergy, kineticEnergy=kineticEnergy,
                          totalEnergy=totalEnergy, temperature=temperature)
        self._sep, self._header = separator, False

    _deferred = True

    def _needs(self):
        return {"energy": self._cols["potentialEnergy"] or self._cols["totalEnergy"]}

    def report(self, simulation, state=None):
        simulation.flushReporters()
        ctx = simulation.context
        need_e = self._needs()["energy"]
        self._write(simulation, {"step": simulation.currentStep, "time_ps": ctx._time_ps, "energy": ctx.energies() if need_e else None,
                                 "kinetic": ctx.kineticEnergies()})

    def _write(self, simulation, snap):
        names = {"step": "Step", "time": "Time (ps)", "potentialEnergy": "Potential Energy (kJ/mole)",
                 "kineticEnergy": "Kinetic Energy (kJ/mole)", "totalEnergy": "Total Energy (kJ/mole)", "temperature": "Temperature (K)"}
        if not self._header:
            print('#"' + (f'"{self._sep}"').join(names[k] for k, on in self._cols.items() if on) + '"', file=self._out)
            self._header = True
        pe = float(snap["energy"][0]) if snap["energy"] is not None else 0.0
        ke = float(snap["kinetic"][0])
        # OpenMM's StateDataReporter: 3 per massive particle, minus 3 when the System has a CMMotionRemover
        dof = 3 * int((simulation.system.masses > 0).sum()) - (3 if getattr(simulation.system, "remove_cm_motion", False) else 0)
        vals = {"step": snap["step"], "time": snap["time_ps"], "potentialEnergy": pe, "kineticEnergy": ke,
                "totalEnergy": pe + ke, "temperature": 2 * ke / (dof * KB)}
        print(self._sep.join(str(vals[k]) for k, on in self._cols.items() if on), file=self._out)
        self._out.flush()

    def __del__(self):
        if getattr(self, "_own", False):
            self._out.close()


class DCDReporter(_Reporter):
    """movie.dcd writer (scripts/mm_run.py:134,161); replica 0 unless `replica` is given."""

    def __init__(self, file, reportInterval, append=False, enforcePeriodicBox=None, replica=0):
        super().__init__(reportInterval)
        self._file, self._append, self._writer, self._replica = file, append, None, replica

    _deferred = True

    def _needs(self):
        return {"positions": self._replica}

    def report(self, simulation, state=None):
        simulation.flushReporters()
        self._write(simulation, {"positions": simulation.context._ctx.get_positions(self._replica)})

    def _write(self, simulation, snap):
        if self._writer is None:
            self._writer = DCDWriter(self._file, simulation.system.getNumParticles(), append=self._append,
                                     dt_ps=_value(simulation.integrator.getStepSize(), u.picosecond), interval=self._interval)
        self._writer.write(snap["positions"])


class PDBReporter(_Reporter):
    """native.pdb writer (scripts/mm_run.py:133): one MODEL per report."""

    def __init__(self, file, reportInterval, enforcePeriodicBox=None, replica=0):
        super().__init__(reportInterval)
        self._out, self._model, self._replica = open(file, "w"), 0, replica

    _deferred = True

    def _needs(self):
        return {"positions": self._replica}

    def report(self, simulation, state=None):
        simulation.flushReporters()
        self._write(simulation, {"positions": simulation.context._ctx.get_positions(self._replica)})

    def _write(self, simulation, snap):
        self._model += 1
        x = snap["positions"] * 10.0
        print(f"MODEL     {self._model:4d}", file=self._out)
        for a in simulation.topology.atoms():
            r = a.residue
            name = a.name if len(a.name) == 4 else " " + a.name
            print("ATOM  %5d %-4s %3s %s%4s    %8.3f%8.3f%8.3f  1.00  0.00          %2s" %
                  ((a.index + 1) % 100000, name, r.name, r.chain.id, r.id, x[a.index, 0], x[a.index, 1], x[a.index, 2],
                   a.element if a.element else a.name[0]), file=self._out)
        print("ENDMDL", file=self._out)
        self._out.flush()


class CheckpointReporter(_Reporter):
    def __init__(self, file, reportInterval):
        super().__init__(reportInterval)
        self._file = file

    _deferred = True

    def _needs(self):
        return {"state": True}

    def report(self, simulation, state=None):
        simulation.saveCheckpoint(self._file)

    def _write(self, simulation, snap):
        simulation.saveCheckpoint(self._file, _snap=snap)


stdout = sys.stdout
__all__ = ["Platform", "LangevinIntegrator", "CustomIntegrator", "Context", "State", "Simulation", "StateDataReporter",
           "DCDReporter", "PDBReporter", "CheckpointReporter", "minimize", "Force", "System"]
