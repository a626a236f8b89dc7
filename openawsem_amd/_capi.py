"""ctypes binding of the C ABI declared in include/awsem_hip.h.

`CApi(lib, prefix)` binds every entry point of the header under a symbol prefix: the product binds
``awsem_`` from libawsem_hip.so (see `load_hip_library`); the test oracle exports the same signatures under
``awsem_oracle_`` and is bound by oracle/awsem_oracle.py with this same class, so that one set of term
descriptors can be uploaded to either.  Nothing in this module imports or loads the oracle.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIBRARY_PATH = os.path.join(_HERE, "csrc", "libawsem_hip.so")

c_int_p = C.POINTER(C.c_int32)
c_dbl_p = C.POINTER(C.c_double)

# name -> (restype, argtypes); 'H' marks the handle
_SIGNATURES = {
    "create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "destroy": (None, ["H"]),
    "set_masses": (C.c_int, ["H", c_dbl_p, C.c_int]),
    "set_cm_motion_removal": (C.c_int, ["H", C.c_int]),
    "set_virtual_sites": (C.c_int, ["H", C.c_int, c_int_p, c_int_p, c_dbl_p]),
    "add_harmonic_bonds": (C.c_int, ["H", C.c_int, C.c_int, c_int_p, c_dbl_p, c_dbl_p]),
    "add_chi": (C.c_int, ["H", C.c_int, C.c_int, c_int_p, C.c_double, C.c_double]),
    "add_rama": (C.c_int, ["H", C.c_int, C.c_int, c_int_p, C.c_int, c_dbl_p, c_dbl_p, C.c_double]),
    "add_excl": (C.c_int, ["H", C.c_int, C.c_int, c_int_p, c_int_p, C.c_int, c_int_p, C.c_double, C.c_double]),
    "add_contact": (C.c_int, ["H", C.c_int, C.c_int, c_int_p, c_int_p, c_int_p, c_int_p, C.c_int, C.c_int, C.c_int,
                              c_dbl_p, c_dbl_p, c_dbl_p]),
    "add_hbond_beta": (C.c_int, ["H", C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, C.c_int, c_int_p, c_int_p, C.c_int,
                                 c_dbl_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "add_hbond_pap": (C.c_int, ["H", C.c_int, C.c_int, c_int_p, c_int_p, C.c_int, c_int_p, c_int_p, C.c_int, c_dbl_p,
                                C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "add_hbond_beta_fn": (C.c_int, ["H", C.c_int, C.c_int, C.c_int, c_int_p, c_int_p, C.c_int, c_int_p, c_int_p, C.c_int, C.c_int, c_int_p,
                                    c_dbl_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "add_hbond_pap_fn": (C.c_int, ["H", C.c_int, C.c_int, c_int_p, c_int_p, C.c_int, c_int_p, c_int_p, C.c_int, C.c_int, c_int_p, c_dbl_p,
                                   C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    "add_helical": (C.c_int, ["H", C.c_int, C.c_int, c_int_p, c_dbl_p, C.c_double, C.c_double, C.c_double, C.c_double]),
    "add_helical_z": (C.c_int, ["H", C.c_int, C.c_int, c_int_p, c_dbl_p, c_dbl_p, C.c_double, C.c_double, C.c_double, C.c_double,
                                C.c_double, C.c_double, C.c_double]),
    "add_fragment_memory": (C.c_int, ["H", C.c_int, C.c_int, c_int_p, C.c_int, c_dbl_p, C.c_double, C.c_double]),
    "add_membrane": (C.c_int, ["H", C.c_int, C.c_int, c_int_p, c_dbl_p, C.c_double, C.c_double, C.c_double, C.c_double]),
    "add_external": (C.c_int, ["H", C.c_int, C.c_int, C.c_int, c_int_p, c_dbl_p, C.c_double, C.c_double]),
    "add_debye_huckel": (C.c_int, ["H", C.c_int, C.c_int, c_int_p, c_dbl_p, C.c_double, C.c_double]),
    "add_gaussian_bonds": (C.c_int, ["H", C.c_int, C.c_int, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p]),
    "add_bond_sum": (C.c_int, ["H", C.c_int, C.c_int, C.c_int, c_int_p, c_dbl_p, c_dbl_p, c_dbl_p, C.c_int, C.c_double, C.c_double]),
    "finalize": (C.c_int, ["H"]),
    "set_positions": (C.c_int, ["H", C.c_int, c_dbl_p, C.c_int]),
    "get_positions": (C.c_int, ["H", C.c_int, c_dbl_p, C.c_int]),
    "set_velocities": (C.c_int, ["H", C.c_int, c_dbl_p, C.c_int]),
    "get_velocities": (C.c_int, ["H", C.c_int, c_dbl_p, C.c_int]),
    "set_state": (C.c_int, ["H", c_dbl_p, c_dbl_p]),
    "get_state": (C.c_int, ["H", c_dbl_p, c_dbl_p]),
    "kinetic_energy": (C.c_int, ["H", c_dbl_p]),
    "compute_virtual_sites": (C.c_int, ["H"]),
    "set_velocities_to_temperature": (C.c_int, ["H", c_dbl_p, C.c_uint64]),
    "scale_velocities": (C.c_int, ["H", c_dbl_p]),
    "energy": (C.c_int, ["H", C.c_uint32, c_dbl_p]),
    "energy_by_group": (C.c_int, ["H", c_dbl_p]),
    "forces": (C.c_int, ["H", C.c_uint32, C.c_int, c_dbl_p]),
    "step": (C.c_int, ["H", C.c_int, c_dbl_p, C.c_double, C.c_double, C.c_uint64]),
    "minimize": (C.c_int, ["H", C.c_double, C.c_int, c_int_p, c_dbl_p]),
    "step_count": (C.c_int64, ["H"]),
    "set_step_count": (C.c_int, ["H", C.c_int64]),
    "neighbors": (C.c_int, ["H", C.c_int, C.c_int, C.c_int, C.c_double, c_int_p, C.c_int, c_int_p]),
    "last_kernel_ms": (C.c_int, ["H", c_dbl_p]),
    "synchronize": (C.c_int, ["H"]),
    "last_error": (C.c_char_p, []),
    "version": (C.c_char_p, []),
}
# entry points that exist only in libawsem_hip.so (declared in include/awsem_hip.h, no oracle counterpart)
_HIP_ONLY = {
    "energy_by_group_noalloc": (C.c_int, ["H", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    # replica exchange with the bookkeeping on the device: about device pointers, nothing for the CPU oracle to mirror (its
    # checker is the host-side sweep of openawsem_amd/ensemble.py)
    "remd_setup": (C.c_int, ["H", C.c_int, c_dbl_p, c_int_p, C.c_int, C.c_uint64]),
    "step_remd": (C.c_int, ["H", C.c_int, C.c_double, C.c_double, C.c_uint64]),
    "remd_energies": (C.c_int, ["H", C.c_uint32, C.c_void_p]),
    "remd_sweep": (C.c_int, ["H", C.c_void_p, C.c_int64]),
    "remd_state": (C.c_int, ["H", c_int_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), c_dbl_p]),
    "remd_set_mixing": (C.c_int, ["H", C.c_int]),
    # several temperature stages in one launch: for the oracle the same thing IS the sequence of step calls (Context.step_stages)
    "step_stages": (C.c_int, ["H", C.c_int, C.c_int, c_dbl_p, C.c_double, C.c_double, C.c_uint64]),
    # asynchronous snapshots for reporters: pinned host memory and stream events, nothing for the CPU oracle to mirror
    "snapshot_begin": (C.c_int, ["H", C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_int]),
    "snapshot_end": (C.c_int, ["H", C.c_int, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p, c_dbl_p]),
}

EXPORTED_SYMBOLS = tuple("awsem_" + n for n in list(_SIGNATURES) + list(_HIP_ONLY))


LIST_EXCL, LIST_CONTACT, LIST_BETA, LIST_PAP = 0, 1, 2, 3


class AwsemError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[awsem error {code}] {message}")
        self.code = code


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _ip(a):
    return a.ctypes.data_as(c_int_p)


def _dp(a):
    return None if a is None else a.ctypes.data_as(c_dbl_p)


class CApi:
    """Typed access to one library implementing the include/awsem_hip.h ABI under `prefix`."""

    def __init__(self, lib, prefix="awsem_", extra=None):
        self.lib = lib
        self.prefix = prefix
        self.fn = {}
        sigs = dict(_SIGNATURES)
        if extra:
            sigs.update(extra)
        for name, (res, args) in sigs.items():
            f = getattr(lib, prefix + name)
            f.restype = res
            f.argtypes = [C.c_void_p if a == "H" else a for a in args]
            self.fn[name] = f

    def check(self, code):
        if code != 0:
            msg = self.fn["last_error"]()
            raise AwsemError(code, msg.decode() if msg else "unknown error")

    def version(self):
        return self.fn["version"]().decode()


class Context:
    """One engine handle: a system of `natoms` particles x `n_replicas` trajectories."""

    def __init__(self, api: CApi, natoms: int, n_replicas: int = 1, device: int = 0, precision: int = 0):
        self.api = api
        self.natoms = int(natoms)
        self.n_replicas = int(n_replicas)
        self.precision = precision
        h = C.c_void_p()
        api.check(api.fn["create"](self.natoms, self.n_replicas, int(device), int(precision), C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.api.fn["destroy"](self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    _READ_ONLY = frozenset(("get_positions", "get_velocities", "get_state", "kinetic_energy", "energy", "energy_by_group",
                            "forces", "neighbors", "synchronize", "snapshot_begin", "snapshot_end"))

    def _call(self, name, *args):
        if name not in self._READ_ONLY:
            self._memo = {}          # any other call may change the state the memoised queries were computed from
        self.api.check(self.api.fn[name](self.h, *args))

    # ---- construction
    def set_masses(self, mass):
        m = _f64(mass)
        self._call("set_masses", _dp(m), len(m))

    def set_cm_motion_removal(self, enabled=True):
        self._call("set_cm_motion_removal", int(bool(enabled)))

    def set_virtual_sites(self, site, parents, weights):
        s, p, w = _i32(site), _i32(parents).reshape(-1), _f64(weights).reshape(-1)
        self._call("set_virtual_sites", len(s), _ip(s), _ip(p), _dp(w))

    def add_harmonic_bonds(self, group, pairs, r0, k):
        p, r0, k = _i32(pairs).reshape(-1), _f64(r0), _f64(k)
        self._call("add_harmonic_bonds", group, len(r0), _ip(p), _dp(r0), _dp(k))

    def add_chi(self, group, atoms4, k_chi, chi0):
        a = _i32(atoms4).reshape(-1)
        self._call("add_chi", group, len(a) // 4, _ip(a), float(k_chi), float(chi0))

    def add_rama(self, group, atoms5, well_params, bond_well_weight, k):
        a = _i32(atoms5).reshape(-1)
        wp = _f64(well_params).reshape(-1, 6)
        bw = None if bond_well_weight is None else _f64(bond_well_weight).reshape(-1)
        self._call("add_rama", group, len(a) // 5, _ip(a), wp.shape[0], _dp(wp), _dp(bw), float(k))

    def add_excl(self, group, atom, cls, excl_pairs, k, r_excl):
        a, c, e = _i32(atom), _i32(cls), _i32(excl_pairs).reshape(-1)
        self._call("add_excl", group, len(a), _ip(a), _ip(c), len(e) // 2, _ip(e), float(k), float(r_excl))

    def add_contact(self, group, atom, restype, resid, chain, env, min_seq_sep, flags, dparams, gamma, burial_gamma):
        a, t, r, ch = _i32(atom), _i32(restype), _i32(resid), _i32(chain)
        dp = _f64(dparams)
        g = None if gamma is None else _f64(gamma).reshape(-1)
        bg = None if burial_gamma is None else _f64(burial_gamma).reshape(-1)
        n = len(a)
        if int(flags) & 16:      # index based: tables by site
            assert g is not None and len(g) == (6 if int(flags) & 1 else 3) * n * n and (bg is None or len(bg) == 3 * n)   # z-dependent: water and membrane sets
        else:
            assert (g is None or len(g) == 2400) and (bg is None or len(bg) == (120 if int(flags) & 8 else 60))
        assert len(dp) == 20
        self._call("add_contact", group, len(a), _ip(a), _ip(t), _ip(r), _ip(ch), int(env), int(min_seq_sep), int(flags),
                   _dp(dp), _dp(g), _dp(bg))

    def add_hbond_beta(self, group, two_theta, acc_atoms, acc_res, don_atoms, don_res, lam, k, r_ON, sigma_NO, r_OH,
                       sigma_HO, cutoff):
        aa, ar = _i32(acc_atoms).reshape(-1), _i32(acc_res)
        da, dr = _i32(don_atoms).reshape(-1), _i32(don_res)
        lam = _f64(lam)
        assert lam.ndim == 2 and lam.shape[0] == lam.shape[1]
        self._call("add_hbond_beta", group, int(two_theta), len(ar), _ip(aa), _ip(ar), len(dr), _ip(da), _ip(dr),
                   lam.shape[0], _dp(lam), float(k), float(r_ON), float(sigma_NO), float(r_OH), float(sigma_HO), float(cutoff))

    def add_hbond_pap(self, group, acc_atoms, acc_res, don_atoms, don_res, gamma, k, eta, r0, d_i4, cutoff):
        aa, ar = _i32(acc_atoms).reshape(-1), _i32(acc_res)
        da, dr = _i32(don_atoms).reshape(-1), _i32(don_res)
        g = _f64(gamma)
        assert g.ndim == 2 and g.shape[0] == g.shape[1]
        self._call("add_hbond_pap", group, len(ar), _ip(aa), _ip(ar), len(dr), _ip(da), _ip(dr), g.shape[0], _dp(g),
                   float(k), float(eta), float(r0), float(d_i4), float(cutoff))

    def add_hbond_beta_fn(self, group, two_theta, acc_atoms, acc_res, don_atoms, don_res, nres, kind, res_info, coef, k, r_ON, sigma_NO,
                          r_OH, sigma_HO, cutoff):
        aa, ar = _i32(acc_atoms).reshape(-1), _i32(acc_res)
        da, dr = _i32(don_atoms).reshape(-1), _i32(don_res)
        info, cf = _i32(res_info).reshape(-1), _f64(coef).reshape(-1)
        assert len(info) == 4 * nres and len(cf) == 838
        self._call("add_hbond_beta_fn", group, int(two_theta), len(ar), _ip(aa), _ip(ar), len(dr), _ip(da), _ip(dr), int(nres), int(kind),
                   _ip(info), _dp(cf), float(k), float(r_ON), float(sigma_NO), float(r_OH), float(sigma_HO), float(cutoff))

    def add_hbond_pap_fn(self, group, acc_atoms, acc_res, don_atoms, don_res, nres, kind, res_info, coef, k, eta, r0, d_i4, cutoff):
        aa, ar = _i32(acc_atoms).reshape(-1), _i32(acc_res)
        da, dr = _i32(don_atoms).reshape(-1), _i32(don_res)
        info, cf = _i32(res_info).reshape(-1), _f64(coef).reshape(-1)
        assert len(info) == 4 * nres and len(cf) == 838
        self._call("add_hbond_pap_fn", group, len(ar), _ip(aa), _ip(ar), len(dr), _ip(da), _ip(dr), int(nres), int(kind), _ip(info), _dp(cf),
                   float(k), float(eta), float(r0), float(d_i4), float(cutoff))

    def add_helical(self, group, atoms3, coef, r_ON, sigma_NO, r_OH, sigma_HO):
        a, c = _i32(atoms3).reshape(-1), _f64(coef)
        self._call("add_helical", group, len(c), _ip(a), _dp(c), float(r_ON), float(sigma_NO), float(r_OH), float(sigma_HO))

    def add_helical_z(self, group, atoms4, coef, coef_membrane, r_ON, sigma_NO, r_OH, sigma_HO, eta_switching, z_m, membrane_center):
        a, c, cm = _i32(atoms4).reshape(-1), _f64(coef), _f64(coef_membrane)
        assert len(a) == 4 * len(c) and len(cm) == len(c)
        self._call("add_helical_z", group, len(c), _ip(a), _dp(c), _dp(cm), float(r_ON), float(sigma_NO), float(r_OH), float(sigma_HO),
                   float(eta_switching), float(z_m), float(membrane_center))

    def add_fragment_memory(self, group, pairs, table, k_fm, dr):
        p, t = _i32(pairs).reshape(-1), _f64(table)
        nb = len(p) // 2
        t = t.reshape(nb, -1) if nb else t.reshape(0, max(2, t.shape[-1] if t.ndim else 2))
        self._call("add_fragment_memory", group, nb, _ip(p), t.shape[1], _dp(t), float(k_fm), float(dr))

    def add_external(self, group, kind, atoms, param, k, center=0.0):
        a, s = _i32(atoms), _f64(param)
        assert len(a) == len(s)
        self._call("add_external", group, int(kind), len(a), _ip(a), _dp(s), float(k), float(center))

    def add_membrane(self, group, atoms, scale, k, k_m, z_m, center):
        a, s = _i32(atoms), _f64(scale)
        self._call("add_membrane", group, len(a), _ip(a), _dp(s), float(k), float(k_m), float(z_m), float(center))

    def add_debye_huckel(self, group, pairs, qq, k_dh, screening_length):
        p, q = _i32(pairs).reshape(-1), _f64(qq)
        self._call("add_debye_huckel", group, len(q), _ip(p), _dp(q), float(k_dh), float(screening_length))

    def add_gaussian_bonds(self, group, pairs, r0, sigma, kgamma):
        p, a, b, c = _i32(pairs).reshape(-1), _f64(r0), _f64(sigma), _f64(kgamma)
        self._call("add_gaussian_bonds", group, len(a), _ip(p), _dp(a), _dp(b), _dp(c))

    def add_bond_sum(self, group, bond_kind, pairs, r0, sigma, coef, cv_kind=0, cv_k=0.0, cv_x0=0.0):
        """S = sum of coef*exp(-(r-r0)^2/(2 sigma^2)) (bond_kind 0) or coef*r^2 (1); E = S, k/2 (S-x0)^2, k sqrt(S) or
        k (sqrt(S)-x0)^2 for cv_kind 0..3 (include/awsem_hip.h: awsem_add_bond_sum)."""
        p, c = _i32(pairs).reshape(-1), _f64(coef)
        a = None if r0 is None else _f64(r0)
        b = None if sigma is None else _f64(sigma)
        self._call("add_bond_sum", group, int(bond_kind), len(c), _ip(p), _dp(a), _dp(b), _dp(c), int(cv_kind), float(cv_k), float(cv_x0))

    def finalize(self):
        self._call("finalize")

    # ---- state
    def set_positions(self, xyz, replica=-1):
        x = _f64(xyz).reshape(-1, 3)
        self._call("set_positions", int(replica), _dp(x), x.shape[0])

    def get_positions(self, replica=0):
        x = np.empty((self.natoms, 3))
        self._call("get_positions", int(replica), _dp(x), self.natoms)
        return x

    def set_velocities(self, v, replica=-1):
        v = _f64(v).reshape(-1, 3)
        self._call("set_velocities", int(replica), _dp(v), v.shape[0])

    def get_velocities(self, replica=0):
        v = np.empty((self.natoms, 3))
        self._call("get_velocities", int(replica), _dp(v), self.natoms)
        return v

    def get_state(self, positions=True, velocities=True):
        """(positions, velocities) of all replicas, each [n_replicas, natoms, 3] (None when not requested)."""
        x = np.empty((self.n_replicas, self.natoms, 3)) if positions else None
        v = np.empty((self.n_replicas, self.natoms, 3)) if velocities else None
        self._call("get_state", _dp(x) if positions else None, _dp(v) if velocities else None)
        return x, v

    def set_state(self, positions=None, velocities=None):
        x = None if positions is None else _f64(positions).reshape(self.n_replicas, self.natoms, 3)
        v = None if velocities is None else _f64(velocities).reshape(self.n_replicas, self.natoms, 3)
        self._call("set_state", None if x is None else _dp(x), None if v is None else _dp(v))

    def kinetic_energy(self):
        memo = self.__dict__.setdefault("_memo", {})
        if "ke" not in memo:         # several reporters ask at the same step: one device reduction serves them all
            ke = np.empty(self.n_replicas)
            self._call("kinetic_energy", _dp(ke))
            memo["ke"] = ke
        return memo["ke"].copy()

    def compute_virtual_sites(self):
        self._call("compute_virtual_sites")

    # ---- asynchronous snapshots (include/awsem_hip.h: awsem_snapshot_begin / _end): what the reporters of a run read, fetched behind
    # the launches already queued and collected after the NEXT launch has been queued, so the GPU does not idle while files are written
    def snapshot_begin(self, slot, group_mask=0xFFFFFFFF, energy_replica=-1, replica=0, with_state=False):
        self._call("snapshot_begin", int(slot), C.c_uint32(group_mask & 0xFFFFFFFF), int(energy_replica), int(replica), int(bool(with_state)))

    def snapshot_end(self, slot, energy=True, replica_positions=True, state=False):
        """-> dict(energy [R], kinetic [R], positions [natoms, 3] or None, state (x, v) or None)"""
        e = np.empty(self.n_replicas) if energy else None
        ke = np.empty(self.n_replicas)
        x1 = np.empty((self.natoms, 3)) if replica_positions else None
        xs = np.empty((self.n_replicas, self.natoms, 3)) if state else None
        vs = np.empty((self.n_replicas, self.natoms, 3)) if state else None
        self._call("snapshot_end", int(slot), None if e is None else _dp(e), _dp(ke), None if x1 is None else _dp(x1),
                   None if xs is None else _dp(xs), None if vs is None else _dp(vs))
        return {"energy": e, "kinetic": ke, "positions": x1, "state": (xs, vs) if state else None}

    def set_velocities_to_temperature(self, T, seed):
        T = np.broadcast_to(_f64(T), (self.n_replicas,)).copy()
        self._call("set_velocities_to_temperature", _dp(T), C.c_uint64(int(seed)))

    def scale_velocities(self, scale):
        sc = np.broadcast_to(_f64(scale), (self.n_replicas,)).copy()
        self._call("scale_velocities", _dp(sc))

    # ---- evaluation
    def energy(self, group_mask=0xFFFFFFFF):
        memo = self.__dict__.setdefault("_memo", {})
        key = ("e", group_mask & 0xFFFFFFFF)
        if key not in memo:
            e = np.empty(self.n_replicas)
            self._call("energy", C.c_uint32(group_mask & 0xFFFFFFFF), _dp(e))
            memo[key] = e
        return memo[key].copy()

    def energy_by_group(self):
        e = np.empty((self.n_replicas, 32))
        self._call("energy_by_group", _dp(e))
        return e

    def forces(self, group_mask=0xFFFFFFFF, redistribute=True):
        f = np.empty((self.n_replicas, self.natoms, 3))
        self._call("forces", C.c_uint32(group_mask & 0xFFFFFFFF), int(bool(redistribute)), _dp(f))
        return f

    def step(self, nsteps, T, dt_ps, friction_per_ps, seed):
        T = np.broadcast_to(_f64(T), (self.n_replicas,)).copy()
        self._call("step", int(nsteps), _dp(T), float(dt_ps), float(friction_per_ps), C.c_uint64(int(seed)))

    def step_stages(self, steps_per_stage, T_stages, dt_ps, friction_per_ps, seed):
        """T_stages[stage][replica] (or [stage]): the stages of an annealing loop in one launch (awsem_step_stages); same trajectory as
        one step() per stage, which is what the oracle library runs."""
        T = _f64(T_stages)
        if T.ndim == 1:
            T = T[:, None]
        T = np.ascontiguousarray(np.broadcast_to(T, (T.shape[0], self.n_replicas)))
        if "step_stages" not in self.api.fn:
            for row in T:
                self.step(steps_per_stage, row, dt_ps, friction_per_ps, seed)
            return
        self._call("step_stages", int(T.shape[0]), int(steps_per_stage), _dp(T), float(dt_ps), float(friction_per_ps), C.c_uint64(int(seed)))

    def minimize(self, tolerance=10.0, max_iterations=0):
        """FIRE on the device for every replica at once (awsem_minimize); returns (iterations, rms_force) per replica."""
        it = np.zeros(self.n_replicas, dtype=np.int32)
        rms = np.zeros(self.n_replicas)
        self._call("minimize", float(tolerance), int(max_iterations), it.ctypes.data_as(c_int_p), _dp(rms))
        return it, rms

    @property
    def step_count(self):
        return int(self.api.fn["step_count"](self.h))

    @step_count.setter
    def step_count(self, v):
        self._call("set_step_count", C.c_int64(int(v)))

    def neighbors(self, kind, index=0, replica=0, skin=0.0, capacity=4096):
        """Sorted pair set (n x 2 int32) of one pair term; kind in LIST_EXCL / LIST_CONTACT / LIST_BETA / LIST_PAP.  `capacity`:
        first guess of the pair count (a second call follows when it was too small)."""
        n = C.c_int32(0)
        cap = int(capacity)
        while True:
            pairs = np.empty((cap, 2), dtype=np.int32)
            self._call("neighbors", int(replica), int(kind), int(index), float(skin), _ip(pairs), cap, C.byref(n))
            if n.value <= cap:
                return pairs[:n.value].copy()
            cap = n.value

    def energy_by_group_noalloc(self, n_replicas, d_xyz, d_energy, stream=None):
        """awsem_energy_by_group_noalloc: device pointers (ints) in and out, asynchronous on `stream` (a hipStream_t as int,
        None = the null stream); allocates nothing -- see include/awsem_hip.h."""
        self.api.check(self.api.fn["energy_by_group_noalloc"](self.h, int(n_replicas), C.c_void_p(int(d_xyz)), C.c_void_p(int(d_energy)),
                                                              C.c_void_p(int(stream)) if stream else None))

    # ---- replica exchange on the device (include/awsem_hip.h: awsem_remd_*)
    def remd_setup(self, ladder_T, local_start=0, gather_index=None, seed=2026):
        T = _f64(ladder_T)
        gi = None if gather_index is None else _i32(gather_index)
        self._remd_n = len(T)
        self._call("remd_setup", len(T), _dp(T), None if gi is None else _ip(gi), int(local_start), C.c_uint64(int(seed)))

    def step_remd(self, nsteps, dt_ps, friction_per_ps, seed):
        self._call("step_remd", int(nsteps), float(dt_ps), float(friction_per_ps), C.c_uint64(int(seed)))

    def remd_energies(self, group_mask=0xFFFFFFFF, d_energy=None):
        self._call("remd_energies", C.c_uint32(group_mask & 0xFFFFFFFF), C.c_void_p(int(d_energy)) if d_energy else None)

    def remd_sweep(self, iteration, d_all_energies=None):
        self._call("remd_sweep", C.c_void_p(int(d_all_energies)) if d_all_energies else None, C.c_int64(int(iteration)))

    def remd_set_mixing(self, scheme):
        """'neighbours' / 0 or 'swap-all' / 1 (openmmtools' default, what scripts/mm_remd.py of the reference runs)."""
        code = {"neighbours": 0, "neighbors": 0, "swap-neighbors": 0, "swap-all": 1}.get(scheme, scheme)
        self._call("remd_set_mixing", int(code))

    def remd_state(self):
        """(ladder index of every replica, attempted, accepted per neighbouring pair, temperatures of the local replicas)."""
        n = self._remd_n
        slot = np.zeros(n, dtype=np.int32)
        att, acc = np.zeros(max(n - 1, 0), dtype=np.int64), np.zeros(max(n - 1, 0), dtype=np.int64)
        T = np.zeros(self.n_replicas)
        p64 = C.POINTER(C.c_int64)
        self._call("remd_state", _ip(slot), att.ctypes.data_as(p64), acc.ctypes.data_as(p64), _dp(T))
        return slot, att, acc, T

    def last_kernel_ms(self):
        ms = C.c_double()
        self.api.check(self.api.fn["last_kernel_ms"](self.h, C.byref(ms)))
        return ms.value

    def synchronize(self):
        self._call("synchronize")


_hip_api = None


def _share_torchs_hip_runtime():
    """One HIP runtime per process.  libawsem_hip.so is linked against libamdhip64.so.7; a ROCm build of torch ships its own copy of
    that library (torch/lib, found through torch's RUNPATH).  Whichever is loaded first serves every later user with the same
    SONAME -- except torch, which always maps its own: if the engine came first, the process holds two runtimes and torch's cannot
    see the GPU any more ("No HIP GPUs are available"; measured on the MI355X box, scratch/runtime_order.py), so pointers and
    streams could not be shared with torch.distributed.  Preloading torch's copy (when torch is installed; no `import torch`)
    before dlopen'ing the engine makes the order irrelevant: engine, torch and RCCL share one runtime, one set of streams and
    one address space.  AWSEM_SYSTEM_HIP=1 keeps the system runtime (hosts that never touch torch)."""
    if os.environ.get("AWSEM_SYSTEM_HIP") == "1":
        return None
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    lib = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(lib):
        return None
    try:
        C.CDLL(lib, mode=C.RTLD_GLOBAL)
    except OSError:
        return None
    return lib


def load_hip_library(path: str | None = None) -> CApi:
    """Load libawsem_hip.so (built in-tree by __graft_entry__.build()).  Fails loudly when it is missing:
    there is no CPU fallback in the product path."""
    global _hip_api
    if _hip_api is not None and path is None:
        return _hip_api
    p = path or os.environ.get("AWSEM_HIP_LIBRARY") or HIP_LIBRARY_PATH   # AWSEM_HIP_LIBRARY: an experiment build (tools/try_*.sh)
    if not os.path.exists(p):
        raise ImportError(f"{p} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()')")
    _share_torchs_hip_runtime()
    api = CApi(C.CDLL(p), "awsem_", extra=_HIP_ONLY)
    if path is None:
        _hip_api = api
    return api
