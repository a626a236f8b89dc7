"""Typed term descriptors: what the openawsem-compatible factories return instead of ``openmm.Force`` objects.

The reference factories build OpenMM Custom*Force objects out of expression strings
(openawsem/functionTerms/*.py).  This engine supports the fixed AWSEM term set only, so a factory's product is a
plain record of index arrays and parameter tables plus the two Force methods the drivers rely on
(``setForceGroup``/``getForceGroup``: openawsem/openAWSEM.py:918-928, openawsem/scripts/mm_analyze.py:119-178).
``_upload(ctx)`` hands the arrays to a `_capi.Context` (include/awsem_hip.h).
"""
from __future__ import annotations

import numpy as np


class Force:
    kind = "force"

    def __init__(self, group=0):
        self._group = int(group)

    def setForceGroup(self, group):
        group = int(group)
        if not 0 <= group < 32:
            raise ValueError("Force group must be between 0 and 31")
        self._group = group

    def getForceGroup(self):
        return self._group

    # OpenMM Force API bits a user script may touch; harmless here.
    def setUsesPeriodicBoundaryConditions(self, flag):
        if flag:
            raise NotImplementedError("periodic boundary conditions are not supported by the MI355X AWSEM engine")

    def usesPeriodicBoundaryConditions(self):
        return False

    def _upload(self, ctx):
        raise NotImplementedError


class HarmonicBondForce(Force):
    kind = "harmonic"

    def __init__(self, group=0):
        super().__init__(group)
        self.pairs, self.r0, self.k = [], [], []

    def addBond(self, i, j, r0, k):
        self.pairs.append((int(i), int(j)))
        self.r0.append(float(r0))
        self.k.append(float(k))
        return len(self.r0) - 1

    def getNumBonds(self):
        return len(self.r0)

    def _upload(self, ctx):
        ctx.add_harmonic_bonds(self._group, np.array(self.pairs, dtype=np.int32).reshape(-1, 2), self.r0, self.k)


class ChiForce(Force):
    kind = "chi"

    def __init__(self, k_chi, chi0, group=0):
        super().__init__(group)
        self.k_chi, self.chi0, self.atoms = float(k_chi), float(chi0), []

    def addBond(self, atoms):
        self.atoms.append(tuple(int(a) for a in atoms))

    def getNumBonds(self):
        return len(self.atoms)

    def _upload(self, ctx):
        ctx.add_chi(self._group, np.array(self.atoms, dtype=np.int32).reshape(-1, 4), self.k_chi, self.chi0)


class RamaForce(Force):
    """-k sum_w weight_w(bond) w_w exp(-sigma_w(omega_phi_w (cos(phi-phi0_w)-1)^2 + omega_psi_w (cos(psi-psi0_w)-1)^2))"""
    kind = "rama"

    def __init__(self, k, w, sigma, omega_phi, phi0, omega_psi, psi0, group=0):
        super().__init__(group)
        self.k = float(k)
        self.well_params = np.stack([np.asarray(x, dtype=float) for x in (w, sigma, omega_phi, phi0, omega_psi, psi0)], axis=1)
        self.atoms, self.weights = [], []

    def addBond(self, atoms, weights=None):
        self.atoms.append(tuple(int(a) for a in atoms))
        nw = self.well_params.shape[0]
        self.weights.append([1.0] * nw if weights is None else [float(x) for x in weights])

    def getNumBonds(self):
        return len(self.atoms)

    def _upload(self, ctx):
        ctx.add_rama(self._group, np.array(self.atoms, dtype=np.int32).reshape(-1, 5), self.well_params,
                     np.array(self.weights, dtype=float).reshape(-1, self.well_params.shape[0]), self.k)


class ExclForce(Force):
    kind = "excl"

    def __init__(self, k, r_excl, group=0):
        super().__init__(group)
        self.k, self.r_excl = float(k), float(r_excl)
        self.atom, self.cls, self.excl_pairs = [], [], []

    def _upload(self, ctx):
        ctx.add_excl(self._group, self.atom, self.cls, np.array(self.excl_pairs, dtype=np.int32).reshape(-1, 2), self.k, self.r_excl)

    def getCutoffDistance(self):
        return self.r_excl


CONTACT_ZDEP, CONTACT_BURIAL_ONLY, CONTACT_NOCUTOFF, CONTACT_MEMBRANE_BURIAL, CONTACT_INDEX_BASED = 1, 2, 4, 8, 16


class ContactForce(Force):
    kind = "contact"

    def __init__(self, group=0):
        super().__init__(group)
        self.atom = self.restype = self.resid = self.chain = None
        self.env, self.min_seq_sep, self.flags = 0, 10, 0
        self.dparams = np.zeros(20)
        self.gamma = None          # [2][3][20][20]: env x (direct, protein, water)
        self.burial_gamma = None   # [20][3]; [2][20][3] (water, membrane) with CONTACT_MEMBRANE_BURIAL
        self.cutoff = 1.0

    def getCutoffDistance(self):
        return self.cutoff

    def getNonbondedMethod(self):
        return 0 if self.flags & CONTACT_NOCUTOFF else 1   # NoCutoff / CutoffNonPeriodic

    def _upload(self, ctx):
        ctx.add_contact(self._group, self.atom, self.restype, self.resid, self.chain, self.env, self.min_seq_sep, self.flags,
                        self.dparams, self.gamma, self.burial_gamma)


HB_DENSE_MAX_RESIDUES = 2048   # above this the nres x nres lambda / gamma tables are not built: the functional form is uploaded


class HbTableFn:
    """A lambda / gamma table of the hydrogen-bond terms as a function of the residue pair (include/awsem_hip.h:
    awsem_add_hbond_beta_fn): kind 1..5, res_info [nres][4] (type, chain, chain edge, helix), coef[838] = C[5][3], G[3],
    T1[20][20], T2[20][20], O[20].  `dense()` builds the nres x nres array the reference feeds to OpenMM (the checker, and
    the form small systems upload)."""

    def __init__(self, kind, res_info, class_coef=None, gammas=None, t1=None, t2=None, one_body=None, dense=None):
        self.kind = int(kind)
        self.res_info = np.ascontiguousarray(res_info, dtype=np.int32).reshape(-1, 4)
        c = np.zeros(838)
        if class_coef is not None:
            cc = np.asarray(class_coef, dtype=float)
            c[:3 * len(cc)] = cc.reshape(-1)
        if gammas is not None:
            c[15:18] = gammas
        if t1 is not None:
            c[18:418] = np.asarray(t1, dtype=float).reshape(-1)
        if t2 is not None:
            c[418:818] = np.asarray(t2, dtype=float).reshape(-1)
        if one_body is not None:
            c[818:838] = one_body
        self.coef = c
        self._dense = dense

    @property
    def nres(self):
        return len(self.res_info)

    def dense(self):
        return np.asarray(self._dense(), dtype=float)

    def wants_functional(self):
        import os
        mode = os.environ.get("AWSEM_HB_TABLES", "")
        if mode in ("functional", "dense"):
            return mode == "functional"
        return self.nres > HB_DENSE_MAX_RESIDUES


class _LazyTable:
    """`force.lam` / `force.gamma`: the dense table, built on first use when the force carries its functional form."""

    def __init__(self, name):
        self.name = name

    def __get__(self, obj, owner=None):
        if obj is None:
            return self
        v = obj.__dict__.get(self.name)
        if v is None and obj.fn is not None:
            v = obj.__dict__[self.name] = obj.fn.dense()
        return v

    def __set__(self, obj, value):
        obj.__dict__[self.name] = None if value is None else np.asarray(value, dtype=float)


class HbondBetaForce(Force):
    kind = "beta"
    lam = _LazyTable("lam")

    def __init__(self, two_theta, k, lam=None, r_ON=.298, sigma_NO=.068, r_OH=.206, sigma_HO=.076, cutoff=1.0, group=0, fn=None):
        super().__init__(group)
        self.fn = fn
        self.two_theta, self.k, self.lam = int(two_theta), float(k), lam
        self.r_ON, self.sigma_NO, self.r_OH, self.sigma_HO, self.cutoff = r_ON, sigma_NO, r_OH, sigma_HO, cutoff
        self.acc_atoms, self.acc_res, self.don_atoms, self.don_res = [], [], [], []

    def addAcceptor(self, a1, a2, a3, params):
        self.acc_atoms.append((int(a1), int(a2), int(a3)))
        self.acc_res.append(int(params[0]))

    def addDonor(self, d1, d2, d3, params):
        self.don_atoms.append((int(d1), int(d2), int(d3)))
        self.don_res.append(int(params[0]))

    def getNumAcceptors(self):
        return len(self.acc_res)

    def getNumDonors(self):
        return len(self.don_res)

    def getCutoffDistance(self):
        return self.cutoff

    def _upload(self, ctx):
        acc, don = np.array(self.acc_atoms, dtype=np.int32).reshape(-1, 3), np.array(self.don_atoms, dtype=np.int32).reshape(-1, 3)
        if self.fn is not None and self.fn.wants_functional():
            ctx.add_hbond_beta_fn(self._group, self.two_theta, acc, self.acc_res, don, self.don_res, self.fn.nres, self.fn.kind,
                                  self.fn.res_info, self.fn.coef, self.k, self.r_ON, self.sigma_NO, self.r_OH, self.sigma_HO, self.cutoff)
        else:
            ctx.add_hbond_beta(self._group, self.two_theta, acc, self.acc_res, don, self.don_res, self.lam, self.k,
                               self.r_ON, self.sigma_NO, self.r_OH, self.sigma_HO, self.cutoff)


class HbondPapForce(Force):
    kind = "pap"
    gamma = _LazyTable("gamma")

    def __init__(self, k, gamma=None, eta=70.0, r0=0.8, d_i4=1.2, cutoff=1.0, group=0, fn=None):
        super().__init__(group)
        self.fn = fn
        self.k, self.gamma = float(k), gamma
        self.eta, self.r0, self.d_i4, self.cutoff = float(eta), float(r0), float(d_i4), float(cutoff)
        self.acc_atoms, self.acc_res, self.don_atoms, self.don_res = [], [], [], []

    def addAcceptor(self, a1, a2, a3, params):
        self.acc_atoms.append((int(a1), int(a2)))
        self.acc_res.append(int(params[0]))

    def addDonor(self, d1, d2, d3, params):
        self.don_atoms.append((int(d1), int(d2)))
        self.don_res.append(int(params[0]))

    def getNumAcceptors(self):
        return len(self.acc_res)

    def getNumDonors(self):
        return len(self.don_res)

    def getCutoffDistance(self):
        return self.cutoff

    def _upload(self, ctx):
        acc, don = np.array(self.acc_atoms, dtype=np.int32).reshape(-1, 2), np.array(self.don_atoms, dtype=np.int32).reshape(-1, 2)
        if self.fn is not None and self.fn.wants_functional():
            ctx.add_hbond_pap_fn(self._group, acc, self.acc_res, don, self.don_res, self.fn.nres, self.fn.kind, self.fn.res_info,
                                 self.fn.coef, self.k, self.eta, self.r0, self.d_i4, self.cutoff)
        else:
            ctx.add_hbond_pap(self._group, acc, self.acc_res, don, self.don_res, self.gamma, self.k,
                              self.eta, self.r0, self.d_i4, self.cutoff)


class HelicalForce(Force):
    kind = "helical"

    def __init__(self, r_ON=.298, sigma_NO=.068, r_OH=.206, sigma_HO=.076, group=0):
        super().__init__(group)
        self.r_ON, self.sigma_NO, self.r_OH, self.sigma_HO = r_ON, sigma_NO, r_OH, sigma_HO
        self.atoms, self.coef = [], []

        # z_dependent_helical_term: every bond carries a fourth atom (CA_i) and a membrane coefficient
        self.coef_membrane, self.eta_switching, self.z_m, self.membrane_center = None, 10.0, 1.5, 0.0

    def addBond(self, atoms, coef, coef_membrane=None):
        self.atoms.append(tuple(int(a) for a in atoms))
        self.coef.append(float(coef))
        if coef_membrane is not None:
            if self.coef_membrane is None:
                self.coef_membrane = []
            self.coef_membrane.append(float(coef_membrane))

    def getNumBonds(self):
        return len(self.coef)

    def _upload(self, ctx):
        if self.coef_membrane is not None:
            assert len(self.coef_membrane) == len(self.coef)
            ctx.add_helical_z(self._group, np.array(self.atoms, dtype=np.int32).reshape(-1, 4), self.coef, self.coef_membrane,
                              self.r_ON, self.sigma_NO, self.r_OH, self.sigma_HO, self.eta_switching, self.z_m, self.membrane_center)
            return
        ctx.add_helical(self._group, np.array(self.atoms, dtype=np.int32).reshape(-1, 3), self.coef, self.r_ON, self.sigma_NO,
                        self.r_OH, self.sigma_HO)


class FragmentMemoryForce(Force):
    kind = "fragment_memory"

    def __init__(self, k_fm, pairs, table, dr, group=0):
        super().__init__(group)
        self.k_fm, self.dr = float(k_fm), float(dr)
        self.pairs = np.asarray(pairs, dtype=np.int32).reshape(-1, 2)
        self.table = np.asarray(table, dtype=float)

    def getNumBonds(self):
        return len(self.pairs)

    def _upload(self, ctx):
        ctx.add_fragment_memory(self._group, self.pairs, self.table, self.k_fm, self.dr)


class MembraneForce(Force):
    kind = "membrane"

    def __init__(self, k, k_m, z_m, center, group=0):
        super().__init__(group)
        self.k, self.k_m, self.z_m, self.center = float(k), float(k_m), float(z_m), float(center)
        self.atoms, self.scale = [], []

    def addParticle(self, atom, params):
        self.atoms.append(int(atom))
        self.scale.append(float(params[0]))

    def _upload(self, ctx):
        ctx.add_membrane(self._group, self.atoms, self.scale, self.k, self.k_m, self.z_m, self.center)


EXT_LINEAR_X, EXT_LINEAR_Y, EXT_LINEAR_Z, EXT_ABS_Z = 1, 2, 3, 4


class ExternalForce(Force):
    """One-parameter CustomExternalForce shapes (awsem_add_external): E = -k p_i x|y|z, or k |z - center - p_i|."""
    kind = "external"

    def __init__(self, shape, k, center=0.0, group=0):
        super().__init__(group)
        self.shape, self.k, self.center = int(shape), float(k), float(center)
        self.atoms, self.param = [], []

    def addParticle(self, atom, params=(1.0,)):
        self.atoms.append(int(atom))
        self.param.append(float(params[0]))
        return len(self.atoms) - 1

    def getNumParticles(self):
        return len(self.atoms)

    def _upload(self, ctx):
        ctx.add_external(self._group, self.shape, self.atoms, self.param, self.k, self.center)


class DebyeHuckelForce(Force):
    kind = "debye_huckel"

    def __init__(self, k_dh, screening_length, group=0):
        super().__init__(group)
        self.k_dh, self.screening_length = float(k_dh), float(screening_length)
        self.pairs, self.qq = [], []

    def addBond(self, i, j, params):
        self.pairs.append((int(i), int(j)))
        self.qq.append(float(params[0]) * float(params[1]))

    def getNumBonds(self):
        return len(self.qq)

    def _upload(self, ctx):
        ctx.add_debye_huckel(self._group, np.array(self.pairs, dtype=np.int32).reshape(-1, 2), self.qq, self.k_dh,
                             self.screening_length)


class GaussianBondForce(Force):
    """CustomBondForce("-k*gamma_ij*exp(-(r-r_ijN)^2/(2*sigma_ij^2))") of er_term / additive_amhgo_term
    (templateTerms.py:600-693): per bond (gamma_ij, r_ijN, sigma_ij); k is folded into gamma on upload."""
    kind = "gaussian_bond"

    def __init__(self, k, group=0):
        super().__init__(group)
        self.k = float(k)
        self.pairs, self.gamma, self.r0, self.sigma = [], [], [], []

    def addBond(self, i, j, params):
        self.pairs.append((int(i), int(j)))
        self.gamma.append(float(params[0]))
        r0 = params[1]
        if hasattr(r0, "value_in_unit"):
            from . import unit as u
            r0 = r0.value_in_unit(u.nanometer)
        self.r0.append(float(r0))
        self.sigma.append(float(params[2]))

    def getNumBonds(self):
        return len(self.r0)

    def _upload(self, ctx):
        ctx.add_gaussian_bonds(self._group, np.array(self.pairs, dtype=np.int32).reshape(-1, 2), self.r0, self.sigma,
                               self.k * np.asarray(self.gamma, dtype=np.float64))


class BondSumForce(Force):
    """A CustomBondForce sum S = sum_b g_b(r_b), alone or as the collective variable of a CustomCVForce E = f(S)
    (biasTerms.py: q_value 146-188, qc_value 191-223, partial_q_value 226-267, qbias_term 270-302, rg_term 461-486,
    rg_bias_term 489-528).  bond_kind "gauss": g = coef exp(-(r-r0)^2/(2 sigma^2)); "r2": g = coef r^2.
    cv: "sum" E = S; "harmonic" E = k/2 (S-x0)^2; "sqrt" E = k sqrt(S); "sqrt_square" E = k (sqrt(S)-x0)^2."""
    kind = "bond_sum"
    BOND = {"gauss": 0, "r2": 1, "r2_xy": 2}
    CV = {"sum": 0, "harmonic": 1, "sqrt": 2, "sqrt_square": 3}

    def __init__(self, bond_kind="gauss", cv="sum", k=0.0, x0=0.0, group=0):
        super().__init__(group)
        self.bond_kind, self.cv, self.k, self.x0 = bond_kind, cv, float(k), float(x0)
        self.pairs, self.coef, self.r0, self.sigma = [], [], [], []

    def addBond(self, i, j, params=()):
        """params = [coef, r0, sigma] for Gaussian bonds (the reference's [gamma_ij, r_ijN, sigma_ij]), [coef] or [] for r^2."""
        from . import unit as u
        self.pairs.append((int(i), int(j)))
        self.coef.append(float(params[0]) if len(params) else 1.0)
        if self.bond_kind == "gauss":
            r0 = params[1]
            self.r0.append(float(r0.value_in_unit(u.nanometer) if hasattr(r0, "value_in_unit") else r0))
            self.sigma.append(float(params[2]))

    def getNumBonds(self):
        return len(self.coef)

    def scale(self, factor):
        self.coef = [c * factor for c in self.coef]

    def _upload(self, ctx):
        gauss = self.bond_kind == "gauss"
        ctx.add_bond_sum(self._group, self.BOND[self.bond_kind], np.array(self.pairs, dtype=np.int32).reshape(-1, 2),
                         self.r0 if gauss else None, self.sigma if gauss else None, self.coef, self.CV[self.cv], self.k, self.x0)


class System:
    """Light stand-in for openmm.System: particle masses, virtual sites and the list of forces."""

    def __init__(self, masses, remove_cm_motion=False):
        self.masses = np.asarray(masses, dtype=float)
        # ForceField.createSystem(topology) adds a CMMotionRemover by default (removeCMMotion=True); a bare openmm.System has none
        self.remove_cm_motion = bool(remove_cm_motion)
        self.virtual_sites = {}   # site atom -> (p1, p2, p3, w1, w2, w3)
        self.forces = []

    def getNumParticles(self):
        return len(self.masses)

    def getParticleMass(self, i):
        return self.masses[i]

    def setVirtualSite(self, index, site):
        self.virtual_sites[int(index)] = site
        self.masses[int(index)] = 0.0

    def isVirtualSite(self, index):
        return int(index) in self.virtual_sites

    def addForce(self, force):
        if not isinstance(force, Force):
            raise TypeError(f"unsupported force object {type(force).__name__}: the MI355X AWSEM engine accepts only the "
                            "term descriptors returned by openawsem_amd.functionTerms")
        self.forces.append(force)
        return len(self.forces) - 1

    def getNumForces(self):
        return len(self.forces)

    def getForce(self, i):
        return self.forces[i]

    def getForces(self):
        return list(self.forces)

    def upload(self, ctx):
        """Push masses, virtual sites and every force to a fresh `_capi.Context`, then finalize it."""
        ctx.set_masses(self.masses)
        ctx.set_cm_motion_removal(self.remove_cm_motion)
        if self.virtual_sites:
            sites = sorted(self.virtual_sites)
            par = [self.virtual_sites[s][:3] for s in sites]
            w = [self.virtual_sites[s][3:] for s in sites]
            ctx.set_virtual_sites(sites, par, w)
        for f in self.forces:
            f._upload(ctx)
        ctx.finalize()
        return ctx


def ThreeParticleAverageSite(p1, p2, p3, w1, w2, w3):
    return (int(p1), int(p2), int(p3), float(w1), float(w2), float(w3))
