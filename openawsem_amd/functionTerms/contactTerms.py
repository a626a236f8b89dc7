"""Contact (direct + water/protein mediated + burial) and standalone burial terms.

Mirrors contact_term / burial_term of openawsem/functionTerms/contactTerms.py (99-375, 1772-1840); the product is
a `ContactForce` descriptor executed by the HIP engine's three-pass contact kernel (density, pair energy and
burial, chain-rule forces) rather than an OpenMM CustomGBForce.
"""
from __future__ import annotations

import os

import numpy as np

from .. import unit as _u
from ..forces import CONTACT_BURIAL_ONLY, CONTACT_INDEX_BASED, CONTACT_MEMBRANE_BURIAL, CONTACT_NOCUTOFF, CONTACT_ZDEP, ContactForce

gamma_se_map_1_letter = {'A': 0, 'R': 1, 'N': 2, 'D': 3, 'C': 4,
                         'Q': 5, 'E': 6, 'G': 7, 'H': 8, 'I': 9,
                         'L': 10, 'K': 11, 'M': 12, 'F': 13, 'P': 14,
                         'S': 15, 'T': 16, 'W': 17, 'Y': 18, 'V': 19}


def read_gamma(gammaFile):
    """First 210 rows: direct gammas (column 0); next 210: mediated (column 0 protein, column 1 water)
    (reference contactTerms.py:62-75)."""
    data = np.loadtxt(gammaFile)
    return data[:210], data[210:]


def inWhichChain(residueId, chain_ends):
    """Chain letter of a global residue index (reference contactTerms.py:78-96)."""
    chain_table = 'ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789'
    for i, end in enumerate(chain_ends):
        if end >= residueId:
            return chain_table[i]


def _chain_index(residueId, chain_ends):
    for i, end in enumerate(chain_ends):
        if end >= residueId:
            return i
    raise ValueError(f"residue {residueId} is beyond the last chain end")


def _symmetric_from_upper(column):
    """20x20 symmetric table from the 210 upper-triangular values in i<=j order (reference contactTerms.py:182-198)."""
    g = np.zeros((20, 20))
    count = 0
    for i in range(20):
        for j in range(i, 20):
            g[i][j] = g[j][i] = column[count]
            count += 1
    return g


def _gamma_tables(path, scale=1.0):
    direct, mediated = read_gamma(path)
    return np.stack([_symmetric_from_upper(direct[:, 0]),      # direct
                     _symmetric_from_upper(mediated[:, 0]),    # protein mediated
                     _symmetric_from_upper(mediated[:, 1])]) * scale   # water mediated


def _sites(oa):
    """The CB-or-CA(Gly) site of every residue (reference contactTerms.py:266: ``x if x > 0 else y``)."""
    cb_fixed = [x if x > 0 else y for x, y in zip(oa.cb, oa.ca)]
    restype = [gamma_se_map_1_letter[oa.seq[oa.resi[a]]] for a in cb_fixed]
    resid = [oa.resi[a] for a in cb_fixed]
    chain = [_chain_index(r, oa.chain_ends) for r in resid]
    return cb_fixed, restype, resid, chain


def _default_parameters():
    from .. import data_path
    return data_path.parameters


def contact_term(oa, k_contact=4.184, z_dependent=False, z_m=1.5, inMembrane=False, membrane_center=0 * _u.angstrom,
                 k_relative_mem=1.0, periodic=False, parametersLocation=None, burialPartOn=True, withExclusion=False,
                 forceGroup=22, gammaName="gamma.dat", burialGammaName="burial_gamma.dat",
                 membraneGammaName="membrane_gamma.dat", r_min=0.45):
    """Direct + mediated contact with the density-dependent water/protein switch and the burial wells
    (reference contactTerms.py:99-375).  `withExclusion` only changes OpenMM's pair enumeration, not the result."""
    if periodic or getattr(oa, "periodic", False):
        raise NotImplementedError("periodic contact_term is not supported by the MI355X AWSEM engine")
    if parametersLocation is None:
        parametersLocation = _default_parameters()
    if isinstance(k_contact, (float, int)):
        k_contact = k_contact * oa.k_awsem
    elif _u.is_quantity(k_contact):
        k_contact = k_contact.value_in_unit(_u.kilojoule_per_mole) * oa.k_awsem
    else:
        raise TypeError(f"Unknown input, {k_contact}, {type(k_contact)}")
    membrane_center = membrane_center.value_in_unit(_u.nanometer)

    f = ContactForce()
    gamma = np.zeros((2, 3, 20, 20))
    gamma[0] = _gamma_tables(os.path.join(parametersLocation, gammaName))
    inMembrane = int(inMembrane)
    if z_dependent or inMembrane:
        gamma[1] = _gamma_tables(os.path.join(parametersLocation, membraneGammaName), k_relative_mem)
    f.gamma = gamma
    f.burial_gamma = np.loadtxt(os.path.join(parametersLocation, burialGammaName)) if burialPartOn else None
    f.atom, f.restype, f.resid, f.chain = _sites(oa)
    f.env = inMembrane
    f.min_seq_sep = 10           # both environments use 10 (min_sequence_separation, ..._mem)
    f.flags = CONTACT_ZDEP if z_dependent else 0
    f.cutoff = 1.0               # CustomGBForce default cutoff; the reference never sets one (contactTerms.py:367)
    f.dparams[:] = [k_contact, r_min, .65, .65, .95, 50, 7.0, 2.6, f.cutoff, 4.0,
                    0.0, 3.0, 6.0, 3.0, 6.0, 9.0, 10, z_m, membrane_center, k_contact]
    f.setForceGroup(forceGroup)
    print("Number of atom: ", oa.natoms, "Number of residue: ", len(f.atom))
    print("Contact cutoff ", f.getCutoffDistance())
    print("NonbondedMethod: ", f.getNonbondedMethod())
    return f


def burial_term(oa, k_burial=4.184, fastaFile="FastaFileMissing"):
    """Standalone burial wells on the CB density (reference contactTerms.py:1772-1840): reads ``burial_gamma.dat``
    from the working directory, uses NoCutoff (no setNonbondedMethod call) and force group 17."""
    k_burial *= oa.k_awsem
    f = ContactForce()
    f.burial_gamma = np.loadtxt("burial_gamma.dat")
    f.gamma = None
    f.atom, f.restype, f.resid, f.chain = _sites(oa)
    f.flags = CONTACT_BURIAL_ONLY | CONTACT_NOCUTOFF
    f.dparams[:] = [0.0, .45, .65, .65, .95, 50, 7.0, 2.6, 1.0, 4.0,
                    0.0, 3.0, 6.0, 3.0, 6.0, 9.0, 10, 1.5, 0.0, k_burial]
    f.setForceGroup(17)
    return f


def read_hybrid_gamma(hybrid_gamma_file):
    """The 1380 numbers of a hybrid_contact_gamma.dat, negated (reference contactTerms.py:1155-1184): for each environment
    (water, membrane) the upper triangles (i <= j, row-major) of the direct, protein-mediated and water-mediated 20x20
    tables, then for each environment three burial wells of 20 residue types.  Returns gamma [2][3][20][20] in this
    package's plane order (direct, protein, water) and burial [2][20][3]."""
    g = -np.loadtxt(hybrid_gamma_file).reshape(-1)
    if g.size != 2 * 3 * 210 + 2 * 3 * 20:
        raise ValueError(f"{hybrid_gamma_file}: expected 1380 values, found {g.size}")
    iu = np.triu_indices(20)
    gamma = np.zeros((2, 3, 20, 20))
    c = 0
    for env in range(2):
        for plane in range(3):            # file order direct, protein, water = plane order here
            gamma[env, plane][iu] = g[c:c + 210]
            gamma[env, plane].T[iu] = g[c:c + 210]
            c += 210
    burial = g[c:].reshape(2, 3, 20).transpose(0, 2, 1).copy()
    return gamma, burial


def hybrid_contact_term(oa, k_contact=4.184, z_m=1.5, membrane_center=0 * _u.angstrom, periodic=False,
                        hybrid_gamma_file="hybrid_contact_gamma.dat", water_pair_part=False):
    """z-dependent contact term with one optimised parameter file for both environments (reference contactTerms.py:1110-1311):
    pair energy (1 - a_i a_j) water + a_i a_j membrane, burial wells of site i blended with (1 - a_i, a_i), force group 18.

    As written, the reference rebuilds `res_table` inside its loop over the two environments (contactTerms.py:1187-1198), so
    the table of environment 0 stays all zero and the water part of the pair energy vanishes; a drop-in has to give the same
    numbers, so that is the default here.  `water_pair_part=True` keeps the water tables (the evident intent)."""
    if periodic or getattr(oa, "periodic", False):
        raise NotImplementedError("periodic contact terms are not supported by the MI355X AWSEM engine")
    k_contact = k_contact * oa.k_awsem
    if _u.is_quantity(membrane_center):
        membrane_center = membrane_center.value_in_unit(_u.nanometer)
    f = ContactForce()
    gamma, burial = read_hybrid_gamma(hybrid_gamma_file)
    if not water_pair_part:
        gamma[0] = 0.0
    f.gamma, f.burial_gamma = gamma, burial
    f.atom, f.restype, f.resid, f.chain = _sites(oa)
    f.env = 0
    f.min_seq_sep = 10
    f.flags = CONTACT_ZDEP | CONTACT_MEMBRANE_BURIAL
    f.cutoff = 1.0
    f.dparams[:] = [k_contact, .45, .65, .65, .95, 50, 7.0, 2.6, f.cutoff, 4.0,
                    0.0, 3.0, 6.0, 3.0, 6.0, 9.0, 10, z_m, membrane_center, k_contact]
    f.setForceGroup(18)
    print("Number of atom: ", oa.natoms, "Number of residue: ", len(f.atom))
    print("Contact cutoff ", f.getCutoffDistance())
    print("NonbondedMethod: ", f.getNonbondedMethod())
    return f


def index_based_contact_term(oa, gamma_folder="ff_contact", k_contact=4.184, z_dependent=False, z_m=1.5, inMembrane=False,
                             membrane_center=0 * _u.angstrom, k_relative_mem=1.0, periodic=False, parametersLocation=".",
                             burialPartOn=True, withExclusion=True, r_min=0.45, forceGroup=22):
    """Contact term whose direct / water / protein gammas are nres x nres tables by residue index and whose burial wells are
    per residue (reference contactTerms.py:668-876): `gamma_folder` holds direct.dat, water.dat, protein.dat (nres x nres)
    and burial.dat (nres x 3); with inMembrane the pair tables come from *_mem.dat (766-773).  A pair reads
    gamma[lower residue][upper residue].  z_dependent=True (812-831) blends the water tables (*.dat) and the membrane tables
    (*_mem.dat) of a pair with alpha_i alpha_j, alpha(z) = 0.5 tanh(10 (z - center + z_m)) + 0.5 tanh(10 (z_m - (z - center)));
    the burial wells stay those of burial.dat."""
    if periodic or getattr(oa, "periodic", False):
        raise NotImplementedError("periodic contact terms are not supported by the MI355X AWSEM engine")
    if isinstance(k_contact, (float, int)):
        k_contact = k_contact * oa.k_awsem
    elif _u.is_quantity(k_contact):
        k_contact = k_contact.value_in_unit(_u.kilojoule_per_mole) * oa.k_awsem
    else:
        raise TypeError(f"Unknown input, {k_contact}, {type(k_contact)}")
    if _u.is_quantity(membrane_center):
        membrane_center = membrane_center.value_in_unit(_u.nanometer)
    n = oa.nres
    tables = []
    for suffix in (("", "_mem") if z_dependent else ("_mem",) if inMembrane else ("",)):   # z-dependent: water tables, then membrane tables
        for name in ("direct", "protein", "water"):                 # this package's plane order
            t = np.loadtxt(os.path.join(gamma_folder, f"{name}{suffix}.dat"))
            if t.shape != (n, n):
                raise ValueError(f"{gamma_folder}/{name}{suffix}.dat: expected a {n} x {n} table, found {t.shape}")
            tables.append(t)
    f = ContactForce()
    f.atom, f.restype, f.resid, f.chain = _sites(oa)
    rid = np.asarray(f.resid)
    f.gamma = np.stack([t[np.ix_(rid, rid)] for t in tables])    # by site; site s is residue f.resid[s]
    f.burial_gamma = None
    if burialPartOn:
        b = np.loadtxt(os.path.join(gamma_folder, "burial.dat"))
        if b.shape != (n, 3):
            raise ValueError(f"{gamma_folder}/burial.dat: expected a {n} x 3 table, found {b.shape}")
        f.burial_gamma = b[rid]
    f.env = 0
    f.min_seq_sep = 10
    f.flags = CONTACT_INDEX_BASED | (CONTACT_ZDEP if z_dependent else 0)
    f.cutoff = 1.0
    f.dparams[:] = [k_contact, r_min, .65, .65, .95, 50, 7.0, 2.6, f.cutoff, 4.0,
                    0.0, 3.0, 6.0, 3.0, 6.0, 9.0, 10, z_m, membrane_center, k_contact]
    f.setForceGroup(forceGroup)
    print("Number of atom: ", oa.natoms, "Number of residue: ", len(f.atom))
    print("Contact cutoff ", f.getCutoffDistance())
    print("NonbondedMethod: ", f.getNonbondedMethod())
    return f
