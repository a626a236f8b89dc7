"""Backbone terms: con, chain, chi, excl, rama, rama_proline, rama_ssweight.

Same names, arguments and defaults as openawsem/functionTerms/basicTerms.py; each returns a typed descriptor
(openawsem_amd.forces) that the HIP engine executes instead of an OpenMM expression force.
"""
from __future__ import annotations

import numpy as np

from ..forces import ChiForce, ExclForce, HarmonicBondForce, RamaForce


def _no_periodic(oa):
    if getattr(oa, "periodic", False):
        raise NotImplementedError("periodic systems are not supported by the MI355X AWSEM engine")


def con_term(oa, k_con=50208, bond_lengths=[.3816, .240, .276, .153], forceGroup=20):
    """CA-O, CA-CB, CA-CA(+1), O-CA(+1) springs, E = 1/2 k (r-r0)^2 (reference basicTerms.py:24-58)."""
    _no_periodic(oa)
    k_con *= oa.k_awsem
    con = HarmonicBondForce()
    for i in range(oa.nres):
        con.addBond(oa.ca[i], oa.o[i], bond_lengths[1], k_con)
        if not oa.res_type[i] == "IGL":
            con.addBond(oa.ca[i], oa.cb[i], bond_lengths[3], k_con)
        if i not in oa.chain_ends:
            con.addBond(oa.ca[i], oa.ca[i + 1], bond_lengths[0], k_con)
            con.addBond(oa.o[i], oa.ca[i + 1], bond_lengths[2], k_con)
    con.setForceGroup(forceGroup)
    return con


def chain_term(oa, k_chain=50208, bond_k=[1, 1, 1], bond_lengths=[0.2459108, 0.2519591, 0.2466597], forceGroup=20):
    """N-CB, C-CB, N-C springs (reference basicTerms.py:61-95)."""
    _no_periodic(oa)
    k_chain *= oa.k_awsem
    chain = HarmonicBondForce()
    for i in range(oa.nres):
        gly = oa.res_type[i] == "IGL"
        if i not in oa.chain_starts and not gly:
            chain.addBond(oa.n[i], oa.cb[i], bond_lengths[0], k_chain * bond_k[0])
        if i not in oa.chain_ends and not gly:
            chain.addBond(oa.c[i], oa.cb[i], bond_lengths[1], k_chain * bond_k[1])
        if i not in oa.chain_starts and i not in oa.chain_ends:
            chain.addBond(oa.n[i], oa.c[i], bond_lengths[2], k_chain * bond_k[2])
    chain.setForceGroup(forceGroup)
    return chain


def chain_no_cb_constraint_term(oa, k_chain=50208, bond_lengths=[0.2459108, 0.2519591, 0.2466597], forceGroup=20):
    """chain_term without the springs that hold CB: only N-C of the interior residues (reference basicTerms.py:542-574)."""
    _no_periodic(oa)
    k_chain *= oa.k_awsem
    chain = HarmonicBondForce()
    for i in range(oa.nres):
        if i not in oa.chain_starts and i not in oa.chain_ends:
            chain.addBond(oa.n[i], oa.c[i], bond_lengths[2], k_chain)
    chain.setForceGroup(forceGroup)
    return chain


def con_no_cb_constraint_term(oa, k_con=50208, bond_lengths=[.3816, .240, .276, .153], forceGroup=20):
    """con_term with the CA-CB spring kept only at the chain ends, where no N / C site holds CB (reference
    basicTerms.py:577-615)."""
    _no_periodic(oa)
    k_con *= oa.k_awsem
    con = HarmonicBondForce()
    for i in range(oa.nres):
        con.addBond(oa.ca[i], oa.o[i], bond_lengths[1], k_con)
        if ((i in oa.chain_starts) or (i in oa.chain_ends)) and (not oa.res_type[i] == "IGL"):
            con.addBond(oa.ca[i], oa.cb[i], bond_lengths[3], k_con)
        if i not in oa.chain_ends:
            con.addBond(oa.ca[i], oa.ca[i + 1], bond_lengths[0], k_con)
            con.addBond(oa.o[i], oa.ca[i + 1], bond_lengths[2], k_con)
    con.setForceGroup(forceGroup)
    return con


def chi_term(oa, k_chi=251.04, chi0=-0.71, forceGroup=20):
    """Chirality of (CA, C, N, CB): E = k (chi - chi0)^2 (reference basicTerms.py:98-138)."""
    _no_periodic(oa)
    k_chi *= oa.k_awsem
    chi = ChiForce(k_chi, chi0)
    for i in range(oa.nres):
        if i not in oa.chain_starts and i not in oa.chain_ends and not oa.res_type[i] == "IGL":
            chi.addBond([oa.ca[i], oa.c[i], oa.n[i], oa.cb[i]])
    chi.setForceGroup(forceGroup)
    return chi


def excl_term(oa, k_excl=8368, r_excl=0.35, periodic=False, excludeCB=False, forceGroup=20):
    """Excluded volume k (r - r_excl)^2 for r < r_excl between CA/CB sites and between O sites, minus directly
    bonded pairs (reference basicTerms.py:141-191).  As in the reference the `periodic` argument is ignored in
    favour of oa.periodic.  The reference's interaction groups are (CA,CA), (CB,CB), (CA,CB), (O,O): a pair
    interacts iff both atoms are CA/CB (class 0) or both are O (class 1); with excludeCB the (CB,CB) group is
    dropped, which the class model cannot express."""
    _no_periodic(oa)
    if excludeCB:
        raise NotImplementedError("excl_term(excludeCB=True) is not supported by the MI355X AWSEM engine")
    k_excl *= oa.k_awsem
    excl = ExclForce(k_excl, r_excl)
    cls = {}
    for a in oa.ca:
        cls[a] = 0
    for a in oa.cb:
        if a > 0:
            cls[a] = 0
    for a in oa.o:
        cls[a] = 1
    excl.atom = sorted(cls)
    excl.cls = [cls[a] for a in excl.atom]
    # createExclusionsFromBonds(oa.bonds, 1): only pairs that can interact matter
    pairs = set()
    for a, b in oa.bonds:
        if a in cls and b in cls and cls[a] == cls[b]:
            pairs.add((min(a, b), max(a, b)))
    excl.excl_pairs = sorted(pairs)
    excl.setForceGroup(forceGroup)
    return excl


def excl_term_v2(oa, k_excl=8368, r_excl=0.35, periodic=False, excludeCB=False, forceGroup=20):
    """excl_term without createExclusionsFromBonds (reference basicTerms.py:194-263): a pair of the same interaction groups
    counts iff step(|res1-res2| - 2 + isChainEdge1*isChainEdge2 + isnot_Ca1 + isnot_Ca2) = 1, i.e. CA-CA pairs of
    neighbouring residue indices and the CA-CB pair of one residue are left out.  isChainEdge is computed as the reference
    does (the ATOM index is looked up in the lists of chain-end RESIDUE indices, basicTerms.py:237)."""
    _no_periodic(oa)
    if excludeCB:
        raise NotImplementedError("excl_term_v2(excludeCB=True) is not supported by the MI355X AWSEM engine")
    k_excl *= oa.k_awsem
    excl = ExclForce(k_excl, r_excl)
    ca, cb, o = list(oa.ca), [x for x in oa.cb if x > 0], list(oa.o)
    cls = {a: 0 for a in ca + cb}
    cls.update({a: 1 for a in o})
    excl.atom = sorted(cls)
    excl.cls = [cls[a] for a in excl.atom]
    edge = set(oa.chain_ends) | set(oa.chain_starts)
    ca_set = set(ca)

    def counted(a, b):
        return abs(oa.resi[a] - oa.resi[b]) - 2 + int(a in edge) * int(b in edge) + int(a not in ca_set) + int(b not in ca_set) >= 0

    pairs = set()
    by_res = {}
    for a in ca + cb:
        by_res.setdefault(int(oa.resi[a]), []).append(a)
    for r, atoms in by_res.items():                       # only |res1 - res2| <= 1 can fail the test
        for a in atoms:
            for b in atoms + by_res.get(r + 1, []):
                if a != b and not counted(a, b):
                    pairs.add((min(a, b), max(a, b)))
    excl.excl_pairs = sorted(pairs)
    excl.setForceGroup(forceGroup)
    return excl


def _rama_bonds(oa, i):
    return [oa.c[i - 1], oa.n[i], oa.ca[i], oa.c[i], oa.n[i + 1]]


def rama_term(oa, k_rama=8.368, num_rama_wells=3, w=[1.3149, 1.32016, 1.0264], sigma=[15.398, 49.0521, 49.0954],
              omega_phi=[0.15, 0.25, 0.65], phi_i=[-1.74, -1.265, 1.041], omega_psi=[0.65, 0.45, 0.25],
              psi_i=[2.138, -0.318, 0.78], forceGroup=21):
    """Ramachandran wells for interior non-Gly, non-Pro residues (reference basicTerms.py:266-321)."""
    _no_periodic(oa)
    k_rama *= oa.k_awsem
    n = num_rama_wells
    rama = RamaForce(k_rama, w[:n], sigma[:n], omega_phi[:n], phi_i[:n], omega_psi[:n], psi_i[:n])
    for i in range(oa.nres):
        if i not in oa.chain_starts and i not in oa.chain_ends and not oa.res_type[i] == "IGL" and not oa.res_type[i] == "IPR":
            rama.addBond(_rama_bonds(oa, i))
    rama.setForceGroup(forceGroup)
    return rama


def rama_proline_term(oa, k_rama_proline=8.368, num_rama_proline_wells=2, w=[2.17, 2.15], sigma=[105.52, 109.09],
                      omega_phi=[1.0, 1.0], phi_i=[-1.153, -0.95], omega_psi=[0.15, 0.15], psi_i=[2.4, -0.218],
                      forceGroup=21):
    """Ramachandran wells for interior prolines (reference basicTerms.py:324-379)."""
    _no_periodic(oa)
    k_rama_proline *= oa.k_awsem
    n = num_rama_proline_wells
    rama = RamaForce(k_rama_proline, w[:n], sigma[:n], omega_phi[:n], phi_i[:n], omega_psi[:n], psi_i[:n])
    for i in range(oa.nres):
        if i not in oa.chain_starts and i not in oa.chain_ends and oa.res_type[i] == "IPR":
            rama.addBond(_rama_bonds(oa, i))
    rama.setForceGroup(forceGroup)
    return rama


def rama_ssweight_term(oa, k_rama_ssweight=8.368, num_rama_wells=2, w=[2.0, 2.0], sigma=[419.0, 15.398],
                       omega_phi=[1.0, 1.0], phi_i=[-0.995, -2.25], omega_psi=[1.0, 1.0], psi_i=[-0.82, 2.16],
                       ssweight_file="ssweight", forceGroup=21):
    """Secondary-structure weighted wells (reference basicTerms.py:382-440).

    Two quirks of the reference are kept: prolines are included (the test at :435 compares the whole
    ``oa.res_type`` list with "IPR", which is never equal), and the table is
    ``Discrete2DFunction(2, nres, ssweight.flatten())`` looked up as ssweight(well, resId), i.e. the value at flat
    index ``well + 2*resId`` = ssweight[resId][well] for the 2-column file."""
    _no_periodic(oa)
    k_rama_ssweight *= oa.k_awsem
    n = num_rama_wells
    ssweight = np.loadtxt(ssweight_file).flatten()
    rama = RamaForce(k_rama_ssweight, w[:n], sigma[:n], omega_phi[:n], phi_i[:n], omega_psi[:n], psi_i[:n])
    for i in range(oa.nres):
        if i not in oa.chain_starts and i not in oa.chain_ends and not oa.res_type[i] == "IGL":
            rama.addBond(_rama_bonds(oa, i), [ssweight[well + 2 * i] for well in range(n)])
    rama.setForceGroup(forceGroup)
    return rama
