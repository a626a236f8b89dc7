"""Order parameters and the biases built on them: Q (fraction of native contacts), Qc, partial Q, the umbrella bias on Q,
the radius of gyration and its bias.

Mirrors openawsem/functionTerms/biasTerms.py: q_value 146-188 (reader 13-83), qc_value 191-223 and partial_q_value 226-267
(reader 86-143), qbias_term 270-302, rg_term 461-486, rg_bias_term 489-528.  In the reference these are OpenMM
CustomBondForce sums, alone or wrapped in a CustomCVForce; here each is one `BondSumForce` descriptor evaluated on the
device (awsem_add_bond_sum), so the order parameters printed by awsem_analyze are force-group energies exactly as in
the reference (groups 1 and 2, scripts/mm_analyze.py:139-182) and the biases act in dynamics.
"""
from __future__ import annotations

import numpy as np

from .. import unit as _u
from ..forces import BondSumForce


def _reference_ca_by_chain(pdb_file):
    chains, seen = [], set()
    with open(pdb_file) as fh:
        for line in fh:
            if line.startswith("ENDMDL"):
                break
            if line.startswith("ATOM") and line[12:16].strip() == "CA":
                key = (line[21], line[22:27])
                if key in seen:
                    continue
                seen.add(key)
                if not chains or chains[-1][0] != line[21]:
                    chains.append((line[21], []))
                chains[-1][1].append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
    return [(cid, np.array(xyz)) for cid, xyz in chains]


def _kj(k):
    return k.value_in_unit(_u.kilojoule_per_mole) if hasattr(k, "value_in_unit") else float(k)


def read_reference_structure_for_q_calculation_3(oa, pdb_file, reference_chain_name="ALL", min_seq_sep=3, max_seq_sep=np.inf,
                                                 contact_threshold=0.95 * _u.nanometers, Qflag=0, a=0.1, removeDNAchains=True):
    """[atom_i, atom_j, [gamma_ij, r_ijN (nm), sigma_ij]] for the intra-chain CA pairs with min_seq_sep <= |i-j| <= max_seq_sep,
    in both orders as the reference's double loop produces them; Qflag 1 (Q_o) keeps native contacts below the threshold only.
    Chains without CA atoms (DNA, ligands) never enter, which is what removeDNAchains does in the reference."""
    thr = contact_threshold.value_in_unit(_u.nanometer) if hasattr(contact_threshold, "value_in_unit") else float(contact_threshold)
    out = []
    chain_start = 0
    for cid, xyz in _reference_ca_by_chain(pdb_file):
        n = len(xyz)
        for i in range(n):
            for j in range(n):
                sep = abs(i - j)
                if sep >= min_seq_sep and sep <= max_seq_sep:
                    r = float(np.linalg.norm(xyz[i] - xyz[j]) / 10.0)
                    if Qflag == 1 and r >= thr:
                        continue
                    if reference_chain_name != "ALL" and cid not in reference_chain_name:
                        continue
                    out.append([oa.ca[i + chain_start], oa.ca[j + chain_start], [1.0, r, a * sep ** 0.15]])
        chain_start += n
    return out


def read_reference_structure_for_qc_calculation(oa, pdb_file, min_seq_sep=3, a=0.1, startResidueIndex=0, endResidueIndex=-1,
                                                residueIndexGroup=None):
    """Pairs i < j of the same chain with j - i >= min_seq_sep over the residues of the whole structure (global indices),
    restricted to [startResidueIndex, endResidueIndex] and to residueIndexGroup when given."""
    chains = _reference_ca_by_chain(pdb_file)
    xyz = np.concatenate([c for _, c in chains]) if chains else np.zeros((0, 3))
    chain_of = np.concatenate([[k] * len(c) for k, (_, c) in enumerate(chains)]) if chains else np.zeros(0, int)
    n = len(xyz)
    if endResidueIndex == -1:
        endResidueIndex = n
    group = set(range(n)) if residueIndexGroup is None else set(residueIndexGroup)
    out = []
    for i in range(n):
        if i < startResidueIndex or i not in group:
            continue
        for j in range(n):
            if j > endResidueIndex or j not in group:
                continue
            if j - i >= min_seq_sep and chain_of[i] == chain_of[j]:
                out.append([oa.ca[i], oa.ca[j], [1.0, float(np.linalg.norm(xyz[i] - xyz[j]) / 10.0), a * abs(i - j) ** 0.15]])
    return out


def _q_force(interactions, forceGroup, cv="sum", k=0.0, x0=0.0):
    f = BondSumForce("gauss", cv, k, x0, group=forceGroup)
    norm = len(interactions)
    for i, j, (gamma, r, sigma) in interactions:
        f.addBond(i, j, [gamma / norm, r, sigma])      # "(1/normalization)*gamma_ij*exp(-(r-r_ijN)^2/(2*sigma_ij^2))"
    return f


def q_value(oa, reference_pdb_file, reference_chain_name="ALL", min_seq_sep=3, max_seq_sep=np.inf,
            contact_threshold=0.95 * _u.nanometers, forceGroup=1):
    """Canonical Q_w: mean over the listed CA pairs of exp(-(r - r_native)^2 / (2 sigma_ij^2)), sigma_ij = 0.1 |i-j|^0.15 nm."""
    inter = read_reference_structure_for_q_calculation_3(oa, reference_pdb_file, reference_chain_name=reference_chain_name,
                                                         min_seq_sep=min_seq_sep, max_seq_sep=max_seq_sep,
                                                         contact_threshold=contact_threshold, Qflag=0)
    return _q_force(inter, forceGroup)


def qc_value(oa, reference_pdb_file, min_seq_sep=10, a=0.2):
    return _q_force(read_reference_structure_for_qc_calculation(oa, reference_pdb_file, min_seq_sep=min_seq_sep, a=a), 3)


def partial_q_value(oa, reference_pdb_file, min_seq_sep=3, a=0.1, startResidueIndex=0, endResidueIndex=-1,
                    residueIndexGroup=None, forceGroup=4):
    inter = read_reference_structure_for_qc_calculation(oa, reference_pdb_file, min_seq_sep=min_seq_sep, a=a,
                                                        startResidueIndex=startResidueIndex, endResidueIndex=endResidueIndex,
                                                        residueIndexGroup=residueIndexGroup)
    if len(inter) == 0:
        raise ValueError("No atom found, Please check your startResidueIndex and endResidueIndex.")
    return _q_force(inter, forceGroup)


def qbias_term(oa, q0, reference_pdb_file, reference_chain_name, k_qbias=100 * _u.kilocalorie_per_mole, qbias_min_seq_sep=3,
               qbias_max_seq_sep=np.inf, qbias_contact_threshold=0.8 * _u.nanometers, forceGroup=4):
    """Umbrella bias 0.5 k_qbias (Q - q0)^2 (CustomCVForce over q_value); k_qbias is not scaled by k_awsem, as in the reference."""
    inter = read_reference_structure_for_q_calculation_3(oa, reference_pdb_file, reference_chain_name=reference_chain_name,
                                                         min_seq_sep=qbias_min_seq_sep, max_seq_sep=qbias_max_seq_sep,
                                                         contact_threshold=qbias_contact_threshold, Qflag=0)
    return _q_force(inter, forceGroup, cv="harmonic", k=_kj(k_qbias), x0=float(q0))


def _rg_square(ca_atoms, cv, k, x0, group, bond="r2"):
    f = BondSumForce(bond, cv, k, x0, group=group)
    n = len(ca_atoms)
    for a in range(n):
        for b in range(a + 1, n):
            f.addBond(ca_atoms[a], ca_atoms[b], [1.0 / (n * n)])
    return f


def rg_term(oa, convertToAngstrom=True):
    """Radius of gyration of the CA trace, unit * sqrt(sum_{i<j} r_ij^2 / nres^2), in Angstrom by default (group 2)."""
    return _rg_square([oa.ca[i] for i in range(oa.nres)], "sqrt", 10.0 if convertToAngstrom else 1.0, 0.0, 2)


def rg_bias_term(oa, k=1 * _u.kilocalorie_per_mole, rg0=0.0, atomGroup=-1, forceGroup=27):
    """k_rg (Rg - rg0)^2 over the CA atoms of the residues in atomGroup (all by default); Rg in nm."""
    group = list(range(oa.nres)) if (isinstance(atomGroup, int) and atomGroup == -1) else list(atomGroup)
    return _rg_square([oa.ca[i] for i in group], "sqrt_square", oa.k_awsem * _kj(k), float(rg0), forceGroup)


def cylindrical_rg_bias_term(oa, k=1 * _u.kilocalorie_per_mole, rg0=0.0, atomGroup=-1, forceGroup=27):
    """rg_bias_term on the xy projection: k_rg (sqrt(sum_{i<j} ((x_i-x_j)^2 + (y_i-y_j)^2) / n^2) - rg0)^2 (reference
    biasTerms.py:531-571)."""
    group = list(range(oa.nres)) if (isinstance(atomGroup, int) and atomGroup == -1) else list(atomGroup)
    return _rg_square([oa.ca[i] for i in group], "sqrt_square", oa.k_awsem * _kj(k), float(rg0), forceGroup, bond="r2_xy")


def pulling_term(oa, k_pulling=4.184, forceDirect="x", appliedToResidue=1, forceGroup=19):
    """Constant force k_pulling (kJ/mol/nm) along +x, +y or +z on every atom of one residue: E = -k x (reference
    biasTerms.py:574-605).  `appliedToResidue` is 1-based, or "FIRST" / "LAST"."""
    from ..forces import EXT_LINEAR_X, ExternalForce
    axis = {"x": 0, "y": 1, "z": 2}.get(str(forceDirect).strip().lower())
    if axis is None:
        raise ValueError(f"forceDirect must be x, y or z, not {forceDirect!r}")
    k_pulling *= oa.k_awsem
    if appliedToResidue == "LAST":
        appliedToResidue = oa.nres
    if appliedToResidue == "FIRST":
        appliedToResidue = 1
    pulling = ExternalForce(EXT_LINEAR_X + axis, k_pulling)
    for i in range(oa.natoms):
        if oa.resi[i] == (appliedToResidue - 1):
            pulling.addParticle(i)
    pulling.setForceGroup(forceGroup)
    return pulling
