"""Beta-sheet hydrogen bond (beta_1/2/3), P-AP (pap_1/2) and helical terms.

Mirrors openawsem/functionTerms/hydrogenBondTerms.py (beta 418-611, pap 614-761, helical 790-832 and the
lambda/gamma table builders 118-359).  The products are HbondBetaForce / HbondPapForce / HelicalForce
descriptors carrying the same dense nres x nres tables the reference feeds to OpenMM's Discrete2DFunction
(row = donor residue, column = acceptor residue).
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np

from .. import unit as _u
from ..forces import HbTableFn, HbondBetaForce, HbondPapForce, HelicalForce
from .contactTerms import _chain_index, inWhichChain  # noqa: F401  (inWhichChain re-exported like the reference)

se_map_1_letter = {'A': 0, 'P': 1, 'K': 2, 'N': 3, 'R': 4,
                   'F': 5, 'D': 6, 'Q': 7, 'E': 8, 'G': 9,
                   'I': 10, 'H': 11, 'L': 12, 'C': 13, 'M': 14,
                   'S': 15, 'T': 16, 'Y': 17, 'V': 18, 'W': 19}


def isChainStart(residueId, chain_starts, n=2):
    """True if residueId is within n residues of a chain start (reference hydrogenBondTerms.py:21-42)."""
    return any((residueId - i) in chain_starts for i in range(n))


def isChainEnd(residueId, chain_ends, n=2):
    """True if residueId is within n residues of a chain end (reference hydrogenBondTerms.py:45-66)."""
    return any((residueId + i) in chain_ends for i in range(n))


def isChainEdge(residueId, chain_starts, chain_ends, n=2):
    return isChainStart(residueId, chain_starts, n) or isChainEnd(residueId, chain_ends, n)


def read_beta_parameters(parametersLocation=None):
    """anti_HB / anti_NHB / para_HB (20 rows, blank, 20 rows) and para_one / anti_one (reference :118-157)."""
    if parametersLocation is None:
        from .. import data_path
        parametersLocation = data_path.parameters
    loc = Path(parametersLocation)
    assert loc.is_dir(), 'Directory does not exist'

    def rows(name):
        with open(loc / name) as fh:
            return fh.readlines()
    anti_HB, anti_NHB, para_HB = rows("anti_HB"), rows("anti_NHB"), rows("para_HB")
    para_one, anti_one = rows("para_one"), rows("anti_one")
    p_par = np.array([float(para_one[i].strip()) for i in range(20)])
    p_anti = np.array([float(anti_one[i].strip()) for i in range(20)])
    p_antihb, p_antinhb, p_parhb = np.zeros((20, 20, 2)), np.zeros((20, 20, 2)), np.zeros((20, 20, 2))
    for block, offset in ((0, 0), (1, 21)):
        for i in range(20):
            p_antihb[i, :, block] = [float(x) for x in anti_HB[i + offset].split()[:20]]
            p_antinhb[i, :, block] = [float(x) for x in anti_NHB[i + offset].split()[:20]]
            p_parhb[i, :, block] = [float(x) for x in para_HB[i + offset].split()[:20]]
    return p_par, p_anti, p_antihb, p_antinhb, p_parhb


_LAMBDA_TABLE = [[1.37, 1.36, 1.17],
                 [3.89, 3.50, 3.52],
                 [0.00, 3.47, 3.62]]
_ALPHA_TABLE = [[1.30, 1.30, 1.30],
                [1.32, 1.32, 1.32],
                [1.22, 1.22, 1.22],
                [0.00, 0.33, 0.33],
                [0.00, 1.01, 1.01]]


def _sep_class(i, j):
    d = abs(j - i)
    if 4 <= d < 18:
        return 0
    if 18 <= d < 45:
        return 1
    if d >= 45:
        return 2
    return -1


def get_lambda_by_index(i, j, lambda_i):
    """Sequence-separation class lookup (reference :160-184)."""
    c = _sep_class(i, j)
    return _LAMBDA_TABLE[lambda_i][c] if c >= 0 else 0.0


def get_alpha_by_index(i, j, alpha_i):
    """Reference :187-213."""
    c = _sep_class(i, j)
    return _ALPHA_TABLE[alpha_i][c] if c >= 0 else 0.0


def get_pap_gamma_APH(donor_idx, acceptor_idx, chain_i, chain_j, gamma_APH):
    """Reference :216-237."""
    if 13 <= donor_idx - acceptor_idx <= 16 and chain_i == chain_j:
        return gamma_APH
    return 0.0


def _pap_scale(donor_idx, acceptor_idx, ssweight):
    return 1.5 if ssweight[donor_idx][1] == 1 and ssweight[acceptor_idx][1] == 1 else 1.0


def get_pap_gamma_AP(donor_idx, acceptor_idx, chain_i, chain_j, gamma_AP, ssweight):
    """Reference :240-267."""
    if donor_idx - acceptor_idx >= 17 or chain_i != chain_j:
        return _pap_scale(donor_idx, acceptor_idx, ssweight) * gamma_AP
    return 0.0


def get_pap_gamma_P(donor_idx, acceptor_idx, chain_i, chain_j, gamma_P, ssweight):
    """Reference :270-297."""
    if donor_idx - acceptor_idx >= 9 or chain_i != chain_j:
        return _pap_scale(donor_idx, acceptor_idx, ssweight) * gamma_P
    return 0.0


def get_Lambda_2(i, j, p_par, p_anti, p_antihb, p_antinhb, p_parhb, a):
    """Reference :300-328."""
    Lambda = get_lambda_by_index(i, j, 1)
    Lambda += -0.5 * get_alpha_by_index(i, j, 0) * p_antihb[a[i], a[j]][0]
    Lambda += -0.25 * get_alpha_by_index(i, j, 1) * (p_antinhb[a[i + 1], a[j - 1]][0] + p_antinhb[a[i - 1], a[j + 1]][0])
    Lambda += -get_alpha_by_index(i, j, 2) * (p_anti[a[i]] + p_anti[a[j]])
    return Lambda


def get_Lambda_3(i, j, p_par, p_anti, p_antihb, p_antinhb, p_parhb, a):
    """Reference :331-359."""
    Lambda = get_lambda_by_index(i, j, 2)
    Lambda += -get_alpha_by_index(i, j, 3) * p_parhb[a[i + 1], a[j]][0]
    Lambda += -get_alpha_by_index(i, j, 4) * p_par[a[i + 1]]
    Lambda += -get_alpha_by_index(i, j, 3) * p_par[a[j]]
    return Lambda


def convert_units(k):
    """Quantity -> kJ/mol, plain numbers pass through (reference :396-415)."""
    if isinstance(k, (float, int)):
        return k
    if _u.is_quantity(k):
        return k.value_in_unit(_u.kilojoule_per_mole)
    raise TypeError(f"Unknown input, {k}, {type(k)}")


def _lambda_table_scalar(oa, fn):
    """lambda[i][j] = fn(i, j) for i, j not at a chain edge (n=1), else 0 (reference :513-519, 577-583): the reference's
    own double loop, kept as the checker of the vectorised builder below (tests/test_host_tables.py)."""
    nres = oa.nres
    params = read_beta_parameters()
    a = [se_map_1_letter[oa.seq[ii]] for ii in range(nres)]
    edge = [isChainEdge(i, oa.chain_starts, oa.chain_ends, n=1) for i in range(nres)]
    lam = np.zeros((nres, nres))
    for i in range(nres):
        if edge[i]:
            continue
        for j in range(nres):
            if edge[j] or _sep_class(i, j) < 0:
                continue   # every contribution carries a separation-class factor that is 0 for |i-j| < 4
            lam[i][j] = fn(i, j, *params, a)
    return lam


def _class_tables(rows, nres):
    """Separation class of (i, j) for i in rows, and a lookup that maps a 3-entry class table onto it (0 where |i-j| < 4)."""
    sep = np.abs(rows[:, None] - np.arange(nres)[None, :])
    cls = np.where(sep >= 45, 2, np.where(sep >= 18, 1, np.where(sep >= 4, 0, -1)))
    valid = cls >= 0
    return valid, lambda table3: np.where(valid, np.asarray(table3)[np.maximum(cls, 0)], 0.0)


def _lambda_table(oa, fn, block=512):
    """The same table as `_lambda_table_scalar`, built with numpy in row blocks (a 2 600-residue system takes 0.3 s
    instead of a minute; the operations and their order are those of get_Lambda_2 / get_Lambda_3, so the entries are
    bit-identical)."""
    nres = oa.nres
    p_par, p_anti, p_antihb, p_antinhb, p_parhb = read_beta_parameters()
    a = np.array([se_map_1_letter[oa.seq[ii]] for ii in range(nres)])
    edge = np.array([isChainEdge(i, oa.chain_starts, oa.chain_ends, n=1) for i in range(nres)])
    a_next, a_prev = np.roll(a, -1), np.roll(a, 1)     # a[i+1], a[i-1]; the wrapped ends are chain edges and zeroed below
    lam = np.zeros((nres, nres))
    for r0 in range(0, nres, block):
        rows = np.arange(r0, min(nres, r0 + block))
        valid, look = _class_tables(rows, nres)
        ai, ai_n, ai_p = a[rows][:, None], a_next[rows][:, None], a_prev[rows][:, None]
        aj, aj_n, aj_p = a[None, :], a_next[None, :], a_prev[None, :]
        if fn is get_Lambda_2:
            L = look(_LAMBDA_TABLE[1])
            L = L + -0.5 * look(_ALPHA_TABLE[0]) * p_antihb[ai, aj, 0]
            L = L + -0.25 * look(_ALPHA_TABLE[1]) * (p_antinhb[ai_n, aj_p, 0] + p_antinhb[ai_p, aj_n, 0])
            L = L + -look(_ALPHA_TABLE[2]) * (p_anti[ai] + p_anti[aj])
        elif fn is get_Lambda_3:
            L = look(_LAMBDA_TABLE[2])
            L = L + -look(_ALPHA_TABLE[3]) * p_parhb[ai_n, aj, 0]
            L = L + -look(_ALPHA_TABLE[4]) * p_par[ai_n]
            L = L + -look(_ALPHA_TABLE[3]) * p_par[aj]
        else:
            raise ValueError("unknown lambda function")
        lam[rows] = np.where(valid & ~edge[rows][:, None] & ~edge[None, :], L, 0.0)
    return lam


def _pap_gamma_table(oa, ssweight, parallel, gamma_aph=1.0, gamma_ap=0.4, gamma_p=0.4, block=512):
    """gamma[donor][acceptor] of pap_term_1 (get_pap_gamma_APH + get_pap_gamma_AP) or pap_term_2 (get_pap_gamma_P)."""
    nres = oa.nres
    chain = np.array([_chain_index(i, oa.chain_ends) for i in range(nres)])
    helix = np.asarray(ssweight)[:nres, 1] == 1
    gamma = np.zeros((nres, nres))
    for r0 in range(0, nres, block):
        rows = np.arange(r0, min(nres, r0 + block))
        d = rows[:, None] - np.arange(nres)[None, :]                     # donor - acceptor
        other = chain[rows][:, None] != chain[None, :]
        scale = np.where(helix[rows][:, None] & helix[None, :], 1.5, 1.0)
        if parallel:
            gamma[rows] = np.where((d >= 9) | other, scale * gamma_p, 0.0)
        else:
            gamma[rows] = np.where((d >= 13) & (d <= 16) & ~other, gamma_aph, 0.0) + np.where((d >= 17) | other, scale * gamma_ap, 0.0)
    return gamma


def _res_info(oa, ssweight=None):
    """[nres][4] per-residue integers of the functional table form: H-bond residue type, chain index, chain-edge flag (n=1), helix flag."""
    nres = oa.nres
    info = np.zeros((nres, 4), dtype=np.int32)
    info[:, 0] = [se_map_1_letter[oa.seq[i]] for i in range(nres)]
    info[:, 1] = [_chain_index(i, oa.chain_ends) for i in range(nres)]
    info[:, 2] = [isChainEdge(i, oa.chain_starts, oa.chain_ends, n=1) for i in range(nres)]
    if ssweight is not None:
        info[:, 3] = np.asarray(ssweight)[:nres, 1] == 1
    return info


def _lambda_1_dense(nres):
    idx = np.arange(nres)
    sep = np.abs(idx[:, None] - idx[None, :])
    return np.where((sep >= 4) & (sep < 18), 1.37, np.where((sep >= 18) & (sep < 45), 1.36, np.where(sep >= 45, 1.17, 0.0)))


def _lambda_fn(oa, which):
    """Functional form of lambda_1/2/3 (include/awsem_hip.h: AWSEM_HB_LAMBDA_*): the class tables of get_lambda_by_index /
    get_alpha_by_index with the signs and the 0.5 / 0.25 factors of get_Lambda_2 / get_Lambda_3 folded in."""
    p_par, p_anti, p_antihb, p_antinhb, p_parhb = read_beta_parameters()
    L, A = np.array(_LAMBDA_TABLE), np.array(_ALPHA_TABLE)
    info = _res_info(oa)
    if which == 1:
        return HbTableFn(1, info, class_coef=[L[0]], dense=lambda: _lambda_1_dense(oa.nres))
    if which == 2:
        return HbTableFn(2, info, class_coef=[L[1], -0.5 * A[0], -0.25 * A[1], -A[2]], t1=p_antihb[:, :, 0], t2=p_antinhb[:, :, 0],
                         one_body=p_anti, dense=lambda: _lambda_table(oa, get_Lambda_2))
    return HbTableFn(3, info, class_coef=[L[2], -A[3], -A[4]], t1=p_parhb[:, :, 0], one_body=p_par, dense=lambda: _lambda_table(oa, get_Lambda_3))


def beta_term_1(oa, k=0.5 * _u.kilocalories_per_mole, forceGroup=27):
    """-k lambda_1(don, acc) theta(O_acc; N_don, H_don) (reference :418-477)."""
    print("beta_1 term ON")
    k_beta = convert_units(k) * oa.k_awsem
    nres = oa.nres
    beta = HbondBetaForce(0, k_beta, fn=_lambda_fn(oa, 1))
    for i in range(nres):
        if oa.o[i] != -1:
            beta.addAcceptor(oa.o[i], -1, -1, [i])
        if oa.n[i] != -1 and oa.h[i] != -1:
            beta.addDonor(oa.n[i], oa.h[i], -1, [i])
    beta.setForceGroup(forceGroup)
    return beta


def beta_term_2(oa, k=0.5 * _u.kilocalories_per_mole, forceGroup=27):
    """-k lambda_2(don, acc) theta(O_acc; N_don, H_don) theta(O_don; N_acc, H_acc) (reference :480-541)."""
    print("beta_2 term ON")
    k_beta = convert_units(k) * oa.k_awsem
    beta = HbondBetaForce(1, k_beta, fn=_lambda_fn(oa, 2))
    o, n, h = oa.o, oa.n, oa.h
    for i in range(oa.nres):
        if o[i] != -1 and n[i] != -1 and h[i] != -1:
            beta.addAcceptor(o[i], n[i], h[i], [i])
            beta.addDonor(n[i], h[i], o[i], [i])
    beta.setForceGroup(forceGroup)
    return beta


def beta_term_3(oa, k=0.5 * _u.kilocalorie_per_mole, forceGroup=27):
    """-k lambda_3(don, acc) theta(O_acc; N_don, H_don) theta(O_don; N_{acc+2}, H_{acc+2}) (reference :544-611)."""
    print("beta_3 term ON")
    k_beta = convert_units(k) * oa.k_awsem
    beta = HbondBetaForce(1, k_beta, fn=_lambda_fn(oa, 3))
    o, n, h = oa.o, oa.n, oa.h
    for i in range(oa.nres):
        if isChainEdge(i, oa.chain_starts, oa.chain_ends, n=2):
            continue
        if o[i] != -1 and n[i + 2] != -1 and h[i + 2] != -1:
            beta.addAcceptor(o[i], n[i + 2], h[i + 2], [i])
        if o[i] != -1 and n[i] != -1 and h[i] != -1:
            beta.addDonor(n[i], h[i], o[i], [i])
    beta.setForceGroup(forceGroup)
    return beta


def _load_ssweight(nres, ssweightFileName):
    if not os.path.exists(ssweightFileName):
        print("No ssweight given, assume all zero")
        return np.zeros((nres, 2))
    return np.loadtxt(ssweightFileName)


def pap_term_1(oa, k=0.5 * _u.kilocalorie_per_mole, dis_i_to_i4=1.2, forceGroup=28, ssweightFileName="ssweight"):
    """Anti-parallel P-AP term: acceptors (CA_i, CA_{i+4}), donors (N_i, N_{i-4}) (reference :614-693)."""
    print("pap_1 term ON")
    k_pap = convert_units(k) * oa.k_awsem
    nres, ca = oa.nres, oa.ca
    gamma_aph, gamma_ap = 1.0, 0.4
    ssweight = _load_ssweight(nres, ssweightFileName)
    fn = HbTableFn(4, _res_info(oa, ssweight), gammas=[gamma_aph, gamma_ap, 0.0],
                   dense=lambda: _pap_gamma_table(oa, ssweight, parallel=False, gamma_aph=gamma_aph, gamma_ap=gamma_ap))
    pap = HbondPapForce(k_pap, eta=70, r0=0.8, d_i4=dis_i_to_i4, fn=fn)
    for i in range(nres):
        if not isChainEnd(i, oa.chain_ends, n=4):
            pap.addAcceptor(ca[i], ca[i + 4], -1, [i])
        if not isChainStart(i, oa.chain_starts, n=4):
            if oa.n[i] != -1 and oa.n[i - 4] != -1:
                pap.addDonor(oa.n[i], oa.n[i - 4], -1, [i])
    pap.setForceGroup(forceGroup)
    return pap


def pap_term_2(oa, k=0.5 * _u.kilocalories_per_mole, dis_i_to_i4=1.2, forceGroup=28, ssweightFileName="ssweight"):
    """Parallel P-AP term: acceptors (CA_i, CA_{i+4}), donors (N_i, N_{i+4}) (reference :696-761)."""
    print("pap_2 term ON")
    k_pap = convert_units(k) * oa.k_awsem
    nres, ca = oa.nres, oa.ca
    gamma_p = 0.4
    ssweight = _load_ssweight(nres, ssweightFileName)
    fn = HbTableFn(5, _res_info(oa, ssweight), gammas=[0.0, 0.0, gamma_p], dense=lambda: _pap_gamma_table(oa, ssweight, parallel=True, gamma_p=gamma_p))
    pap = HbondPapForce(k_pap, eta=70, r0=0.8, d_i4=dis_i_to_i4, fn=fn)
    for i in range(nres):
        if not isChainEnd(i, oa.chain_ends, n=4):
            pap.addAcceptor(ca[i], ca[i + 4], -1, [i])
            if oa.n[i] != -1 and oa.n[i + 4] != -1:
                pap.addDonor(oa.n[i], oa.n[i + 4], -1, [i])
    pap.setForceGroup(forceGroup)
    return pap


def get_helical_f(oneLetterCode, inMembrane=False):
    """Helical propensities (reference :764-787)."""
    if inMembrane:
        table = {"A": 0.79, "R": 0.62, "N": 0.49, "D": 0.44, "C": 0.76, "Q": 0.61, "E": 0.57, "G": 0.57, "H": 0.63, "I": 0.81,
                 "L": 0.81, "K": 0.56, "M": 0.80, "F": 0.76, "P": 0.44, "S": 0.6, "T": 0.67, "W": 0.74, "Y": 0.71, "V": 0.79}
    else:
        table = {"A": 0.77, "R": 0.68, "N": 0.07, "D": 0.15, "C": 0.23, "Q": 0.33, "E": 0.27, "G": 0.0, "H": 0.06, "I": 0.23,
                 "L": 0.62, "K": 0.65, "M": 0.5, "F": 0.41, "P": 0.4, "S": 0.35, "T": 0.11, "W": 0.45, "Y": 0.17, "V": 0.14}
    return table[oneLetterCode]


def helical_term(oa, k_helical=4.184, inMembrane=False, forceGroup=29):
    """-k (f_i + f_{i+4}) theta(O_i; N_{i+4}, H_{i+4}) (reference :790-832)."""
    k_helical *= oa.k_awsem
    helical = HelicalForce()
    for i in range(oa.nres):
        if not isChainEnd(i, oa.chain_ends, n=4) and oa.res_type[i + 4] != "IPR":
            fa_i = get_helical_f(oa.seq[i], inMembrane=inMembrane)
            fa_ip4 = get_helical_f(oa.seq[i + 4], inMembrane=inMembrane)
            helical.addBond([oa.o[i], oa.n[i + 4], oa.h[i + 4]], k_helical * (fa_i + fa_ip4))
    helical.setForceGroup(forceGroup)
    return helical


def z_dependent_helical_term(oa, k_helical=4.184, membrane_center=0 * _u.angstrom, z_m=1.5, forceGroup=29):
    """Helical term whose propensities blend from the water to the membrane table with the z of CA_i (reference :835-887):
    E = -k theta(O_i; N_i+4, H_i+4) ((fa_i + fa_i+4)(1 - alpha) + (fa_i^mem + fa_i+4^mem) alpha), eta_switching = 10."""
    k_helical *= oa.k_awsem
    helical = HelicalForce()
    helical.eta_switching, helical.z_m = 10.0, float(z_m)
    helical.membrane_center = membrane_center.value_in_unit(_u.nanometer)
    helical.coef_membrane = []
    for i in range(oa.nres):
        if not isChainEnd(i, oa.chain_ends, n=4) and oa.res_type[i + 4] != "IPR":
            fa = get_helical_f(oa.seq[i], inMembrane=False) + get_helical_f(oa.seq[i + 4], inMembrane=False)
            fm = get_helical_f(oa.seq[i], inMembrane=True) + get_helical_f(oa.seq[i + 4], inMembrane=True)
            helical.addBond([oa.o[i], oa.n[i + 4], oa.h[i + 4], oa.ca[i]], k_helical * fa, k_helical * fm)
    helical.setForceGroup(forceGroup)
    return helical
