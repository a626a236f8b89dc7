"""Fragment (associative) memory term.

Mirrors fragment_memory_term of openawsem/functionTerms/templateTerms.py:122-294: per target atom pair a 500-point
table of weighted Gaussians summed over the memories, evaluated with linear interpolation.  The product is a
FragmentMemoryForce descriptor; the HIP engine gathers the two neighbouring table entries per bond.
"""
from __future__ import annotations

import os

import numpy as np

from ..forces import FragmentMemoryForce, GaussianBondForce


def read_memory_list(frag_file_list_file):
    """Rows of a .mem file after its 4 header lines: (gro path, target_start, fragment_start, frag_len, weight)
    (reference templateTerms.py:183)."""
    rows = []
    with open(frag_file_list_file) as fh:
        lines = fh.readlines()[4:]
    for line in lines:
        t = line.split()
        if len(t) < 5:
            continue
        rows.append((t[0], int(t[1]), int(t[2]), int(t[3]), float(t[4])))
    return rows


def read_gro_ca_cb(frag_name, fragment_start, frag_len):
    """CA/CB atoms of residues [fragment_start, fragment_start+frag_len) from a .gro file: skip 2 lines, whitespace
    columns (Res_id, Res, Type, i, x, y, z) in nm (reference templateTerms.py:192-193)."""
    out = []
    with open(frag_name) as fh:
        lines = fh.readlines()[2:]
    for line in lines:
        t = line.split()
        if len(t) < 7 or t[2] not in ("CA", "CB"):
            continue
        try:
            res_id = int(t[0])
        except ValueError:
            continue
        if fragment_start <= res_id < fragment_start + frag_len:
            out.append((res_id, t[2], float(t[4]), float(t[5]), float(t[6])))
    return out


def build_fragment_table(oa, frag_file_list_file, min_seq_sep=3, max_seq_sep=9, fm_well_width=0.1,
                         frag_table_rmin=0, frag_table_rmax=5, frag_table_dr=0.01):
    """(pairs[nbonds,2], table[nbonds,500]) restating reference templateTerms.py:154-244."""
    r_array = np.arange(frag_table_rmin, frag_table_rmax, frag_table_dr)
    data_dic = {}
    ca_set, cb_set = set(oa.ca), set(oa.cb)
    for i in range(oa.natoms):
        if i in ca_set:
            data_dic[("CA", 1 + int(oa.resi[i]))] = i      # .mem target residue ids are 1-based and global
        if i in cb_set:
            data_dic[("CB", 1 + int(oa.resi[i]))] = i
    frag_location_pre = os.path.dirname(str(frag_file_list_file))
    tables = {}
    for location, target_start, fragment_start, frag_len, weight in read_memory_list(frag_file_list_file):
        frag = read_gro_ca_cb(os.path.join(frag_location_pre, location), fragment_start, frag_len)
        for i in range(len(frag)):
            res_i, type_i, xi, yi, zi = frag[i]
            for j in range(i, len(frag)):
                res_j, type_j, xj, yj, zj = frag[j]
                seq_sep = res_j - res_i
                if seq_sep > max_seq_sep or seq_sep < min_seq_sep:
                    continue
                ti = data_dic.get((type_i, res_i - fragment_start + target_start))
                tj = data_dic.get((type_j, res_j - fragment_start + target_start))
                if ti is None or tj is None:     # e.g. a CB of the memory facing a glycine of the target
                    continue
                sigma_ij = fm_well_width * seq_sep ** 0.15
                rm = ((xi - xj) ** 2 + (yi - yj) ** 2 + (zi - zj) ** 2) ** 0.5
                if tj - ti <= 0:
                    raise AssertionError("fragment memory pair is not in increasing atom order")
                g = weight * 1 * np.exp((r_array - rm) ** 2 / (-2.0 * sigma_ij ** 2))
                key = (ti, tj)
                if key in tables:
                    tables[key] += g
                else:
                    tables[key] = g
    pairs = sorted(tables)
    table = np.array([tables[p] for p in pairs], dtype=np.float64).reshape(len(pairs), len(r_array))
    return np.array(pairs, dtype=np.int32).reshape(-1, 2), table


def fragment_memory_term(oa, k_fm=0.04184, frag_file_list_file="./frag.mem", npy_frag_table="./frag_table.npy",
                         min_seq_sep=3, max_seq_sep=9, fm_well_width=0.1, UseSavedFragTable=True, caOnly=False,
                         forceGroup=23):
    """E = -k_fm * lerp(table[bond], r / 0.01) with the index clamped to 498 (reference templateTerms.py:122-294).

    The reference caches (table, pair set, dict) as a *pickle* under `npy_frag_table`; this engine never unpickles:
    its cache is a plain ``numpy.savez`` archive under the same name, and a file it cannot read that way is
    ignored and rebuilt from the .mem/.gro files."""
    k_fm *= oa.k_awsem
    frag_table_dr = 0.01
    frag_table_file = str(npy_frag_table)
    pairs = table = None
    if os.path.isfile(frag_table_file) and UseSavedFragTable:
        try:
            with np.load(frag_table_file, allow_pickle=False) as z:
                pairs, table = z["pairs"], z["table"]
            print(f"Reading Fragment table from {frag_table_file}.")
            print(f"Fragment table loaded, number of bonds: {len(pairs)}")
        except Exception:
            print(f"{frag_table_file} is not a numpy archive written by this engine (the reference writes a pickle, "
                  "which is never loaded here); rebuilding the table from the memory files.")
    if pairs is None:
        print("Loading Fragment files(Gro files)")
        pairs, table = build_fragment_table(oa, frag_file_list_file, min_seq_sep, max_seq_sep, fm_well_width,
                                            frag_table_dr=frag_table_dr)
        try:
            with open(frag_table_file, "wb") as fh:
                np.savez(fh, pairs=pairs, table=table)
        except OSError:
            pass
    if caOnly:
        ca = set(oa.ca)
        keep = [b for b, (i, j) in enumerate(pairs) if i in ca and j in ca]
        pairs, table = pairs[keep], table[keep]
    fm = FragmentMemoryForce(k_fm, pairs, table, frag_table_dr)
    fm.setForceGroup(forceGroup)
    return fm


# --------------------------------------------------------------------------- Gaussian-well restraint terms
def _nm(x):
    """float nm from a plain number (already nm) or a unit Quantity."""
    from .. import unit as u
    return float(x.value_in_unit(u.nanometer)) if hasattr(x, "value_in_unit") else float(x)


def _pdb_ca_cb(pdb_file, chain_name, with_resseq=False):
    """Per residue of `chain_name` (first model): (residue name, CA xyz, CB xyz or None), coordinates in Angstrom; with
    `with_resseq` a fourth entry, the residue number of the PDB file."""
    residues, index = [], {}
    with open(pdb_file) as fh:
        for line in fh:
            if line.startswith("ENDMDL"):
                break
            if not line.startswith(("ATOM", "HETATM")) or line[21] != chain_name:
                continue
            name = line[12:16].strip()
            if name not in ("CA", "CB"):
                continue
            key = line[22:27]
            if key not in index:
                index[key] = len(residues)
                residues.append([line[17:20].strip(), None, None] + ([int(line[22:26])] if with_resseq else []))
            xyz = np.array([float(line[30:38]), float(line[38:46]), float(line[46:54])])
            residues[index[key]][1 if name == "CA" else 2] = xyz
    return residues


def read_amhgo_structure(oa, pdb_file, chain_name, amhgo_min_seq_sep=4, amhgo_contact_threshold=0.8, amhgo_well_width=0.1):
    """Native contacts of the AMH-Go term (reference templateTerms.py:535-597): for residues i - j >= min_seq_sep every
    CA/CB x CA/CB pair (no CB for GLY/IGL) closer than the threshold in `pdb_file` gives a bond
    [atom_i, atom_j, [gamma = 1, r_native (nm), sigma = well_width |i-j|^0.15]]."""
    thr = _nm(amhgo_contact_threshold)
    residues = _pdb_ca_cb(pdb_file, chain_name)
    out = []
    for i, (name_i, ca_i, cb_i) in enumerate(residues):
        for j, (name_j, ca_j, cb_j) in enumerate(residues):
            if i - j < amhgo_min_seq_sep:
                continue
            atoms_i = [(ca_i, oa.ca[i])] + ([(cb_i, oa.cb[i])] if name_i not in ("GLY", "IGL") else [])
            atoms_j = [(ca_j, oa.ca[j])] + ([(cb_j, oa.cb[j])] if name_j not in ("GLY", "IGL") else [])
            for xi, ai in atoms_i:
                for xj, aj in atoms_j:
                    r = float(np.linalg.norm(xi - xj)) / 10.0
                    if r <= thr:
                        out.append([ai, aj, [1.0, r, amhgo_well_width * abs(i - j) ** 0.15]])
    return out


def additive_amhgo_term(oa, pdb_file, chain_name, k_amhgo=4.184, amhgo_min_seq_sep=3, amhgo_contact_threshold=0.8,
                        amhgo_well_width=0.1, forceGroup=22):
    """AMH-Go structure-based term (reference templateTerms.py:600-640): Gaussian wells at the native distances."""
    force = GaussianBondForce(k_amhgo * oa.k_awsem)
    for a, b, params in read_amhgo_structure(oa, pdb_file, chain_name, amhgo_min_seq_sep, amhgo_contact_threshold, amhgo_well_width):
        force.addBond(a, b, params)
    force.setForceGroup(forceGroup)
    return force


def read_memory(oa, pdb_file, chain_name, target_start, fragment_start, length, weight, min_seq_sep, max_seq_sep, am_well_width=0.1):
    """Bonds of one associative memory (reference templateTerms.py:297-370): the residues of `chain_name` numbered
    fragment_start .. fragment_start + length - 2 (the reference's `range(fragment_start, fragment_start+length-1)`) are
    laid on the target residues from `target_start` (1-based); every CA/CB x CA/CB pair of residues with
    min_seq_sep <= i - j <= max_seq_sep gives [atom_i, atom_j, [weight, gamma = 1, r_memory (nm), well_width |i-j|^0.15]]."""
    residues = [r for r in _pdb_ca_cb(pdb_file, chain_name, with_resseq=True) if fragment_start <= r[3] < fragment_start + length - 1]
    out = []
    for i, (name_i, ca_i, cb_i, _) in enumerate(residues):
        for j, (name_j, ca_j, cb_j, _) in enumerate(residues):
            if abs(i - j) > max_seq_sep or i - j < min_seq_sep:
                continue
            ti, tj = target_start + i - 1, target_start + j - 1
            atoms_i, atoms_j = [(ca_i, oa.ca[ti])], [(ca_j, oa.ca[tj])]
            if name_i != "GLY" and oa.cb[ti] >= 0:
                atoms_i.append((cb_i, oa.cb[ti]))
            if name_j != "GLY" and oa.cb[tj] >= 0:
                atoms_j.append((cb_j, oa.cb[tj]))
            for xi, ai in atoms_i:
                for xj, aj in atoms_j:
                    if xi is None or xj is None:
                        raise ValueError(f"{pdb_file}: a residue of chain {chain_name} lacks the CA or CB atom the memory needs")
                    out.append([ai, aj, [weight, 1.0, float(np.linalg.norm(xi - xj)) / 10.0, am_well_width * abs(i - j) ** 0.15]])
    return out


def associative_memory_term(oa, memories, k_am=0.8368, min_seq_sep=3, max_seq_sep=9, am_well_width=0.1):
    """Associative-memory term over whole structures (reference templateTerms.py:373-411):
    E = -k_am sum w_m gamma_ij exp(-(r - r_ijm)^2 / (2 sigma_ij^2)); `memories` holds tuples
    (pdb_file, chain, target_start, fragment_start, length, weight).  Force group 0, as the reference leaves it."""
    force = GaussianBondForce(k_am * oa.k_awsem)
    for memory in memories:
        for a, b, (w_m, gamma_ij, r_ijm, sigma_ij) in read_memory(oa, *memory, min_seq_sep, max_seq_sep, am_well_width=am_well_width):
            force.addBond(a, b, [w_m * gamma_ij, r_ijm, sigma_ij])
    return force


def er_term(oa, k_er=4.184, er_min_seq_sep=2, er_cutoff=99.0, er_well_width=0.1, forceGroup=25):
    """Evolutionary-restraint term (reference templateTerms.py:643-693): Gaussian wells at the predicted CA-CA, CA-CB and
    CB-CB distances read from go_rnativeCACA.dat / go_rnativeCACB.dat / go_rnativeCBCB.dat (Angstrom, nres x nres) in the
    current directory, for |i-j| >= er_min_seq_sep and predicted distance < er_cutoff."""
    force = GaussianBondForce(k_er * oa.k_awsem)
    caca, cacb, cbcb = (np.loadtxt(f) for f in ("go_rnativeCACA.dat", "go_rnativeCACB.dat", "go_rnativeCBCB.dat"))
    for i in range(oa.nres):
        for j in range(oa.nres):
            if abs(i - j) < er_min_seq_sep:
                continue
            sigma = er_well_width * abs(i - j) ** 0.15
            if caca[i][j] < er_cutoff:
                force.addBond(oa.ca[i], oa.ca[j], [1.0, caca[i][j] / 10.0, sigma])
            if cacb[i][j] < er_cutoff and oa.cb[j] != -1:
                force.addBond(oa.ca[i], oa.cb[j], [1.0, cacb[i][j] / 10.0, sigma])
            if cbcb[i][j] < er_cutoff and oa.cb[j] != -1 and oa.cb[i] != -1:
                force.addBond(oa.cb[i], oa.cb[j], [1.0, cbcb[i][j] / 10.0, sigma])
    force.setForceGroup(forceGroup)
    return force


def read_reference_structure_for_q_calculation_4(oa, contact_threshold, rnative_dat, min_seq_sep=3, max_seq_sep=np.inf):
    """[ca_i, ca_j, [1.0, r_ijN (nm), sigma_ij]] for j - i in [min_seq_sep, max_seq_sep] from an nres x nres table of native
    CA distances in Angstrom; pairs whose native distance is BELOW the threshold are skipped, as in the reference
    (templateTerms.py:17-58)."""
    thr = _nm(contact_threshold)
    table = np.loadtxt(rnative_dat)
    out = []
    for i in range(oa.nres):
        for j in range(oa.nres):
            if j - i >= min_seq_sep and j - i <= max_seq_sep:
                r = table[i][j] / 10.0
                if r < thr:
                    continue
                out.append([oa.ca[i], oa.ca[j], [1.0, float(r), 0.1 * abs(i - j) ** 0.15]])
    return out


def q_value_dat(oa, contact_threshold, rnative_dat="rnative.dat", min_seq_sep=3, max_seq_sep=np.inf, forceGroup=0):
    """Q against the distances of rnative.dat (templateTerms.py:60-91): a plain bond sum."""
    from ..forces import BondSumForce
    inter = read_reference_structure_for_q_calculation_4(oa, contact_threshold, rnative_dat, min_seq_sep, max_seq_sep)
    f = BondSumForce("gauss", "sum", group=forceGroup)
    for i, j, (gamma, r, sigma) in inter:
        f.addBond(i, j, [gamma / len(inter), r, sigma])
    return f


def tbm_q_term(oa, k_tbm_q, rnative_dat="rnative.dat", tbm_q_min_seq_sep=3, tbm_q_cutoff=0.2, tbm_q_well_width=0.1,
               target_q=1.0, forceGroup=26):
    """Template-based-modelling bias k_tbm_q (q - target_q)^2 (templateTerms.py:94-121, CustomCVForce over q_value_dat);
    tbm_q_cutoff in nm or a Quantity (the reference's default is 0.2 nm)."""
    print("TBM_Q term ON")
    f = q_value_dat(oa, tbm_q_cutoff, rnative_dat, tbm_q_min_seq_sep, np.inf, forceGroup)
    f.cv, f.k, f.x0 = "harmonic", 2.0 * float(k_tbm_q), float(target_q)      # k/2 (S - x0)^2 with k = 2 k_tbm_q
    return f
