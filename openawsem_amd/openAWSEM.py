"""Host-side mirror of openawsem/openAWSEM.py's OpenMMAWSEMSystem for the MI355X engine.

Same constructor, attributes and methods as the reference class (openawsem/openAWSEM.py:819-929) -- ``nres``,
``natoms``, ``resi``, ``n/h/ca/c/o/cb`` (-1 when absent), ``res_type``, ``chain_starts/ends``, ``bonds``,
``seq``, ``k_awsem``, ``periodic``, ``system``, ``pdb``, ``addForce(s)`` -- but with a pure-Python PDB reader
(no openmm.app / Biopython) and a light `forces.System` collecting term descriptors instead of an
``openmm.System``.
"""
from __future__ import annotations

import os
from pathlib import Path

import numpy as np

from .forces import System, ThreeParticleAverageSite

__location__ = Path(__file__).resolve().parent
xml = __location__ / "topology" / "awsem.xml"

three_to_one = {'ALA': 'A', 'ARG': 'R', 'ASN': 'N', 'ASP': 'D', 'CYS': 'C',
                'GLU': 'E', 'GLN': 'Q', 'GLY': 'G', 'HIS': 'H', 'ILE': 'I',
                'LEU': 'L', 'LYS': 'K', 'MET': 'M', 'PHE': 'F', 'PRO': 'P',
                'SER': 'S', 'THR': 'T', 'TRP': 'W', 'TYR': 'Y', 'VAL': 'V'}
one_to_three = {v: k for k, v in three_to_one.items()}


# --------------------------------------------------------------------------- topology / PDB
class Chain:
    def __init__(self, index, id):
        self.index, self.id, self._residues = index, id, []

    def residues(self):
        return iter(self._residues)


class Residue:
    def __init__(self, index, name, id, chain, insertionCode=""):
        self.index, self.name, self.id, self.chain, self.insertionCode = index, name, id, chain, insertionCode
        self._atoms = []

    def atoms(self):
        return iter(self._atoms)


class Atom:
    def __init__(self, index, name, element, residue, serial):
        self.index, self.name, self.element, self.residue, self.id = index, name, element, residue, serial


class Topology:
    def __init__(self):
        self._chains, self._residues, self._atoms = [], [], []

    def chains(self):
        return iter(self._chains)

    def residues(self):
        return iter(self._residues)

    def atoms(self):
        return iter(self._atoms)

    def getNumResidues(self):
        return len(self._residues)

    def getNumAtoms(self):
        return len(self._atoms)

    def getNumChains(self):
        return len(self._chains)


class PDBFile:
    """ATOM/HETATM reader with openmm.app.PDBFile's topology rules (first model only): a new chain starts when
    the chain ID changes or after TER, a new residue when (resSeq, iCode, resName) changes.  Positions are kept
    in nm (Angstrom * 0.1 in double, as openmm.app.PDBFile does)."""

    def __init__(self, filename):
        self.topology = Topology()
        pos = []
        chain = residue = None
        new_chain_pending = True
        last_res_key = None
        with open(filename) as fh:
            for line in fh:
                rec = line[:6]
                if rec.startswith("ENDMDL"):
                    break
                if rec.startswith("TER"):
                    new_chain_pending = True
                    continue
                if not (rec.startswith("ATOM") or rec.startswith("HETATM")):
                    continue
                name = line[12:16].strip()
                resname = line[17:20].strip()
                chain_id = line[21]
                resseq = line[22:26].strip()
                icode = line[26].strip() if len(line) > 26 else ""
                x, y, z = float(line[30:38]), float(line[38:46]), float(line[46:54])
                element = line[76:78].strip() if len(line) >= 78 else ""
                if chain is None or new_chain_pending or chain.id != chain_id:
                    chain = Chain(len(self.topology._chains), chain_id)
                    self.topology._chains.append(chain)
                    new_chain_pending = False
                    last_res_key = None
                key = (resseq, icode, resname)
                if key != last_res_key:
                    residue = Residue(len(self.topology._residues), resname, resseq, chain, icode)
                    self.topology._residues.append(residue)
                    chain._residues.append(residue)
                    last_res_key = key
                atom = Atom(len(self.topology._atoms), name, element, residue, line[6:11].strip())
                self.topology._atoms.append(atom)
                residue._atoms.append(atom)
                pos.append((x * 0.1, y * 0.1, z * 0.1))
        self.positions = np.array(pos, dtype=np.float64).reshape(-1, 3)

    def getTopology(self):
        return self.topology

    def getPositions(self, asNumpy=False):
        return self.positions


# masses of openawsem/topology/awsem.xml:4-9 (N, H, C are massless virtual sites)
_DEFAULT_MASSES = {"N": 0.0, "H": 0.0, "CA": 12.0, "C": 0.0, "O": 16.0, "CB": 12.0}


def read_forcefield_masses(xml_filename):
    """Atom-type masses from an awsem.xml-style force field file (<Type name=.. mass=..>)."""
    import xml.etree.ElementTree as ET
    masses = dict(_DEFAULT_MASSES)
    try:
        root = ET.parse(str(xml_filename)).getroot()
        for t in root.iter("Type"):
            masses[t.attrib["name"]] = float(t.attrib["mass"])
    except (OSError, ET.ParseError, KeyError):
        pass
    return masses


# --------------------------------------------------------------------------- helpers mirrored from the reference
def build_lists_of_atoms_2(nres, residues, atoms):
    """openAWSEM.py:592-605: per-residue atom indices by name, -1 where the PDB has no such atom."""
    table = {'N': 'n', 'H': 'h', 'CA': 'ca', 'C': 'c', 'O': 'o', 'CB': 'cb'}
    res_types = [r.name for r in residues]
    atom_lists = {k: [-1] * nres for k in ('n', 'h', 'ca', 'c', 'o', 'cb')}
    for atom in atoms:
        atom_lists[table[atom.name]][atom.residue.index] = atom.index
    return atom_lists, res_types


def get_chain_starts_and_ends(all_res):
    """openAWSEM.py:653-664."""
    chain_starts, chain_ends, pos = [], [], -1
    for i, res in enumerate(all_res):
        if res.chain.index != pos:
            chain_starts.append(i)
            pos = res.chain.index
            if i > 0:
                chain_ends.append(i - 1)
    chain_ends.append(len(all_res) - 1)
    return chain_starts, chain_ends


def setup_virtual_sites(nres, system, n, h, ca, c, o, cb, res_type, chain_starts, chain_ends, use_cis_proline=False):
    """openAWSEM.py:666-696 (weights 685-696; cis-proline variant 671-682)."""
    for i in range(nres):
        cis = use_cis_proline and res_type[i] == "IPR"
        if i not in chain_starts:
            wn = (-0.2094, 0.6908, 0.5190) if cis else (0.48318, 0.70328, -0.18643)
            system.setVirtualSite(n[i], ThreeParticleAverageSite(ca[i - 1], ca[i], o[i - 1], *wn))
            if not res_type[i] == "IPR":
                system.setVirtualSite(h[i], ThreeParticleAverageSite(ca[i - 1], ca[i], o[i - 1], 0.84100, 0.89296, -0.73389))
        if i not in chain_ends:
            wc = (0.2196, 0.2300, 0.5507) if cis else (0.44365, 0.23520, 0.32115)
            system.setVirtualSite(c[i], ThreeParticleAverageSite(ca[i], ca[i + 1], o[i], *wc))


def setup_bonds(nres, n, h, ca, c, o, cb, res_type, chain_starts, chain_ends):
    """openAWSEM.py:698-715: the bond list behind excl.createExclusionsFromBonds(bonds, 1)."""
    bonds = []
    for i in range(nres):
        bonds.append((ca[i], o[i]))
        if not res_type[i] == "IGL":
            bonds.append((ca[i], cb[i]))
        if i not in chain_ends:
            bonds.append((ca[i], ca[i + 1]))
            bonds.append((o[i], ca[i + 1]))
    for i in range(nres):
        if i not in chain_starts and not res_type[i] == "IGL":
            bonds.append((n[i], cb[i]))
        if i not in chain_ends and not res_type[i] == "IGL":
            bonds.append((c[i], cb[i]))
        if i not in chain_starts and i not in chain_ends:
            bonds.append((n[i], c[i]))
    return bonds


def read_fasta(fastaFile):
    """openAWSEM.py:717-727 / helperFunctions/myFunctions.py:1143-1160: concatenate the non-'>' lines."""
    data = ""
    with open(fastaFile) as fh:
        for line in fh:
            if line[0] == ">" or line == "\n":
                continue
            data += line.strip("\n").strip()
    return data


def getAllChains(pdbFile, removeDNAchains=True):
    """helperFunctions/myFunctions.py:799-826 without pdbfixer: chain IDs of protein ATOM records, in file order."""
    protein = set(three_to_one) | {"NGP", "IGL", "IPR"}
    out = ""
    with open(pdbFile) as fh:
        for line in fh:
            if line.startswith("ENDMDL"):
                break
            if line.startswith("ATOM") and line[17:20].strip() in protein:
                cid = line[21]
                if cid not in out and cid in 'ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789':
                    out += cid
    return out


def getSeqFromCleanPdb(input_pdb_filename, chains='A'):
    """Sequence from `<name>-cleaned.pdb` next to `<name>-openmmawsem.pdb` (openAWSEM.py:738-764)."""
    cleaned = str(input_pdb_filename).replace("openmmawsem.pdb", "cleaned.pdb")
    seq_by_chain = {}
    seen = set()
    with open(cleaned) as fh:
        for line in fh:
            if line.startswith("ENDMDL"):
                break
            if not line.startswith("ATOM"):
                continue
            key = (line[21], line[22:27])
            if key in seen:
                continue
            seen.add(key)
            seq_by_chain.setdefault(line[21], []).append(three_to_one.get(line[17:20].strip(), 'A'))
    return "".join("".join(seq_by_chain.get(ch, [])) for ch in chains)


# --------------------------------------------------------------------------- the system object
class OpenMMAWSEMSystem:
    """Drop-in for openawsem.OpenMMAWSEMSystem (openAWSEM.py:819-929)."""

    def __init__(self, pdb_filename, chains='A', xml_filename=xml, k_awsem=1.0, seqFromPdb=None, includeLigands=False,
                 periodic=False):
        if includeLigands:
            raise NotImplementedError("includeLigands=True is outside the MI355X AWSEM hot path")
        if periodic:
            raise NotImplementedError("periodic=True is not supported by the MI355X AWSEM engine")
        self.pdb = PDBFile(str(pdb_filename))
        self.periodic = periodic
        top = self.pdb.topology
        self.nres = top.getNumResidues()
        self.natoms = top.getNumAtoms()
        self.residues = list(top.residues())
        atoms = list(top.atoms())
        self.resi = [a.residue.index for a in atoms]
        masses_by_type = read_forcefield_masses(xml_filename)
        for a in atoms:
            if a.name not in masses_by_type:
                raise ValueError(f"atom name {a.name!r} (residue {a.residue.name} {a.residue.id}) is not an AWSEM atom type")
        # forcefield.createSystem(topology) (reference openAWSEM.py:826,878): removeCMMotion defaults to True -> CMMotionRemover
        self.system = System([masses_by_type[a.name] for a in atoms], remove_cm_motion=True)
        self.atom_lists, self.res_type = build_lists_of_atoms_2(self.nres, self.residues, atoms)
        self.n = self.atom_lists['n']
        self.h = self.atom_lists['h']
        self.ca = self.atom_lists['ca']
        self.c = self.atom_lists['c']
        self.o = self.atom_lists['o']
        self.cb = self.atom_lists['cb']
        self.chain_starts, self.chain_ends = get_chain_starts_and_ends(self.residues)
        setup_virtual_sites(self.nres, self.system, self.n, self.h, self.ca, self.c, self.o, self.cb, self.res_type,
                            self.chain_starts, self.chain_ends)
        self.bonds = setup_bonds(self.nres, self.n, self.h, self.ca, self.c, self.o, self.cb, self.res_type,
                                 self.chain_starts, self.chain_ends)
        self.k_awsem = k_awsem
        self.force_names = []
        if seqFromPdb is None:
            self.seq = getSeqFromCleanPdb(pdb_filename, chains=chains)
        else:
            self.seq = seqFromPdb
        if len(self.seq) != self.nres:
            raise ValueError(f"sequence length {len(self.seq)} does not match the {self.nres} residues of {pdb_filename}")

    def addForces(self, forces):
        for i, force in enumerate(forces):
            self.addForce(force)
            force.setForceGroup(i + 1)

    def addForce(self, force):
        self.system.addForce(force)

    def addForcesWithDefaultForceGroup(self, forces):
        for force in forces:
            self.addForce(force)
