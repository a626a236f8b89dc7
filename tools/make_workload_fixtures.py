#!/usr/bin/env python3
"""Generates the small data fixtures of the T0958 and 1r69-many-memory workloads from the reference's example
folders (run once in the build container; the outputs under openawsem_amd/data are committed, this script is the
record of how they were made -- nothing reads /root/reference at run time).

  * T0958: examples/T0958 ships no fragment library (0 of the 1760 .gro files its frags.mem names exist), so the
    workload uses a single memory made from its own native structure: t0958-cleaned.pdb -> t0958_A.gro (backbone
    + CB atoms, nm) and single_frags.mem.
  * 1r69 many-memory stress: examples/1r69/frags.mem (1100 memories over 614 .gro files, 71 MB).  Only the CA/CB
    atoms of the residue window a memory names are ever read (templateTerms.py:192-193 of the reference), so each
    .gro is trimmed to those lines; the file format (2 header lines, whitespace columns) is unchanged.
"""
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "openawsem_amd", "data")


def pdb_to_gro(pdb, gro, keep=("N", "CA", "C", "O", "CB")):
    rows, res_ids = [], {}
    for line in open(pdb):
        if line.startswith("ENDMDL"):
            break
        if not line.startswith("ATOM"):
            continue
        name = line[12:16].strip()
        if name not in keep:
            continue
        key = (line[21], line[22:27])
        rid = res_ids.setdefault(key, len(res_ids) + 1)
        x, y, z = (float(line[30:38]) / 10, float(line[38:46]) / 10, float(line[46:54]) / 10)
        rows.append((rid, line[17:20], name, len(rows) + 1, x, y, z))
    with open(gro, "w") as fh:
        fh.write(" Structure-Based gro file\n%12d\n" % len(rows))
        for r in rows:
            fh.write("%5d  %-3s %-4s %5d %8.3f %8.3f %8.3f\n" % r)
    return len(res_ids)


def t0958():
    d = os.path.join(OUT, "T0958")
    n = pdb_to_gro(os.path.join(REF, "examples/T0958/t0958-cleaned.pdb"), os.path.join(d, "t0958_A.gro"))
    with open(os.path.join(d, "single_frags.mem"), "w") as fh:
        fh.write("[Target]\nquery\n\n[Memories]\nt0958_A.gro 1 1 %d 20\n" % n)
    print("T0958:", n, "residues")


def lzm():
    """The chain the synthetic multi-chain system (BASELINE.json configs[4]) is tiled from: the reference's own 2lzm test structure."""
    import shutil
    d = os.path.join(OUT, "2lzm")
    os.makedirs(d, exist_ok=True)
    for name in ("2lzm-openmmawsem.pdb", "2lzm-crystal_structure.fasta"):
        shutil.copy(os.path.join(REF, "tests/data", name), os.path.join(d, name))
    print("2lzm: copied")


def r69_library():
    src = os.path.join(REF, "examples/1r69")
    d = os.path.join(OUT, "1r69")
    os.makedirs(os.path.join(d, "fraglib"), exist_ok=True)
    lines = open(os.path.join(src, "frags.mem")).read().split("\n")
    windows = {}
    for line in lines[4:]:
        t = line.split()
        if len(t) == 5:
            windows.setdefault(t[0], []).append((int(t[2]), int(t[2]) + int(t[3])))
    open(os.path.join(d, "frags.mem"), "w").write("\n".join(lines))
    total = 0
    for rel, wins in windows.items():
        glines = open(os.path.join(src, rel)).read().split("\n")
        keep = []
        for g in glines[2:]:
            t = g.split()
            if len(t) < 7 or t[2] not in ("CA", "CB"):
                continue
            try:
                rid = int(t[0])
            except ValueError:
                continue
            if any(a <= rid < b for a, b in wins):
                keep.append(g)
        out = os.path.join(d, rel)
        with open(out, "w") as fh:
            fh.write(glines[0] + "\n%12d\n" % len(keep) + "\n".join(keep) + "\n")
        total += os.path.getsize(out)
    print("1r69 library:", len(windows), "files,", total, "bytes")


if __name__ == "__main__":
    t0958()
    lzm()
    r69_library()
