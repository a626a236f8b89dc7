"""Order parameters used by awsem_analyze: Q (fraction of native contacts) and Rg.

Mirrors q_value (openawsem/functionTerms/biasTerms.py:146-188, reference reader :13-87) and rg_term (:461-486).
In the reference these are OpenMM forces whose "energy" is the order parameter (force groups 1 and 2, printed by
scripts/mm_analyze.py:139-182).  Here they are analysis-only descriptors evaluated on the host from the
coordinates the engine returns (`host_only`); they exert no force in MD, so adding one to a system that is then
stepped raises instead of silently ignoring the bias.
"""
from __future__ import annotations

import numpy as np

from .. import unit as _u
from ..forces import Force


class HostOnlyForce(Force):
    host_only = True

    def _upload(self, ctx):
        raise NotImplementedError(f"{type(self).__name__} is an analysis-only order parameter, not a dynamics term")

    def host_energy(self, x_nm):
        raise NotImplementedError


class QValueForce(HostOnlyForce):
    kind = "q_value"

    def __init__(self, pairs, r_native, sigma, group=1):
        super().__init__(group)
        self.pairs = np.asarray(pairs, dtype=np.int64).reshape(-1, 2)
        self.r_native, self.sigma = np.asarray(r_native, float), np.asarray(sigma, float)

    def host_energy(self, x):
        if len(self.pairs) == 0:
            return 0.0
        r = np.linalg.norm(x[self.pairs[:, 0]] - x[self.pairs[:, 1]], axis=1)
        return float(np.exp(-(r - self.r_native) ** 2 / (2 * self.sigma ** 2)).sum() / len(self.pairs))


class RgForce(HostOnlyForce):
    kind = "rg"

    def __init__(self, ca, unit=10.0, group=2):
        super().__init__(group)
        self.ca, self.unit = np.asarray(ca, dtype=np.int64), float(unit)

    def host_energy(self, x):
        c = x[self.ca]
        n = len(c)
        d2 = ((c[:, None, :] - c[None, :, :]) ** 2).sum(-1)
        return float(self.unit * np.sqrt(np.triu(d2, 1).sum() / (n * n)))


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


def q_value(oa, reference_pdb_file, reference_chain_name="ALL", min_seq_sep=3, max_seq_sep=np.inf,
            contact_threshold=0.95 * _u.nanometers, forceGroup=1):
    """Canonical Q_w: mean over intra-chain CA pairs with |i-j| >= min_seq_sep (both orders, as the reference's
    double loop produces them) of exp(-(r - r_native)^2 / (2 sigma_ij^2)), sigma_ij = 0.1 |i-j|^0.15 nm."""
    pairs, rn, sg = [], [], []
    chain_start = 0
    for cid, xyz in _reference_ca_by_chain(reference_pdb_file):
        n = len(xyz)
        for i in range(n):
            for j in range(n):
                sep = abs(i - j)
                if sep >= min_seq_sep and sep <= max_seq_sep:
                    if reference_chain_name != "ALL" and cid not in reference_chain_name:
                        continue
                    pairs.append((oa.ca[i + chain_start], oa.ca[j + chain_start]))
                    rn.append(np.linalg.norm(xyz[i] - xyz[j]) / 10.0)
                    sg.append(0.1 * sep ** 0.15)
        chain_start += n
    return QValueForce(pairs, rn, sg, forceGroup)


def rg_term(oa, convertToAngstrom=True):
    """Radius of gyration of the CA trace, sqrt(sum_{i<j} r_ij^2 / nres^2), in Angstrom by default (group 2)."""
    return RgForce(oa.ca, 10.0 if convertToAngstrom else 1.0, 2)
