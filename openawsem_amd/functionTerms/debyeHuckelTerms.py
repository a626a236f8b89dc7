"""Debye-Hueckel screened Coulomb term between charged residues.

Mirrors debye_huckel_term of openawsem/functionTerms/debyeHuckelTerms.py:10-76 (a CustomBondForce over an explicit
pair list): E = k_dh q_i q_j / r exp(-r / screening_length), k_dh = 4.15*4.184*0.1*k_awsem.
"""
from __future__ import annotations

import numpy as np

from ..forces import DebyeHuckelForce


def debye_huckel_term(self, k_dh=4.15 * 4.184, forceGroup=30, screening_length=1.0, chargeFile=None):
    """`self` is the OpenMMAWSEMSystem (the reference names its first argument that way)."""
    print("Debye Huckel term is ON")
    k_dh *= self.k_awsem * 0.1
    k_screening = 1.0
    min_seq_sep = 1
    dh = DebyeHuckelForce(k_dh, screening_length / k_screening)
    if chargeFile is None:
        charges = [1.0 if s in "RK" else (-1.0 if s in "DE" else 0.0) for s in self.seq[:self.nres]]
    else:
        chargeInfo = np.loadtxt(chargeFile, dtype=[('index', int), ('charge', float)])
        charges = [float(chargeInfo[i][1]) for i in range(self.nres)]
    site = [self.cb[i] if self.cb[i] != -1 else self.ca[i] for i in range(self.nres)]
    for i in range(self.nres):
        for j in range(i + min_seq_sep, self.nres):
            if charges[i] * charges[j] != 0.0:
                dh.addBond(site[i], site[j], [charges[i], charges[j]])
    dh.setForceGroup(forceGroup)
    return dh
