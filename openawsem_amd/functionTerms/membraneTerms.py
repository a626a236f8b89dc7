"""Membrane slab terms (CustomExternalForce in the reference).

Mirrors membrane_term (openawsem/functionTerms/membraneTerms.py:12-54) and membrane_preassigned_term (115-156):
E = k * scale_i * (1/2 tanh(k_m((z-c)+z_m)) + 1/2 tanh(k_m(z_m-(z-c)))).
"""
from __future__ import annotations

import numpy as np

from .. import unit as _u
from ..forces import MembraneForce


def membrane_term(oa, k=1 * _u.kilocalorie_per_mole, k_m=20, z_m=1.5, membrane_center=0 * _u.angstrom, forceGroup=24):
    """Hydrophobicity-scaled slab potential on every CA; reads ``zim`` (one value per residue) from the working
    directory like the reference (membraneTerms.py:47)."""
    membrane_center = membrane_center.value_in_unit(_u.nanometer)
    k = k.value_in_unit(_u.kilojoule_per_mole)
    k_membrane = k * oa.k_awsem
    membrane = MembraneForce(k_membrane, k_m, z_m, membrane_center)
    zim = np.atleast_1d(np.loadtxt("zim"))
    for i in oa.ca:
        membrane.addParticle(i, [zim[oa.resi[i]]])
    membrane.setForceGroup(forceGroup)
    return membrane


def membrane_preassigned_term(oa, k=1 * _u.kilocalorie_per_mole, k_m=20, z_m=1.5, membrane_center=0 * _u.angstrom,
                              zimFile="zimPosition", forceGroup=24):
    """Slab potential on the CB (CA for Gly) sites with a preassigned sign: -1 where the zim position file says 2
    (membrane), +1 elsewhere (reference membraneTerms.py:115-156)."""
    membrane_center = membrane_center.value_in_unit(_u.nanometer)
    k = k.value_in_unit(_u.kilojoule_per_mole)
    k_membrane = k * oa.k_awsem
    membrane = MembraneForce(k_membrane, k_m, z_m, membrane_center)
    zimPosition = np.atleast_1d(np.loadtxt(zimFile))
    zim = [-1 if z == 2 else 1 for z in zimPosition]
    cb_fixed = [x if x > 0 else y for x, y in zip(oa.cb, oa.ca)]
    for i in cb_fixed:
        membrane.addParticle(i, [zim[oa.resi[i]]])
    membrane.setForceGroup(forceGroup)
    return membrane
