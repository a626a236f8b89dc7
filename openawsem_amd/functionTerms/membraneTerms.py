"""Membrane slab terms (CustomExternalForce in the reference).

Mirrors membrane_term (openawsem/functionTerms/membraneTerms.py:12-54) and membrane_preassigned_term (115-156):
E = k * scale_i * (1/2 tanh(k_m((z-c)+z_m)) + 1/2 tanh(k_m(z_m-(z-c)))), and membrane_preassigned_side_term (177-215):
E = k |z - c - z_m,i|.
"""
from __future__ import annotations

import numpy as np

from .. import unit as _u
from ..forces import EXT_ABS_Z, ExternalForce, MembraneForce


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


def SideToZ_m(side):
    """'down' / 'up' / 'middle' -> -1.5 / 1.5 / 0 nm (reference membraneTerms.py:159-174; anything else gives None)."""
    side = side.strip()
    if side == "down":
        return -1.5
    if side == "up":
        return 1.5
    if side == "middle":
        return 0
    return None


def membrane_preassigned_side_term(oa, k=1 * _u.kilocalorie_per_mole, membrane_center=0 * _u.angstrom,
                                   zimFile="PredictedZimSide", forceGroup=24):
    """Pulls the CB (CA for Gly) site of every residue towards the plane its line of `zimFile` names: E = k |z - c - z_m|
    with z_m = -1.5 / 0 / 1.5 nm for down / middle / up (reference membraneTerms.py:177-215)."""
    membrane_center = membrane_center.value_in_unit(_u.nanometer)
    k = k.value_in_unit(_u.kilojoule_per_mole)
    k_membrane = k * oa.k_awsem
    membrane = ExternalForce(EXT_ABS_Z, k_membrane, membrane_center)
    with open(zimFile) as f:
        a = f.readlines()
    cb_fixed = [x if x > 0 else y for x, y in zip(oa.cb, oa.ca)]
    for i in cb_fixed:
        z_m = SideToZ_m(a[oa.resi[i]])
        if z_m is None:
            raise ValueError(f"{zimFile}: line {oa.resi[i] + 1} is not one of down / up / middle")
        membrane.addParticle(i, [z_m])
    membrane.setForceGroup(forceGroup)
    return membrane
