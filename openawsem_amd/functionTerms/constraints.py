"""Distance restraints (openawsem/functionTerms/constraints.py).

`constraint_by_distance` (reference :13-41) is CustomBondForce("0.5*k*(r-d0)^2") on the CA atoms of two residues: a
harmonic bond, evaluated by the engine's harmonic-bond kernel.  The centroid-based group constraints of the reference
(CustomCentroidBondForce / CustomCVForce, :43-141) are collective-variable forces outside the hot-path term set and
raise NotImplementedError."""
from __future__ import annotations

from .. import unit as u
from ..forces import HarmonicBondForce


def _value(x, unit_):
    return float(x.value_in_unit(unit_)) if hasattr(x, "value_in_unit") else float(x)


def constraint_by_distance(oa, res1, res2, d0=0 * u.angstrom, forceGroup=3, k=1 * u.kilocalorie_per_mole):
    """Harmonic restraint 0.5 k (r - d0)^2 between CA of residue `res1` and CA of residue `res2` (0-based)."""
    k_constraint = _value(k, u.kilojoule_per_mole) * oa.k_awsem
    force = HarmonicBondForce()
    force.addBond(oa.ca[res1], oa.ca[res2], _value(d0, u.nanometer), k_constraint)
    force.setForceGroup(forceGroup)
    return force


def group_constraint_by_distance(*args, **kwargs):
    raise NotImplementedError("group_constraint_by_distance (CustomCentroidBondForce) is not part of the MI355X engine's term set")


def group_constraint_by_position(*args, **kwargs):
    raise NotImplementedError("group_constraint_by_position (CustomCVForce) is not part of the MI355X engine's term set")
