"""Minimal stand-in for ``openmm.unit`` covering what openawsem user files use.

The reference's forces files pass ``openmm.unit`` quantities into the term factories
(``k=1*kilocalorie_per_mole``, ``membrane_center=0*angstrom``: openawsem/scripts/forces_setup.py:18,56;
conversions at openawsem/functionTerms/hydrogenBondTerms.py:396-415, contactTerms.py:140-149) and the drivers
build ``Tstart*kelvin``, ``1/picosecond``, ``2*femtoseconds`` (openawsem/scripts/mm_run.py:125,145).
Only multiplication/division by numbers and other units and ``value_in_unit`` are needed.  Base units are
OpenMM's: nm, ps, kJ/mol, K, amu.
"""
from __future__ import annotations

import math


class Unit:
    """A unit = scale factor to base units x a dimension vector (length, time, energy, temperature, mass)."""

    __slots__ = ("name", "scale", "dims")

    def __init__(self, name, scale, dims):
        self.name = name
        self.scale = float(scale)
        self.dims = tuple(dims)

    def __repr__(self):
        return f"Unit({self.name})"

    def _combine(self, other, sign):
        dims = tuple(a + sign * b for a, b in zip(self.dims, other.dims))
        scale = self.scale * (other.scale ** sign)
        name = f"{self.name}{'*' if sign > 0 else '/'}{other.name}"
        return Unit(name, scale, dims)

    def __mul__(self, other):
        if isinstance(other, Unit):
            return self._combine(other, +1)
        if isinstance(other, Quantity):
            return Quantity(other._value, self * other.unit)
        return Quantity(other, self)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Unit):
            return self._combine(other, -1)
        if isinstance(other, Quantity):
            return Quantity(1.0 / other._value, self / other.unit)
        return Quantity(1.0 / other, self)

    def __rtruediv__(self, other):
        inv = Unit(f"1/{self.name}", 1.0 / self.scale, tuple(-d for d in self.dims))
        if isinstance(other, Quantity):
            return Quantity(other._value, other.unit * inv)
        return Quantity(other, inv)

    def __pow__(self, p):
        return Unit(f"{self.name}**{p}", self.scale ** p, tuple(d * p for d in self.dims))

    def is_compatible(self, other):
        return all(abs(a - b) < 1e-12 for a, b in zip(self.dims, other.dims))


class Quantity:
    __slots__ = ("_value", "unit")

    def __init__(self, value, unit=None):
        if unit is None:
            unit = dimensionless
        self._value = value
        self.unit = unit

    def value_in_unit(self, unit):
        if not self.unit.is_compatible(unit):
            raise TypeError(f"Unit {self.unit.name} is not compatible with {unit.name}")
        return self._value * (self.unit.scale / unit.scale)

    def in_units_of(self, unit):
        return Quantity(self.value_in_unit(unit), unit)

    def __repr__(self):
        return f"Quantity(value={self._value!r}, unit={self.unit.name})"

    def __mul__(self, other):
        if isinstance(other, Quantity):
            return Quantity(self._value * other._value, self.unit * other.unit)
        if isinstance(other, Unit):
            return Quantity(self._value, self.unit * other)
        return Quantity(self._value * other, self.unit)

    __rmul__ = __mul__

    def __truediv__(self, other):
        if isinstance(other, Quantity):
            return Quantity(self._value / other._value, self.unit / other.unit)
        if isinstance(other, Unit):
            return Quantity(self._value, self.unit / other)
        return Quantity(self._value / other, self.unit)

    def __rtruediv__(self, other):
        return Quantity(other / self._value, dimensionless / self.unit)

    def __neg__(self):
        return Quantity(-self._value, self.unit)

    def __add__(self, other):
        if isinstance(other, Quantity):
            return Quantity(self._value + other.value_in_unit(self.unit), self.unit)
        raise TypeError("cannot add a Quantity and a plain number")

    def __sub__(self, other):
        if isinstance(other, Quantity):
            return Quantity(self._value - other.value_in_unit(self.unit), self.unit)
        raise TypeError("cannot subtract a plain number from a Quantity")

    def _cmp_value(self, other):
        if isinstance(other, Quantity):
            return other.value_in_unit(self.unit)
        raise TypeError("cannot compare a Quantity with a plain number")

    def __lt__(self, other):
        return self._value < self._cmp_value(other)

    def __gt__(self, other):
        return self._value > self._cmp_value(other)

    def __eq__(self, other):
        return isinstance(other, Quantity) and self.unit.is_compatible(other.unit) and \
            math.isclose(self._value, other.value_in_unit(self.unit))

    def __hash__(self):
        return hash((self._value, self.unit.dims))


def is_quantity(x):
    return isinstance(x, Quantity)


#                         name            scale  (L, T, E, K, M)
dimensionless = Unit("dimensionless", 1.0, (0, 0, 0, 0, 0))
nanometer = nanometers = Unit("nanometer", 1.0, (1, 0, 0, 0, 0))
angstrom = angstroms = Unit("angstrom", 0.1, (1, 0, 0, 0, 0))
picosecond = picoseconds = Unit("picosecond", 1.0, (0, 1, 0, 0, 0))
femtosecond = femtoseconds = Unit("femtosecond", 1e-3, (0, 1, 0, 0, 0))
nanosecond = nanoseconds = Unit("nanosecond", 1e3, (0, 1, 0, 0, 0))
kilojoule_per_mole = kilojoules_per_mole = Unit("kilojoule/mole", 1.0, (0, 0, 1, 0, 0))
kilocalorie_per_mole = kilocalories_per_mole = Unit("kilocalorie/mole", 4.184, (0, 0, 1, 0, 0))
kelvin = kelvins = Unit("kelvin", 1.0, (0, 0, 0, 1, 0))
amu = dalton = daltons = Unit("dalton", 1.0, (0, 0, 0, 0, 1))

# OpenMM's BOLTZ (kJ/mol/K); the MOLAR_GAS_CONSTANT_R of openmm.unit
MOLAR_GAS_CONSTANT_R = Quantity(0.00831446261815324, kilojoule_per_mole / kelvin)
BOLTZMANN_KJ_PER_MOL_K = 0.00831446261815324
