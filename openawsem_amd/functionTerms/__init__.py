"""Term factories with the reference's module layout (openawsem/functionTerms/__init__.py), so that a user
``forces_setup.py`` doing ``from openawsem.functionTerms import *`` finds ``basicTerms``, ``contactTerms``,
``hydrogenBondTerms``, ``templateTerms``, ``membraneTerms`` and ``debyeHuckelTerms``."""
from . import basicTerms, biasTerms, constraints, contactTerms, debyeHuckelTerms, hydrogenBondTerms, membraneTerms, templateTerms

__all__ = ["basicTerms", "biasTerms", "constraints", "contactTerms", "debyeHuckelTerms", "hydrogenBondTerms", "membraneTerms", "templateTerms"]
