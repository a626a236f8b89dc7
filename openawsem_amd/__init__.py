"""openawsem_amd: MI355X-native AWSEM force-and-integrator engine behind the openawsem Python API.

Host-side mirror of the reference package surface that the hot path needs (openawsem/__init__.py,
openawsem/openAWSEM.py): ``OpenMMAWSEMSystem``, ``functionTerms``, ``data_path``, ``xml`` and the unit names.
The arithmetic runs in libawsem_hip.so (csrc/, C ABI in include/awsem_hip.h) -- there is no CPU fallback.
`install_as_openawsem()` registers this package under the names ``openawsem`` / ``openmm`` so that an
unmodified user ``forces_setup.py`` (openawsem/scripts/forces_setup.py) imports and runs against it.
"""
from __future__ import annotations

import sys
from pathlib import Path

__location__ = Path(__file__).resolve().parent
__version__ = "0.1.0"


class _DataPath:
    """The two data locations the hot path resolves (openawsem/helperFunctions/DataPath.py, config.ini)."""

    def __init__(self, root):
        self.default_location = root
        self.parameters = root / "parameters"
        self.topology = root / "topology"
        self.data = root / "data"


data_path = _DataPath(__location__)

from . import unit  # noqa: E402
from . import unit as unit_definitions  # noqa: E402,F401
from .unit import *  # noqa: E402,F401,F403
from .openAWSEM import (OpenMMAWSEMSystem, PDBFile, getAllChains, read_fasta, xml,  # noqa: E402,F401
                        three_to_one, one_to_three)
from . import functionTerms  # noqa: E402,F401
from .functionTerms import (basicTerms, biasTerms, contactTerms, debyeHuckelTerms, hydrogenBondTerms, membraneTerms,  # noqa: E402,F401
                            templateTerms)


def install_as_openawsem(force: bool = False):
    """Make ``import openawsem`` (and, when OpenMM is absent, ``import openmm`` / ``openmm.unit`` /
    ``openmm.app``) resolve to this package, so user forces files written for the reference run unmodified."""
    import importlib
    import types

    if force or "openawsem" not in sys.modules:
        me = sys.modules[__name__]
        sys.modules["openawsem"] = me
        for sub in ("functionTerms", "openAWSEM", "unit", "forces"):
            sys.modules[f"openawsem.{sub}"] = importlib.import_module(f"{__name__}.{sub}")
        for sub in ("basicTerms", "biasTerms", "contactTerms", "debyeHuckelTerms", "hydrogenBondTerms", "membraneTerms", "templateTerms"):
            sys.modules[f"openawsem.functionTerms.{sub}"] = importlib.import_module(f"{__name__}.functionTerms.{sub}")
        helper = types.ModuleType("openawsem.helperFunctions")
        myf = types.ModuleType("openawsem.helperFunctions.myFunctions")
        from . import openAWSEM as _oa
        for name in ("read_fasta", "getAllChains", "three_to_one", "one_to_three"):
            setattr(myf, name, getattr(_oa, name))
        helper.myFunctions = myf
        sys.modules["openawsem.helperFunctions"] = helper
        sys.modules["openawsem.helperFunctions.myFunctions"] = myf
        me.helperFunctions = helper
    try:
        import openmm  # noqa: F401
    except ModuleNotFoundError:
        from . import engine, forces
        omm = types.ModuleType("openmm")
        for name in ("LangevinIntegrator", "CustomIntegrator", "Platform", "Context", "State"):
            setattr(omm, name, getattr(engine, name))
        omm.Force = forces.Force
        omm.System = forces.System
        omm.unit = unit
        app = types.ModuleType("openmm.app")
        for name in ("Simulation", "DCDReporter", "StateDataReporter", "CheckpointReporter", "PDBReporter"):
            setattr(app, name, getattr(engine, name))
        app.PDBFile = PDBFile
        omm.app = app
        sys.modules["openmm"] = omm
        sys.modules["openmm.unit"] = unit
        sys.modules["openmm.app"] = app
