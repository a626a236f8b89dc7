"""CPU-side checks of the C-ABI library: it loads without a GPU, exports every symbol include/awsem_hip.h declares,
and fails loudly (no CPU fallback) when asked to create a context without a device."""
import ctypes
import os
import re

import pytest

from openawsem_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "awsem_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(awsem_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert set(_declared_symbols()) == set(_capi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_capi.HIP_LIBRARY_PATH), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(_capi.HIP_LIBRARY_PATH)
    for sym in _declared_symbols():
        assert hasattr(lib, sym), sym


def test_oracle_exports_the_same_abi(oracle_api):
    hip_only = {"awsem_" + n for n in _capi._HIP_ONLY}       # entry points about device memory: nothing for a CPU restatement to mirror
    for sym in _declared_symbols():
        if sym not in hip_only:
            assert hasattr(oracle_api.lib, sym.replace("awsem_", "awsem_oracle_", 1)), sym


def test_no_cpu_fallback_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    api = _capi.load_hip_library()
    assert "gfx950" in api.version()
    with pytest.raises(_capi.AwsemError) as err:
        _capi.Context(api, 10, 1)
    assert err.value.code == -3 and "no CPU fallback" in str(err.value)
