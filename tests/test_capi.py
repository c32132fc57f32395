"""The C-ABI library loads and exports every symbol include/graphaudio_hip.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import pytest

from graphaudio_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "graphaudio_hip.h")).read()
    names = re.findall(r"GA_EXPORT[^;]*?GA_FN\((\w+)\)", text, flags=re.S)
    assert len(names) >= 40
    return names


def test_header_and_binding_table_agree():
    assert sorted(declared_symbols()) == sorted(_capi.SIGNATURES)


def test_product_library_exports_every_declared_symbol():
    path = _capi.library_path()
    assert os.path.exists(path), "run __graft_entry__.build() first"
    lib = C.CDLL(path)
    for name in declared_symbols():
        assert hasattr(lib, "ga_" + name), name


def test_oracle_exports_same_surface():
    from tests._oracle import oracle_lib
    lib = oracle_lib()
    for name in declared_symbols():
        assert hasattr(lib, "gao_" + name), name


def test_no_cpu_fallback_without_device():
    api = _capi.product_api()
    assert api.version().startswith(b"graphaudio-hip")
    if api.device_count() > 0:
        pytest.skip("a GPU is present")
    from graphaudio_amd import DeviceException, OfflineAudioContext
    with pytest.raises(DeviceException):
        OfflineAudioContext(48000)


def test_error_strings():
    api = _capi.product_api()
    assert api.strerror(0) == b"ok"
    assert b"range" in api.strerror(-2)
