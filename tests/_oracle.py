"""Test-side loader of the CPU oracle.  The oracle is test infrastructure (oracle/ga_oracle.cpp header)."""
import ctypes as C
import os
import subprocess

import numpy as np

from graphaudio_amd._capi import CApi
from graphaudio_amd.core import OfflineAudioContext

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_LIB = os.path.join(ORACLE_DIR, "libga_oracle.so")

_api = None
_lib = None


def build_oracle():
    src = os.path.join(ORACLE_DIR, "ga_oracle.cpp")
    hdr = os.path.join(ROOT, "include", "graphaudio_hip.h")
    if (not os.path.exists(ORACLE_LIB)
            or os.path.getmtime(ORACLE_LIB) < max(os.path.getmtime(src), os.path.getmtime(hdr))):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return ORACLE_LIB


def oracle_lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build_oracle())
        dp = C.POINTER(C.c_double)
        fp = C.POINTER(C.c_float)
        _lib.gao_test_rfft256.argtypes = [dp, dp, dp]
        _lib.gao_test_irfft256.argtypes = [dp, dp, dp]
        _lib.gao_test_normalization_scale.argtypes = [fp, C.c_int]
        _lib.gao_test_normalization_scale.restype = C.c_float
        _lib.gao_test_convolve.argtypes = [fp, C.c_int, C.c_int, fp, fp, C.c_int]
        _lib.gao_test_resample.argtypes = [fp, C.c_int, fp, C.c_int, C.c_double, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    return _lib


def oracle_api() -> CApi:
    global _api
    if _api is None:
        _api = CApi(oracle_lib(), "gao_")
    return _api


def OracleContext(sampleRate=48000) -> OfflineAudioContext:
    """An OfflineAudioContext whose native side is the CPU oracle (same host code, gao_ prefix)."""
    return OfflineAudioContext(sampleRate, _api=oracle_api())


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def rfft256(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    re = np.zeros(129)
    im = np.zeros(129)
    oracle_lib().gao_test_rfft256(_dp(x), _dp(re), _dp(im))
    return re + 1j * im


def irfft256(X):
    re = np.ascontiguousarray(X.real, dtype=np.float64)
    im = np.ascontiguousarray(X.imag, dtype=np.float64)
    x = np.zeros(256)
    oracle_lib().gao_test_irfft256(_dp(re), _dp(im), _dp(x))
    return x


def normalization_scale(ir):
    ir = np.ascontiguousarray(ir, dtype=np.float32)
    return float(oracle_lib().gao_test_normalization_scale(_fp(ir), len(ir)))


def convolve(ir, x, normalize=True):
    ir = np.ascontiguousarray(ir, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=np.float32)
    assert len(x) % 128 == 0
    out = np.zeros_like(x)
    rc = oracle_lib().gao_test_convolve(_fp(ir), len(ir), 1 if normalize else 0, _fp(x), _fp(out), len(x) // 128)
    assert rc == 0
    return out


def resample(x, n_out, rate):
    x = np.ascontiguousarray(x, dtype=np.float32)
    out = np.zeros(n_out, dtype=np.float32)
    c, p = C.c_int(0), C.c_int(0)
    oracle_lib().gao_test_resample(_fp(x), len(x), _fp(out), n_out, float(rate), C.byref(c), C.byref(p))
    return out[:p.value], c.value, p.value
