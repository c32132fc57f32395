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


def test_csharp_binding_covers_the_header():
    """bindings/csharp/GraphAudioHip.cs: one P/Invoke stub per entry point of the header (what a GraphAudio maintainer binds)"""
    cs = open(os.path.join(ROOT, "bindings", "csharp", "GraphAudioHip.cs")).read()
    bound = set(re.findall(r'EntryPoint = "ga_(\w+)"', cs))
    assert sorted(declared_symbols()) == sorted(bound)


def test_ga_stats_layout_is_the_same_in_c_python_and_csharp(tmp_path):
    """ga_get_stats writes the whole struct: the three declarations have to agree in size"""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "graphaudio_hip.h"\nint main(void) { printf("%zu", sizeof(ga_stats)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    c_size = int(subprocess.check_output([str(exe)]).decode())
    assert C.sizeof(_capi.Stats) == c_size
    cs = open(os.path.join(ROOT, "bindings", "csharp", "GraphAudioHip.cs")).read()
    body = cs[cs.index("public unsafe struct Stats"):]
    body = body[:body.index("}")]
    size = 0
    for decl in re.findall(r"public\s+(fixed\s+)?(long|double|int|byte)\s+([^;]+);", body):
        width = {"long": 8, "double": 8, "int": 4, "byte": 1}[decl[1]]
        for name in decl[2].split(","):
            m = re.search(r"\[(\d+)\]", name)
            size += width * (int(m.group(1)) if m else 1)
    assert size == c_size


def test_stage_name_table_covers_every_stage_of_the_header():
    """Stats.STAGES is zipped with stage_ms / stage_bytes / stage_kernel by consumers -- a missing name drops a stage
    (GA_STAGE_COARSE_PREMIX, the headline's dominant kernel, was missing)."""
    hdr = open(os.path.join(ROOT, "include", "graphaudio_hip.h")).read()
    count = int(re.search(r"GA_STAGE_COUNT\s*=\s*(\d+)", hdr).group(1))
    assert len(_capi.Stats.STAGES) == count
    names = re.findall(r"GA_STAGE_(\w+)\s*=\s*(\d+)", hdr)
    for name, idx in names:
        if name != "COUNT":
            assert _capi.Stats.STAGES[int(idx)] == name.lower().replace("rfft_fwd", "rfft_fwd"), (name, idx)
