"""Device-resident output: ga_render_device / ga_process_blocks*(out_on_device=1) on a caller-provided HIP stream
(what the multi-GPU path of bench.py uses before the RCCL reduce)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G

SR = 48000


def _torch():
    torch = pytest.importorskip("torch")
    try:
        torch.zeros(1, device="cuda")
    except Exception as e:   # e.g. a second HIP runtime in the process (see tests/conftest.py): nothing to test against
        pytest.skip(f"torch cannot use the GPU in this process: {e}")
    return torch


def _ctx():
    ctx = OfflineAudioContext(SR)
    ch = G.config3_convolver(ctx, voices=6, taps=3000, frames=128 * 40)
    return ctx, ch


def test_render_device_on_a_torch_stream_matches_render():
    torch = _torch()
    ref_ctx, ch = _ctx()
    ref = G.render(ref_ctx, ch, 128 * 32)
    ctx, _ = _ctx()
    stream = torch.cuda.Stream()
    ctx.SetStream(stream.cuda_stream)
    with torch.cuda.stream(stream):
        out = torch.zeros((ch, 128 * 32), dtype=torch.float32, device="cuda")
        ctx.RenderDevice([out[c].data_ptr() for c in range(ch)], 128 * 20 + 5, 0)
        ctx.RenderDevice([out[c].data_ptr() for c in range(ch)], 128 * 12 - 5, 128 * 20 + 5)
        doubled = out * 2.0          # consumer on the same stream: ordered after the render without a host sync
    stream.synchronize()
    got = out.cpu().numpy()
    assert np.array_equal(ref, got)
    assert np.array_equal(doubled.cpu().numpy(), ref * 2.0)


def test_process_blocks_into_device_memory():
    torch = _torch()
    ref_ctx, ch = _ctx()
    ref = G.render(ref_ctx, ch, 128 * 16)
    ctx, _ = _ctx()
    planar = torch.zeros((ch, 128 * 16), dtype=torch.float32, device="cuda")
    ptrs = (C.c_void_p * ch)(*[planar[c].data_ptr() for c in range(ch)])
    ctx._call("process_blocks", ptrs, ch, 16, 1)
    assert np.array_equal(planar.cpu().numpy(), ref)
    ctx2, _ = _ctx()
    inter = torch.zeros(128 * 16 * ch, dtype=torch.float32, device="cuda")
    ctx2._call("process_blocks_interleaved", C.cast(C.c_void_p(inter.data_ptr()), C.POINTER(C.c_float)), ch, 16, 1)
    assert np.array_equal(inter.cpu().numpy().reshape(128 * 16, ch).T, ref)
