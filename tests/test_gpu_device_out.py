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


def test_async_renders_match_synchronous_ones():
    """SetOption("async", 1): render calls return once their work is enqueued; the graph is edited between the calls and the
    output (page-locked host memory) is read after Synchronize().  Same kernels, same order -> bit-identical."""
    torch = _torch()
    from tests import _graphs as G
    from graphaudio_amd import AudioBufferSourceNode, ConvolverNode, GainNode, PlayableAudioBuffer

    def run(async_mode):
        ctx = OfflineAudioContext(48000)
        ctx.SetOption("max_chunk_blocks", 7)
        ctx.SetOption("profile", 1)
        if async_mode:
            ctx.SetOption("async", 1)
        frames = 128 * 60
        ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 128 * 70) for c in range(2)], 48000)
        gains = []
        for v in range(6):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(300 + v, frames), 48000)
            cv = ConvolverNode(ctx)
            cv.Buffer = ir if v % 2 else PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 900, seed0=9 + v) for c in range(2)], 48000)
            g = GainNode(ctx)
            g.Gain.Value = 0.5
            s.Connect(cv).Connect(g).Connect(ctx.Destination)
            s.Start(v * 0.003)
            gains.append(g)
        pin = torch.zeros((2, frames), dtype=torch.float32).pin_memory()
        out = pin.numpy()
        pos = 0
        for i, n in enumerate((128 * 9 + 17, 128 * 20 - 17, 128 * 11, 128 * 20)):
            ctx.Render(out, n, pos)
            pos += n
            gains[i].Gain.Value = 0.25 + 0.1 * i
            if i == 1:
                gains[4].Gain.LinearRampToValueAtTime(0.0, pos / 48000 + 0.05)
        ctx.Synchronize()
        st = ctx.GetStats()
        assert st["chunks"] >= 9 and st["device_ms_total"] > 0
        return out.copy()

    a, b = run(False), run(True)
    assert G.rms(a) > 1e-3
    assert np.array_equal(a, b)


# ---- sharded render (include/graphaudio_hip.h "sharded render"): ga_comm_* + ga_render_reduce on the HIP path ----
def _reduce_matches_render(n_ranks_id):
    ref_ctx, ch = _ctx()
    frames = 128 * 32
    ref = G.render(ref_ctx, ch, frames)
    ctx, _ = _ctx()
    assert ctx.CommInfo() == {"ranks": 1, "rank": 0, "uses_rccl": False}   # (before and after ga_comm_init at one rank: no RCCL involved)
    ctx.CommInit(n_ranks_id, 1, 0)
    assert ctx.CommInfo() == {"ranks": 1, "rank": 0, "uses_rccl": False}
    got = np.zeros((ch, frames), np.float32)
    ctx.RenderReduce(got, 128 * 20 + 5, 0)               # uneven pieces: the leftover-frame cache behind the reduce
    ctx.RenderReduce(got, 128 * 12 - 5, 128 * 20 + 5)
    assert np.array_equal(ref, got)
    ctx.CommDestroy()
    ctx.Dispose()


def test_render_reduce_single_rank_equals_render():
    """n_ranks = 1 needs no RCCL: render into device memory, copy to the caller -- bit-identical to ga_render."""
    _reduce_matches_render(None)


def test_render_reduce_requires_comm_init():
    from graphaudio_amd import InvalidOperationException
    ctx, ch = _ctx()
    with pytest.raises(InvalidOperationException):
        ctx.RenderReduce(np.zeros((ch, 256), np.float32), 256)


def test_comm_unique_id_loads_rccl():
    """ga_comm_unique_id loads librccl.so.1 at run time (the product library itself links the HIP runtime only)."""
    ctx, _ = _ctx()
    a, b = ctx.CommUniqueId(), ctx.CommUniqueId()
    assert len(a) == 128 and a != b


def test_render_reduce_pipelined_steps():
    """async contexts: consecutive sharded renders are enqueued back to back (host planning overlaps the device), results after
    Synchronize()."""
    ref_ctx, ch = _ctx()
    frames = 128 * 16
    ref = [G.render(ref_ctx, ch, frames) for _ in range(3)]
    ctx, _ = _ctx()
    ctx.CommInit(None, 1, 0)
    ctx.SetOption("async", 1)
    torch = _torch()
    outs = [torch.zeros((ch, frames), dtype=torch.float32).pin_memory().numpy() for _ in range(3)]
    for o in outs:
        ctx.RenderReduce(o, frames)
    ctx.Synchronize()
    for r, o in zip(ref, outs):
        assert np.array_equal(r, o)


def test_page_locked_host_rows_are_written_by_the_last_kernel():
    """option host_direct (default): rows in page-locked host memory the device can address take the bus without copy kernels;
    same bits as the copy path, at row offsets that are 16-byte aligned and at one that is not (falls back to copies)"""
    import torch
    from graphaudio_amd import OfflineAudioContext
    from tests import _graphs as G
    frames = 128 * 90
    outs = []
    for direct in (1, 0):
        ctx = OfflineAudioContext(48000)
        ctx.SetOption("host_direct", direct)
        G.config3_convolver(ctx, voices=3, taps=3000, frames=frames)
        pin = torch.zeros((2, frames + 8), dtype=torch.float32).pin_memory()
        out = pin.numpy()
        ctx.Render(out, 128 * 30, 0)
        ctx.Render(out, 128 * 20, 128 * 30)          # aligned offset
        ctx.Render(out, 128 * 40, 128 * 50 + 3)      # 12-byte offset: not a 16-byte aligned row
        outs.append(out.copy())
        ctx.Dispose()
    assert np.abs(outs[0]).max() > 1e-3
    assert np.array_equal(outs[0], outs[1])
    assert np.all(outs[0][:, 128 * 50:128 * 50 + 3] == 0)
