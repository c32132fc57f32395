"""Asynchronous renders into page-locked host rows (include/graphaudio_hip.h "pipelined renders", option host_defer).

An asynchronous render leaves its bus in device staging rows; the rows cross PCIe inside the next chunk's pre-mix launch, or inside
its first forward-transform launch, or -- a chunk without a convolver stage -- as copies in front of it; after the last chunk: from
ga_synchronize.  Whatever route a step's bus takes it
has to arrive, bit for bit what a blocking render of the same step produces -- also when consecutive steps write into the SAME
rows (the benchmark's loop) and when renders of several chunks are in flight.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch

from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G

SR = 48000


def _steps(builder, frames, steps, async_, pinned, same_rows=False, **opts):
    ctx = OfflineAudioContext(SR)
    for k, v in opts.items():
        ctx.SetOption(k, v)
    ch = builder(ctx)
    if async_:
        ctx.SetOption("async", 1)
    keep, outs = [], []
    for k in range(steps):
        if pinned:
            if not same_rows or not keep:
                keep.append(torch.zeros((ch, frames), dtype=torch.float32).pin_memory())
            a = keep[-1].numpy()
        else:
            a = np.zeros((ch, frames), np.float32)
        ctx.Render(a, frames)
        if same_rows:
            ctx.Synchronize()
            outs.append(a.copy())
        else:
            outs.append(a)
    ctx.Synchronize()
    st = ctx.GetStats()
    outs = [o.copy() for o in outs]
    ctx.Dispose()
    return outs, st


@pytest.mark.parametrize("shared", [True, False])   # rides in the next chunk's pre-mix launch / in its forward-transform launch
def test_deferred_hand_over_equals_blocking_renders(shared):
    frames, steps = 128 * 300, 5
    build = lambda c: G.config3_convolver(c, voices=12, taps=30000, frames=frames * steps, shared=shared)
    ref, _ = _steps(build, frames, steps, async_=False, pinned=False, coarse_min_blocks=1)
    got, st = _steps(build, frames, steps, async_=True, pinned=True, coarse_min_blocks=1)
    assert st["deferred_handovers"] == steps - 1   # every step's bus but the last rode along with the next chunk's first long launch
    for k in range(steps):
        assert G.rms(ref[k]) > 1e-4
        assert np.array_equal(ref[k], got[k]), k
    off, st2 = _steps(build, frames, steps, async_=True, pinned=True, coarse_min_blocks=1, host_defer=0)
    assert st2["deferred_handovers"] == 0
    for k in range(steps):
        assert np.array_equal(ref[k], off[k]), k


def test_sixteen_channel_bus():
    """config 5's shape: the 16-channel bus of step k (31 MB per 10 s at full size) crosses PCIe inside step k + 1's forward launch,
    whichever multiply-accumulate kernel the stage uses.  (Tried in round 3: carrying it in the matrix-core launch instead -- the
    forward stage drops 1.01 -> 0.65 ms, the multiply-accumulate stage grows 1.60 -> 2.14: its copy workgroups each hold a CU's LDS.)"""
    frames, steps = 128 * 320, 4
    build = lambda c: G.config5_ambisonic(c, sources=5, taps=32768, frames=frames * steps)
    for opts in ({}, {"coarse_mfma": 0}, {"coarse_wide": 0}):   # (different kernels round differently: a blocking reference per route)
        ref, _ = _steps(build, frames, steps, async_=False, pinned=False, coarse_min_blocks=1, **opts)
        got, st = _steps(build, frames, steps, async_=True, pinned=True, coarse_min_blocks=1, **opts)
        assert st["deferred_handovers"] == steps - 1
        for k in range(steps):
            assert G.rms(ref[k]) > 1e-5
            assert np.array_equal(ref[k], got[k]), (opts, k)
        kernels = " ".join(st["stage_kernel"]) if isinstance(st.get("stage_kernel"), (list, tuple)) else str(st.get("stage_kernel"))
        assert ("mfma16" in kernels) == (not opts), (opts, kernels)


def test_steps_that_reuse_the_same_rows():
    frames, steps = 128 * 260, 4
    build = lambda c: G.config3_convolver(c, voices=9, taps=20000, frames=frames * steps)
    ref, _ = _steps(build, frames, steps, async_=False, pinned=False, coarse_min_blocks=1)
    got, _ = _steps(build, frames, steps, async_=True, pinned=True, same_rows=True, coarse_min_blocks=1)
    for k in range(steps):
        assert np.array_equal(ref[k], got[k]), k


def test_render_spanning_several_chunks_and_a_biquad_graph():
    """chunks of 64 blocks: every chunk's hand-over rides with (or goes in front of) the next one; a graph without convolvers"""
    frames = 128 * 300
    for build in (lambda c: G.config3_convolver(c, voices=5, taps=20000, frames=frames), lambda c: G.config2_biquad(c, voices=16, frames=frames)):
        ref, _ = _steps(build, frames, 2, async_=False, pinned=False, coarse_min_blocks=1, max_chunk_blocks=64)
        got, _ = _steps(build, frames, 2, async_=True, pinned=True, coarse_min_blocks=1, max_chunk_blocks=64)
        for k in range(2):
            assert np.array_equal(ref[k], got[k]), k


def test_caller_supplied_stream_rows_are_complete_after_a_stream_wait():
    """on a stream the caller supplied (ga_context_set_stream) a host may wait on that stream -- or order its own work
    behind the render on it -- instead of calling ga_synchronize.  The deferred hand-over (the bus of call k crossing PCIe inside
    call k + 1) must therefore not apply there: every call enqueues its own copy."""
    frames, steps = 128 * 300, 3
    build = lambda c: G.config3_convolver(c, voices=12, taps=30000, frames=frames * steps)
    ref, _ = _steps(build, frames, steps, async_=False, pinned=False, coarse_min_blocks=1)
    ctx = OfflineAudioContext(SR)
    ctx.SetOption("coarse_min_blocks", 1)
    ch = build(ctx)
    stream = torch.cuda.Stream()
    ctx.SetStream(stream.cuda_stream)
    ctx.SetOption("async", 1)
    rows = [torch.zeros((ch, frames), dtype=torch.float32).pin_memory() for _ in range(steps)]
    for k in range(steps):
        ctx.Render(rows[k].numpy(), frames)
        stream.synchronize()   # the caller's own wait -- NOT ga_synchronize
        assert np.array_equal(ref[k], rows[k].numpy()), k
    assert ctx.GetStats()["deferred_handovers"] == 0
    # work ordered behind the render on the same stream sees the rows too (a device-side copy of the pinned rows)
    ctx2 = OfflineAudioContext(SR)
    ctx2.SetOption("coarse_min_blocks", 1)
    build(ctx2)
    ctx2.SetStream(stream.cuda_stream)
    ctx2.SetOption("async", 1)
    row = torch.zeros((ch, frames), dtype=torch.float32).pin_memory()
    ctx2.Render(row.numpy(), frames)
    with torch.cuda.stream(stream):
        dev = row.to("cuda", non_blocking=True)
    stream.synchronize()
    assert np.array_equal(ref[0], dev.cpu().numpy())
    ctx.Synchronize()
    ctx2.Synchronize()
    ctx.Dispose()
    ctx2.Dispose()


def test_the_resampler_table_doubles_without_stopping_the_pipeline():
    """The per-rate table of sample positions (resample_fast_kernel) holds 4096 blocks at first and doubles when a voice plays past it:
    the copy into the larger table is a plan entry, the old table is freed later (Context::retired) -- no wait, same samples."""
    from graphaudio_amd import AudioBufferSourceNode, GainNode, PlayableAudioBuffer
    from tests._oracle import OracleContext
    frames, steps = 128 * 700, 7   # 4,900 blocks
    total = frames * steps

    def build(ctx):
        ctx.Destination.SetChannelCount(2)
        for v in range(3):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(40 + v, int(total * 44100 / SR) + 4096), 44100)
            g = GainNode(ctx)
            g.Gain.Value = 0.3
            s.Connect(g).Connect(ctx.Destination)
            s.Start(0.0 if v else 0.01)
        return 2

    got, _ = _steps(build, frames, steps, async_=True, pinned=True)
    blocking, _ = _steps(build, frames, steps, async_=False, pinned=False)
    o = OracleContext(SR)
    build(o)
    ref = G.render(o, 2, total)
    for k in range(steps):
        assert np.array_equal(got[k], blocking[k]), k
        assert np.array_equal(got[k], ref[:, k * frames:(k + 1) * frames]), k
