"""Device memory follows the managed objects: dropped PlayableAudioBuffers, replaced impulse responses and disposed nodes
give their storage back (the reference relies on the .NET garbage collector for the same objects)."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import AudioBufferSourceNode, ConvolverNode, DelayNode, OfflineAudioContext, PlayableAudioBuffer
from tests import _graphs as G
from tests._oracle import OracleContext

SR = 48000


@pytest.mark.parametrize("coarse_min_blocks", [256, 1])   # 1: the 20,000-tap impulse responses take formulation D (history buffers, coarse spectra, kept taps)
def test_released_buffers_and_spectra_are_freed_but_playing_ones_stay(coarse_min_blocks):
    ctx = OfflineAudioContext(SR)
    ctx.SetOption("coarse_min_blocks", coarse_min_blocks)
    rng = np.random.default_rng(0)
    out = np.zeros((2, 128 * 4), np.float32)
    conv = ConvolverNode(ctx)
    conv.Connect(ctx.Destination)
    src = AudioBufferSourceNode(ctx)
    src.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(128 * 400) * 0.2).astype(np.float32), SR)
    src.Loop = True
    src.Connect(conv)
    src.Start()
    sizes = []
    for it in range(12):
        conv.Buffer = PlayableAudioBuffer.FromStereoArrays((rng.standard_normal(20000) * 0.05).astype(np.float32),
                                                          (rng.standard_normal(20000) * 0.05).astype(np.float32), SR)
        tmp = AudioBufferSourceNode(ctx)                     # a voice that comes and goes
        tmp.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(48000) * 0.2).astype(np.float32), SR)
        tmp.Connect(ctx.Destination)
        tmp.Start()
        ctx.Render(out, 128 * 4)
        tmp.Dispose()
        del tmp
        gc.collect()
        ctx.Render(out, 128 * 4)
        sizes.append(ctx.GetStats()["device_bytes_in_use"])
    assert sizes[-1] <= sizes[3] + (1 << 20), sizes      # steady state: no growth with the number of swaps
    assert G.rms(out) > 1e-4                                # the looping voice (its buffer is only referenced natively) still plays


def test_node_keeps_a_buffer_alive_after_the_host_dropped_it():
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(SR)
        ctx.Destination.SetChannelCount(1)
        rng = np.random.default_rng(1)
        s = AudioBufferSourceNode(ctx)
        b = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(128 * 30) * 0.2).astype(np.float32), SR)
        s.Buffer = b
        s.Connect(ctx.Destination)
        s.Start()
        out = np.zeros((1, 128 * 20), np.float32)
        ctx.Render(out, 128 * 5, 0)
        if mk is OfflineAudioContext:
            ctx._api.buffer_release(ctx._h, b._native_id(ctx))   # what PlayableAudioBuffer.__del__ does
        ctx.Render(out, 128 * 15, 128 * 5)
        outs.append(out)
    assert np.array_equal(outs[0], outs[1])


def test_disposed_delay_returns_its_lines():
    ctx = OfflineAudioContext(SR)
    out = np.zeros((2, 128 * 4), np.float32)
    ctx.Render(out, 128 * 4)
    base = ctx.GetStats()["device_bytes_in_use"]
    rng = np.random.default_rng(2)
    for it in range(6):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(4096) * 0.2).astype(np.float32), SR)
        d = DelayNode(ctx, 5.0)                               # 240,000-sample lines
        d.DelayTime.Value = 0.5
        s.Connect(d)
        d.Connect(ctx.Destination)
        s.Start()
        ctx.Render(out, 128 * 4)
        d.Dispose()
        s.Dispose()
        del s, d
        gc.collect()
        ctx.Render(out, 128 * 4)
    assert ctx.GetStats()["device_bytes_in_use"] <= base + (2 << 20)
