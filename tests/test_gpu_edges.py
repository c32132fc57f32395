"""Edge cases of the boundary on the device path (empty / ragged inputs, maximum sizes), against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import (ArgumentException, ArgumentOutOfRangeException, AudioBufferSourceNode, ConvolverNode, GainNode,
                            ObjectDisposedException, OfflineAudioContext, PlayableAudioBuffer)
from tests import _graphs as G
from tests._oracle import OracleContext

SR = 48000


def both(fn):
    return [fn(OracleContext(SR)), fn(OfflineAudioContext(SR))]


def test_empty_graph_and_single_frame_renders():
    def run(ctx):
        out = np.full((2, 300), 5.0, np.float32)
        ctx.Render(out, 1, 0)          # one frame: a whole block is rendered, 127 frames cached
        ctx.Render(out, 299, 1)
        return out, ctx.CurrentBlock
    (ro, rb), (go, gb) = both(run)
    assert np.array_equal(ro, go) and rb == gb == 3 and np.abs(go).max() == 0.0


def test_32_channel_destination_and_buffers():
    def run(ctx):
        rng = np.random.default_rng(0)
        ctx.Destination.SetChannelCount(32)
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromChannelArrays([(rng.standard_normal(700) * 0.1).astype(np.float32) for _ in range(32)], SR)
        g = GainNode(ctx)
        g.Inputs[0].SetChannelCount(32)
        s.Connect(g)
        g.Connect(ctx.Destination)
        s.Start()
        out = np.zeros((32, 128 * 8), np.float32)
        ctx.Render(out, 128 * 8)
        return out
    ro, go = both(run)
    assert G.rms(ro) > 1e-3 and np.array_equal(ro, go)
    with pytest.raises(ArgumentOutOfRangeException):
        OfflineAudioContext(SR).Destination.SetChannelCount(33)


def test_ragged_buffers_one_sample_and_one_tap():
    def run(ctx):
        ctx.Destination.SetChannelCount(1)
        ctx.Destination.Inputs[0].SetChannelCount(1)
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(np.array([0.5], np.float32), SR)      # a single sample: its only block is dropped
        s.Connect(ctx.Destination)
        s.Start()
        s2 = AudioBufferSourceNode(ctx)
        s2.Buffer = PlayableAudioBuffer.FromMonoArray(np.linspace(-1, 1, 129).astype(np.float32), SR)   # one block + 1 sample
        c = ConvolverNode(ctx)
        c.Normalize = False
        c.Buffer = PlayableAudioBuffer.FromMonoArray(np.array([0.25], np.float32), SR)    # a 1-tap "impulse response"
        s2.Connect(c)
        c.Connect(ctx.Destination)
        s2.Start()
        out = np.zeros((1, 128 * 4), np.float32)
        ctx.Render(out, 128 * 4)
        return out
    ro, go = both(run)
    assert np.count_nonzero(ro) > 100
    assert np.abs(ro - go).max() <= 1e-7
    with pytest.raises(ArgumentException):
        PlayableAudioBuffer.FromChannelArrays([], SR)


def test_errors_after_dispose_and_bad_arguments():
    ctx = OfflineAudioContext(SR)
    out = np.zeros((2, 256), np.float32)
    with pytest.raises(ArgumentOutOfRangeException):
        ctx.Render(out, 0)
    with pytest.raises(ArgumentOutOfRangeException):
        ctx.Render(out, 10, -1)
    ctx.Render(out, 256)
    ctx.Dispose()
    with pytest.raises(ObjectDisposedException):
        ctx.Render(out, 256)


def test_many_control_segments_in_one_render():
    """150 voices that start and end at different blocks: far more than 96 control segments, so the engine closes chunks early
    (kMaxSegs) and continues -- the result must not depend on where the chunks end."""
    def run(ctx):
        ctx.Destination.SetChannelCount(1)
        ctx.Destination.Inputs[0].SetChannelCount(1)
        for v in range(150):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(400 + v, 128 * (3 + v % 17) + 11 * v), SR)
            g = GainNode(ctx)
            g.Inputs[0].SetChannelCount(1)
            g.Gain.Value = 0.05
            s.Connect(g)
            g.Connect(ctx.Destination)
            s.Start((2 * v + 0.5) * 128 / SR)
        out = np.zeros((1, 128 * 340), np.float32)
        ctx.Render(out, 128 * 340)
        segs = ctx.GetStats()["segments"] if hasattr(ctx, "GetStats") else 0
        return out, segs
    (ro, _), (go, segs) = both(run)
    assert G.rms(ro) > 1e-3
    assert np.array_equal(ro, go)
    assert segs > 300


@pytest.mark.parametrize("taps,shared", [(128 * 9, False), (128 * 70, False), (128 * 9, True)])
def test_convolver_unplugged_for_a_while_keeps_its_state(taps, shared):   # per-node MFMA path / block-axis FFT / shared-IR GEMM rows
    """A convolver that is not reachable from the destination is not processed (pull model): its frequency-domain delay
    line freezes and continues when it is connected again.  On the device the delay line of a convolver lives in the
    spectra planes of the chunk that last ran it -- it has to survive the chunks in which other convolvers reuse them."""
    def run(ctx):
        if isinstance(ctx, OfflineAudioContext):
            ctx.SetOption("max_chunk_blocks", 2)
        ctx.Destination.SetChannelCount(2)
        convs = []
        common = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(ch, taps, seed0=40) for ch in range(2)], SR)
        for v in range(9 if shared else 3):   # 8 or more users of one impulse response become rows of one group
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(70 + v, 128 * 40), SR)
            c = ConvolverNode(ctx)
            c.Buffer = common if shared else PlayableAudioBuffer.FromChannelArrays(
                [G.synth_ir(ch, taps, seed0=50 + 10 * v) for ch in range(2)], SR)
            s.Connect(c)
            c.Connect(ctx.Destination)
            s.Start()
            convs.append(c)
        out = np.zeros((2, 128 * 40), np.float32)
        pos = 0
        def piece(nblk):
            nonlocal pos
            ctx.Render(out, 128 * nblk, pos)
            pos += 128 * nblk
        piece(5)
        convs[1].Disconnect()          # node 1 drops out: nodes 0 and 2 keep running (their rows move up)
        piece(3)
        convs[0].Disconnect()
        piece(4)
        convs[1].Connect(ctx.Destination)
        piece(7)
        convs[0].Connect(ctx.Destination)
        piece(21)
        return out
    ro, go = both(run)
    assert G.rms(ro) > 1e-3
    err = G.rms(ro - go)
    assert err <= 2e-6 * max(G.rms(ro), 1e-3), err
