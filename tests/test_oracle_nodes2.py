"""Pins the oracle's restatement of the remaining pure-Core nodes against closed forms:
ChannelSplitter/Merger, ConstantSource, StereoPanner, Oscillator, Delay."""
import math

import numpy as np
import pytest

from graphaudio_amd import (AudioBufferSourceNode, ChannelMergerNode, ChannelSplitterNode, ConstantSourceNode, DelayNode,
                            GainNode, InvalidOperationException, ArgumentOutOfRangeException, OscillatorNode, OscillatorType,
                            PlayableAudioBuffer, StereoPannerNode)
from tests._oracle import OracleContext

SR = 48000


def ctx_n(ch):
    ctx = OracleContext(SR)
    ctx.Destination.SetChannelCount(ch)
    return ctx


def render(ctx, ch, frames):
    out = np.zeros((ch, frames), np.float32)
    ctx.Render(out, frames)
    return out


def stereo_source(ctx, frames, seed=0):
    rng = np.random.default_rng(seed)
    L = (rng.standard_normal(frames) * 0.25).astype(np.float32)
    R = (rng.standard_normal(frames) * 0.25).astype(np.float32)
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromStereoArrays(L, R, SR)
    return s, L, R


def test_splitter_and_merger_route_channels():
    ctx = ctx_n(2)
    s, L, R = stereo_source(ctx, 128 * 6)
    sp = ChannelSplitterNode(ctx, 3)         # third output: input has no channel 2 -> cleared (ChannelSplitterNode.cs:49-52)
    mg = ChannelMergerNode(ctx, 2)
    s.Connect(sp)
    sp.Connect(mg, 0, 1)                     # swap: L -> channel 1, R -> channel 0
    sp.Connect(mg, 1, 0)
    mg.Connect(ctx.Destination)
    s.Start()
    out = render(ctx, 2, 128 * 5)
    assert np.array_equal(out[0], R[:128 * 5]) and np.array_equal(out[1], L[:128 * 5])
    with pytest.raises(ArgumentOutOfRangeException):
        ChannelSplitterNode(ctx, 33)
    with pytest.raises(ArgumentOutOfRangeException):
        ChannelMergerNode(ctx, 0)


def test_merger_takes_channel_zero_of_each_input_after_the_input_mix():
    ctx = ctx_n(2)
    s, L, R = stereo_source(ctx, 128 * 4)
    mg = ChannelMergerNode(ctx, 2)
    s.Connect(mg, 0, 0)      # default input: 2 channels, Max -> the stereo source arrives as stereo, channel 0 = L is taken
    mg.Connect(ctx.Destination)
    s.Start()
    out = render(ctx, 2, 128 * 3)
    assert np.array_equal(out[0], L[:128 * 3])
    assert np.abs(out[1]).max() == 0.0


def test_constant_source_start_stop_are_sample_accurate_and_offset_is_a_rate():
    ctx = ctx_n(1)
    cs = ConstantSourceNode(ctx)
    cs.Offset.SetValueAtTime(0.0, 0.0)
    cs.Offset.LinearRampToValueAtTime(1.0, 1000 / SR)
    cs.Connect(ctx.Destination)
    ctx.Destination.Inputs[0].SetChannelCount(1)
    start, stop = 200.5 / SR, 700.25 / SR
    cs.Start(start)
    cs.Stop(stop)
    out = render(ctx, 1, 128 * 8)[0]
    sf = math.ceil(200.5)          # Math.Ceiling((startTime - t0) * sr) in the block that contains the start (:95-100)
    ef = math.floor(700.25)        # Math.Floor((stopTime - t0) * sr)  (:102-108)
    want = np.zeros(128 * 8)
    t = np.arange(128 * 8)
    want[sf:ef] = np.minimum(t[sf:ef] / 1000.0, 1.0)
    assert np.abs(out - want).max() < 2e-6
    assert out[sf - 1] == 0.0 and out[sf] != 0.0 and out[ef - 1] != 0.0 and out[ef] == 0.0
    assert ctx._api.node_has_ended(ctx._h, cs._id) == 1


def test_constant_source_second_start_is_ignored_and_oscillator_second_start_throws():
    ctx = ctx_n(1)
    cs = ConstantSourceNode(ctx)
    cs.Start(0.0)
    cs.Start(0.5)        # ConstantSourceNode.cs:48-49: silently ignored
    osc = OscillatorNode(ctx)
    osc.Connect(ctx.Destination)
    osc.Start(0.0)
    osc.Start(0.1)       # OscillatorNode.cs:58-59 throws inside the queued command; DrainCommands swallows it (AudioContextBase.cs:291-305)
    out = render(ctx, 1, 256)
    assert out[0, 1] != 0.0   # the first Start stands


@pytest.mark.parametrize("typ", list(OscillatorType))
def test_oscillator_waveforms(typ):
    ctx = ctx_n(1)
    ctx.Destination.Inputs[0].SetChannelCount(1)
    osc = OscillatorNode(ctx)
    osc.Type = typ
    f = 440.0
    osc.Frequency.Value = f
    osc.Connect(ctx.Destination)
    osc.Start(0.0)
    n = 128 * 20
    out = render(ctx, 1, n)[0].astype(np.float64)
    # the reference accumulates the phase in double with a conditional 2 pi wrap (:137-143)
    ph = 0.0
    want = np.zeros(n)
    inc = (2.0 * math.pi * float(np.float32(f))) / SR
    for i in range(n):
        if typ == OscillatorType.Sine:
            want[i] = math.sin(ph)
        elif typ == OscillatorType.Square:
            want[i] = 1.0 if ph < math.pi else -1.0
        elif typ == OscillatorType.Sawtooth:
            want[i] = 2.0 * (ph / (2.0 * math.pi)) - 1.0
        else:
            t = ph / (2.0 * math.pi)
            want[i] = 4.0 * abs(t - math.floor(t + 0.5)) - 1.0
        ph += inc
        if ph >= 2.0 * math.pi:
            ph -= 2.0 * math.pi
    assert np.abs(out - want).max() < 1e-6
    if typ == OscillatorType.Sine:   # and it IS a 440 Hz sine
        assert np.abs(out - np.sin(2 * np.pi * f * np.arange(n) / SR)).max() < 1e-5


def test_stereo_panner_equal_power_mono_and_stereo_laws():
    for pan in (-1.0, -0.3, 0.0, 0.6, 1.0):
        rng = np.random.default_rng(1)
        x = (rng.standard_normal(128 * 4) * 0.25).astype(np.float32)
        xx = (np.float32(pan) + np.float32(1)) * np.float32(0.5)
        gl, gr = math.cos(xx * math.pi / 2), math.sin(xx * math.pi / 2)
        xs = pan + 1.0 if pan <= 0 else pan
        sl, sr_ = math.cos(xs * math.pi / 2), math.sin(xs * math.pi / 2)
        assert abs(gl * gl + gr * gr - 1.0) < 1e-6
        # (a) mono input (input limited to 1 channel): x = (pan + 1) / 2 ; L = cos(x pi/2), R = sin(x pi/2)  (:88-101)
        ctx = ctx_n(2)
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(x, SR)
        p = StereoPannerNode(ctx)
        p.Inputs[0].SetChannelCount(1)
        p.Pan.Value = pan
        s.Connect(p)
        p.Connect(ctx.Destination)
        s.Start()
        out = render(ctx, 2, 128 * 3)
        assert np.abs(out[0] - x[:384] * gl).max() < 1e-6 and np.abs(out[1] - x[:384] * gr).max() < 1e-6
        # (b) default input (2 channels, ClampedMax): block 0 finds no upstream buffer of the PREVIOUS block (AudioNodeInput.cs:109,
        # 140-168), mixes the mono source into 2 channels and applies the STEREO law to L = R = x; from block 1 on the input is
        # mono, but the gains are only recomputed when pan CHANGES (:92), so the mono path keeps the stereo-law gains
        ctx = ctx_n(2)
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(x, SR)
        p = StereoPannerNode(ctx)
        p.Pan.Value = pan
        s.Connect(p)
        p.Connect(ctx.Destination)
        s.Start()
        out = render(ctx, 2, 128 * 3)
        x0 = x[:128].astype(np.float64)
        w0 = (x0 + x0 * sl, x0 * sr_) if pan <= 0 else (x0 * sl, x0 + x0 * sr_)
        assert np.abs(out[0, :128] - w0[0]).max() < 1e-6 and np.abs(out[1, :128] - w0[1]).max() < 1e-6
        assert np.abs(out[0, 128:] - x[128:384] * sl).max() < 1e-6 and np.abs(out[1, 128:] - x[128:384] * sr_).max() < 1e-6
        # (c) stereo input (:123-147)
        ctx = ctx_n(2)
        s, L, R = stereo_source(ctx, 128 * 4, seed=2)
        p = StereoPannerNode(ctx)
        p.Pan.Value = pan
        s.Connect(p)
        p.Connect(ctx.Destination)
        s.Start()
        out = render(ctx, 2, 128 * 3)
        L, R = L[:384].astype(np.float64), R[:384].astype(np.float64)
        if pan <= 0:
            wl, wr = L + R * sl, R * sr_
        else:
            wl, wr = L * sl, R + L * sr_
        assert np.abs(out[0] - wl).max() < 1e-6 and np.abs(out[1] - wr).max() < 1e-6


def test_stereo_panner_silent_input_gives_silent_stereo_and_clamped_max_keeps_mono_mono():
    ctx = ctx_n(2)
    p = StereoPannerNode(ctx)
    g = GainNode(ctx)
    g.Connect(p)
    p.Connect(ctx.Destination)
    out = render(ctx, 2, 256)
    assert np.abs(out).max() == 0.0


def test_delay_is_a_pure_sample_delay_with_truncated_delay_samples():
    ctx = ctx_n(1)
    ctx.Destination.Inputs[0].SetChannelCount(1)
    rng = np.random.default_rng(4)
    x = (rng.standard_normal(128 * 12) * 0.25).astype(np.float32)
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromMonoArray(x, SR)
    d = DelayNode(ctx, 0.05)
    d.Inputs[0].SetChannelCount(1)
    dt = 300.7 / SR
    d.DelayTime.Value = dt
    s.Connect(d)
    d.Connect(ctx.Destination)
    s.Start()
    n = 128 * 10
    out = render(ctx, 1, n)[0]
    ds = int(np.float32(np.float32(dt) * np.float32(SR)))   # (int)(delayTimes[i] * Context.SampleRate): float math, truncation (:68)
    want = np.zeros(n, np.float32)
    want[ds:] = x[: n - ds]
    assert np.array_equal(out, want)
    with pytest.raises(ArgumentOutOfRangeException):
        DelayNode(ctx, 11.0)


def test_delay_zero_delay_reads_zero_and_tail_outlives_the_source():
    ctx = ctx_n(1)
    ctx.Destination.Inputs[0].SetChannelCount(1)
    x = np.ones(128 * 2, np.float32)
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromMonoArray(x, SR)
    d = DelayNode(ctx, 0.01)
    d.Inputs[0].SetChannelCount(1)
    s.Connect(d)
    d.Connect(ctx.Destination)
    s.Start()
    out = render(ctx, 1, 128 * 3)[0]
    assert np.abs(out).max() == 0.0            # delaySamples <= 0 -> Read returns 0 (:138-140)
    ctx = ctx_n(1)
    ctx.Destination.Inputs[0].SetChannelCount(1)
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromMonoArray(np.ones(128 * 3, np.float32), SR)   # plays 2 blocks (last one dropped)
    d = DelayNode(ctx, 0.01)
    d.Inputs[0].SetChannelCount(1)
    d.DelayTime.Value = 200 / SR
    s.Connect(d)
    d.Connect(ctx.Destination)
    s.Start()
    out = render(ctx, 1, 128 * 5)[0]
    want = np.zeros(128 * 5, np.float32)
    want[200:200 + 256] = 1.0
    assert np.array_equal(out, want)
