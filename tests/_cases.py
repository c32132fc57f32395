"""Small fixed graphs used for the golden fixtures (tests/golden/oracle_v1.npz) and the GPU-vs-golden tests."""
import numpy as np

from graphaudio_amd import (AudioBufferSourceNode, BiQuadFilterNode, ConvolverNode, FilterType, GainNode,
                            PlayableAudioBuffer)
from tests import _graphs as G

SR = 48000


def case_plumbing(ctx):
    return G.config1_plumbing(ctx, voices=8, frames=128 * 6)


def case_biquad(ctx):
    return G.config2_biquad(ctx, voices=12, frames=128 * 20)


def case_convolver(ctx):
    return G.config3_convolver(ctx, voices=5, taps=1500, frames=128 * 24)


def case_eq_resample(ctx):
    return G.config4_eq(ctx, voices=3, frames=128 * 24)


def case_true_stereo(ctx):
    irs = [G.synth_ir(c, 700) for c in range(4)]
    l, r = G.voice(50, 128 * 30), G.voice(51, 128 * 30)
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromStereoArrays(l, r, SR)
    cv = ConvolverNode(ctx)
    cv.Buffer = PlayableAudioBuffer.FromChannelArrays(irs, SR)
    s.Connect(cv).Connect(ctx.Destination)
    s.Start()
    return 2


def case_scheduling(ctx):
    """Staggered starts / stops / durations / loops: exercises segments, Ended -> Dispose, block-aligned start."""
    ctx.Destination.SetChannelCount(1)
    bus = GainNode(ctx)
    bus.Inputs[0].SetChannelCount(1)
    bus.Gain.Value = 0.5
    bus.Connect(ctx.Destination)
    for v in range(6):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(200 + v, 128 * (4 + 2 * v) + 17 * v), SR)
        g = GainNode(ctx)
        g.Inputs[0].SetChannelCount(1)
        g.Gain.Value = 0.25 + 0.1 * v
        s.Connect(g).Connect(bus)
        if v == 4:
            s.Loop = True
            s.LoopStart = 100 / SR
            s.LoopEnd = 700 / SR
        if v == 5:
            s.Start(0.003 * v, 64 / SR, 0.01)
        else:
            s.Start(0.004 * v)
        if v == 2:
            s.Stop(0.02)
        if v == 4:
            s.Stop(0.05)
    return 1


def case_source_replay(ctx):
    """Looping + resampling, a k-rate playbackRate ramp, a start offset beyond the loop end (general source replay)."""
    ctx.Destination.SetChannelCount(1)
    ctx.Destination.Inputs[0].SetChannelCount(1)
    a = AudioBufferSourceNode(ctx)
    a.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(300, 2000), 44100)
    a.Loop = True
    a.LoopStart = 300 / 44100
    a.LoopEnd = 1500 / 44100
    a.PlaybackRate.SetValueAtTime(0.8, 0.0)
    a.PlaybackRate.LinearRampToValueAtTime(1.6, 0.04)
    a.Connect(ctx.Destination)
    a.Start(0.0)
    b = AudioBufferSourceNode(ctx)
    b.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(301, 900), SR)
    b.Loop = True
    b.LoopStart = 100 / SR
    b.LoopEnd = 400 / SR
    b.Connect(ctx.Destination)
    b.Start(0.005, 600 / SR)      # offset beyond the loop end
    return 1


def case_core_nodes(ctx):
    """The remaining pure-Core nodes: oscillator -> panner, constant source on a gain, splitter / merger swap, delay."""
    from graphaudio_amd import (ChannelMergerNode, ChannelSplitterNode, ConstantSourceNode, DelayNode, OscillatorNode,
                                OscillatorType, StereoPannerNode)
    o = OscillatorNode(ctx)
    o.Type = OscillatorType.Sawtooth
    o.Frequency.Value = 311.0
    p = StereoPannerNode(ctx)
    p.Inputs[0].SetChannelCount(1)
    p.Pan.Value = -0.35
    g = GainNode(ctx)
    g.Gain.Value = 0.0
    cs = ConstantSourceNode(ctx)
    cs.Offset.SetValueAtTime(0.0, 0.0)
    cs.Offset.LinearRampToValueAtTime(0.3, 0.03)
    cs.Connect(g.Gain)                       # fade-in driven by the constant source
    o.Connect(p).Connect(g).Connect(ctx.Destination)
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromStereoArrays(G.voice(310, 128 * 30), G.voice(311, 128 * 30), SR)
    sp = ChannelSplitterNode(ctx, 2)
    mg = ChannelMergerNode(ctx, 2)
    d = DelayNode(ctx, 0.05)
    d.Inputs[0].SetChannelCount(1)
    d.DelayTime.Value = 0.0123
    s.Connect(sp)
    sp.Connect(mg, 0, 1)                     # L -> right channel, dry
    sp.Connect(d, 1, 0)
    d.Connect(mg, 0, 0)                      # R -> delay -> left channel
    mg.Connect(ctx.Destination)
    o.Start(0.002)
    o.Stop(0.07)
    cs.Start(0.0)
    s.Start(0.0)
    return 2


CASES = {
    "plumbing": (case_plumbing, 128 * 8),
    "biquad": (case_biquad, 128 * 16),
    "convolver": (case_convolver, 128 * 24),
    "eq_resample": (case_eq_resample, 128 * 20),
    "true_stereo": (case_true_stereo, 128 * 24),
    "scheduling": (case_scheduling, 128 * 30),
    "source_replay": (case_source_replay, 128 * 24),
    "core_nodes": (case_core_nodes, 128 * 32),
}
