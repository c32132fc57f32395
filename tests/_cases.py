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


CASES = {
    "plumbing": (case_plumbing, 128 * 8),
    "biquad": (case_biquad, 128 * 16),
    "convolver": (case_convolver, 128 * 24),
    "eq_resample": (case_eq_resample, 128 * 20),
    "true_stereo": (case_true_stereo, 128 * 24),
    "scheduling": (case_scheduling, 128 * 30),
}
