"""Random graph generator shared by the fuzz tests: builds the SAME random graph on any context."""
import numpy as np

from graphaudio_amd import (AudioBufferSourceNode, BiQuadFilterNode, ChannelCountMode, ConvolverNode, FilterType,
                            GainNode, PlayableAudioBuffer)

SR = 48000


def build_random_graph(ctx, seed, frames, keep=None):
    """Sources -> random chains (gain / biquad / convolver) -> optional shared bus nodes -> destination."""
    rng = np.random.default_rng(seed)
    dest_ch = int(rng.choice([1, 2, 2, 4]))
    ctx.Destination.SetChannelCount(dest_ch)
    if rng.random() < 0.3:
        ctx.Destination.Inputs[0].SetChannelCountMode(ChannelCountMode.Explicit)
    ir_len = int(rng.integers(50, 1500))
    shared_ir = PlayableAudioBuffer.FromChannelArrays(
        [(rng.standard_normal(ir_len) * 0.05).astype(np.float32) for _ in range(int(rng.choice([1, 2])))], SR)
    buses = []
    for _ in range(int(rng.integers(0, 3))):
        g = GainNode(ctx)
        g.Gain.Value = float(rng.uniform(0.3, 1.0))
        if rng.random() < 0.5:
            g.Inputs[0].SetChannelCount(int(rng.choice([1, 2])))
        if rng.random() < 0.3:
            g.Inputs[0].SetChannelCountMode(ChannelCountMode(int(rng.integers(0, 3))))
        g.Connect(ctx.Destination)
        buses.append(g)
    nvoices = int(rng.integers(2, 10))
    for v in range(nvoices):
        nch = int(rng.choice([1, 1, 2]))
        src_sr = int(rng.choice([SR, SR, 44100]))
        length = int(rng.integers(128 * 2, frames + 600))
        data = [(rng.standard_normal(length) * 0.25).astype(np.float32) for _ in range(nch)]
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromChannelArrays(data, src_sr)
        if src_sr == SR and rng.random() < 0.3:
            s.Loop = True
            if rng.random() < 0.5:
                s.LoopStart = float(rng.integers(0, length // 3)) / SR
                s.LoopEnd = float(rng.integers(length // 2, length)) / SR
        if src_sr == SR and rng.random() < 0.2:
            s.PlaybackRate.Value = float(rng.choice([0.5, 1.25]))
            s.Loop = False
        node = s
        for _ in range(int(rng.integers(0, 4))):
            kind = rng.choice(["gain", "gain_auto", "biquad", "biquad", "conv_shared", "conv_private"])
            if kind == "gain":
                n = GainNode(ctx)
                n.Gain.Value = float(rng.uniform(0.2, 1.2))
            elif kind == "gain_auto":
                n = GainNode(ctx)
                n.Gain.SetValueAtTime(float(rng.uniform(0, 1)), 0.0)
                n.Gain.LinearRampToValueAtTime(float(rng.uniform(0, 1)), float(rng.uniform(0.005, frames / SR)))
                if rng.random() < 0.5:
                    n.Gain.SetTargetAtTime(float(rng.uniform(0, 1)), float(rng.uniform(0.0, frames / SR)), float(rng.uniform(0.001, 0.05)))
            elif kind == "biquad":
                n = BiQuadFilterNode(ctx)
                n.Type = FilterType(int(rng.integers(0, 8)))
                n.Frequency.Value = float(rng.uniform(80, 12000))
                n.Q.Value = float(rng.uniform(0.3, 3.0))
                n.Gain.Value = float(rng.uniform(-9, 9))
            elif kind == "conv_shared":
                n = ConvolverNode(ctx)
                n.Buffer = shared_ir
            else:
                n = ConvolverNode(ctx)
                c = int(rng.choice([1, 2, 4]))
                taps = int(rng.integers(10, 700))
                n.Normalize = bool(rng.random() < 0.7)
                n.EnableTrueStereo = bool(rng.random() < 0.7)
                n.Buffer = PlayableAudioBuffer.FromChannelArrays(
                    [(rng.standard_normal(taps) * 0.1).astype(np.float32) for _ in range(c)], SR)
            if rng.random() < 0.25 and not isinstance(n, ConvolverNode):
                n.Inputs[0].SetChannelCount(int(rng.choice([1, 2, 3])))
                n.Inputs[0].SetChannelCountMode(ChannelCountMode(int(rng.integers(0, 3))))
            node.Connect(n)
            node = n
        target = buses[int(rng.integers(0, len(buses)))] if buses and rng.random() < 0.6 else ctx.Destination
        live = keep is None or v in keep  # (minimiser hook: unconnected voices are never pulled)
        if live:
            node.Connect(target)
        if rng.random() < 0.15 and buses and live:
            node.Connect(buses[0])  # fan-out
        when = float(rng.choice([0.0, 0.0, rng.uniform(0, frames / SR * 0.7)]))
        if rng.random() < 0.25:
            s.Start(when, float(rng.uniform(0, 0.01)), float(rng.uniform(0.005, frames / SR)))
        else:
            s.Start(when)
        if rng.random() < 0.2:
            s.Stop(float(rng.uniform(when, frames / SR)))
    return dest_ch
