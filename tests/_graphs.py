"""Graph builders shared by the oracle tests, the GPU parity tests, bench.py and smoke().

Every builder takes a context (HIP product or CPU oracle -- same host classes) and returns (ctx, out_channels).
Synthetic inputs follow BASELINE.md section 3 / SURVEY.md section 8(d):
  voices : default_rng(1000+v).standard_normal(n).astype(float32) * 0.25
  IRs    : default_rng(7+ch).standard_normal(taps) * exp(-6.9 * n / taps), float32, Normalize = true
"""
import numpy as np

from graphaudio_amd import (AudioBufferSourceNode, BiQuadFilterNode, ConvolverNode, FilterType, GainNode,
                            PlayableAudioBuffer)


def voice(v, n, scale=0.25):
    return (np.random.default_rng(1000 + v).standard_normal(n) * scale).astype(np.float32)


def synth_ir(ch, taps, seed0=7):
    n = np.arange(taps)
    return (np.random.default_rng(seed0 + ch).standard_normal(taps) * np.exp(-6.9 * n / taps)).astype(np.float32)


def config1_plumbing(ctx, voices=8, frames=128 * 6, sr=48000):
    """8 AudioBufferSourceNode -> GainNode(0.125) -> destination, mono (BASELINE.json configs[0])."""
    ctx.Destination.SetChannelCount(1)
    g = GainNode(ctx)
    g.Gain.Value = 0.125
    g.Inputs[0].SetChannelCount(1)
    g.Connect(ctx.Destination)
    for v in range(voices):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(voice(v, frames), sr)
        s.Connect(g)
        s.Start()
    return 1


def config2_biquad(ctx, voices=256, frames=48000, sr=48000, mono=True):
    """voices -> BiQuadFilterNode(lowpass, f = 200 * 2^(v/32) capped 20 kHz, Q = 0.707) -> Gain(1/16) -> mix."""
    if mono:
        ctx.Destination.SetChannelCount(1)
    for v in range(voices):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(voice(v, frames), sr)
        bq = BiQuadFilterNode(ctx)
        bq.Type = FilterType.Lowpass
        bq.Frequency.Value = min(20000.0, 200.0 * 2.0 ** (v / 32.0))
        bq.Q.Value = 0.707
        g = GainNode(ctx)
        g.Gain.Value = 1.0 / 16.0
        if mono:
            bq.Inputs[0].SetChannelCount(1)
            g.Inputs[0].SetChannelCount(1)
        s.Connect(bq).Connect(g).Connect(ctx.Destination)
        s.Start()
    return 1 if mono else 2


def config3_convolver(ctx, voices=1024, taps=65536, frames=48000, sr=48000, ir_channels=2, shared=True, voice_len=None):
    """voices -> per-voice ConvolverNode (shared stereo IR) -> destination (2 ch)  (BASELINE.json configs[2])."""
    vlen = voice_len or (frames + 256)
    irbuf = PlayableAudioBuffer.FromChannelArrays([synth_ir(c, taps) for c in range(ir_channels)], sr)
    ctx.Destination.SetChannelCount(ir_channels)
    for v in range(voices):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(voice(v, vlen), sr)
        cv = ConvolverNode(ctx)
        if shared:
            cv.Buffer = irbuf
        else:
            cv.Buffer = PlayableAudioBuffer.FromChannelArrays(
                [synth_ir(c, taps, seed0=7 + 100 * (v + 1)) for c in range(ir_channels)], sr)
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()
    return ir_channels


def config4_eq(ctx, voices=4096, frames=48000, sr=48000, src_sr=44100):
    """voices: 44.1k mono buffer -> CubicResampler -> 5-band biquad EQ -> gain automation -> mix."""
    n_in = int(frames * src_sr / sr) + 2048
    bands = [(FilterType.Lowshelf, 100.0, 1.0, 6.0), (FilterType.Peaking, 400.0, 1.0, -6.0),
             (FilterType.Peaking, 1000.0, 1.0, 6.0), (FilterType.Peaking, 4000.0, 1.0, -6.0),
             (FilterType.Highshelf, 10000.0, 1.0, 6.0)]
    for v in range(voices):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(voice(v, n_in), src_sr)
        node = s
        for (ft, f, q, gdb) in bands:
            bq = BiQuadFilterNode(ctx)
            bq.Type = ft
            bq.Frequency.Value = f
            bq.Q.Value = q
            bq.Gain.Value = gdb
            node = node.Connect(bq)
        g = GainNode(ctx)
        g.Gain.SetValueAtTime(0.0, 0.0)
        g.Gain.LinearRampToValueAtTime(1.0 / 64.0, 0.5)
        g.Gain.SetTargetAtTime(0.0, 8.0, 0.3)
        node.Connect(g).Connect(ctx.Destination)
        s.Start()
    return 2


def config5_ambisonic(ctx, sources=64, taps=32768, frames=48000, ir_channels=16, v0=0, sr=48000):
    """sources -> per-source ConvolverNode with its own 16-channel impulse response -> 16-channel destination
    (BASELINE.json configs[4]; one GPU's shard is 64 of the 512 sources, `v0` = first source of the shard)."""
    ctx.Destination.SetChannelCount(ir_channels)
    for v in range(v0, v0 + sources):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(voice(v, frames + 256), sr)
        cv = ConvolverNode(ctx)
        cv.Buffer = PlayableAudioBuffer.FromChannelArrays(
            [synth_ir(c, taps, seed0=7 + 100 * v) for c in range(ir_channels)], sr)
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()
    return ir_channels


def kit_scene(ctx, voices=64, frames=48000, sr=48000, taps=32768):
    """the graph shapes GraphAudio.Kit builds around the hot path.
    Voices (Sound: source -> StereoPanner, GraphAudio.Kit Sound's panner) -> two child buses (AudioBus = one GainNode,
    AudioBus.cs:76-91; one of them fading, AudioBus.Fade) -> master bus -> ReverbEffect (ReverbEffect.cs:63-81: inputSplit ->
    dry -> outputMerge ; inputSplit -> downmixer (explicit mono) -> convolver -> wet -> outputMerge) -> destination.
    The convolver sits AFTER the mix: one post-mix, non-sharded instance instead of one per voice."""
    from graphaudio_amd import ChannelCountMode, StereoPannerNode
    master = GainNode(ctx)
    master.Gain.Value = 0.8
    sfx, music = GainNode(ctx), GainNode(ctx)
    sfx.Gain.Value = 0.9
    music.Gain.SetValueAtTime(1.0, 0.0)                 # AudioBus.Fade: a linear ramp on the bus gain
    music.Gain.LinearRampToValueAtTime(0.3, frames / sr * 0.6)
    sfx.Connect(master)
    music.Connect(master)
    # ReverbEffect
    split, merge, dry, wet, down = (GainNode(ctx) for _ in range(5))
    dry.Gain.Value = 0.7
    wet.Gain.Value = 0.4
    down.Inputs[0].SetChannelCount(1)
    down.Inputs[0].SetChannelCountMode(ChannelCountMode.Explicit)
    conv = ConvolverNode(ctx)
    conv.Buffer = PlayableAudioBuffer.FromChannelArrays([synth_ir(c, taps) for c in range(2)], sr)
    master.Connect(split)
    split.Connect(dry)
    dry.Connect(merge)
    split.Connect(down)
    down.Connect(conv)
    conv.Connect(wet)
    wet.Connect(merge)
    merge.Connect(ctx.Destination)
    for v in range(voices):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(voice(v, frames, 0.25 / 8), sr)
        p = StereoPannerNode(ctx)
        p.Inputs[0].SetChannelCount(1)
        p.Pan.Value = -1.0 + 2.0 * v / max(voices - 1, 1)
        s.Connect(p)
        p.Connect(sfx if v % 2 == 0 else music)
        s.Start(0.0 if v % 4 else 0.01)
    return 2


def kit_bus_hierarchy(ctx, voices=24, frames=48000, sr=48000, taps=(9000, 20000, 3000)):
    """SURVEY.md 8(f) rank 4, the other Kit shape: a HIERARCHY of buses, each with its own EffectChain.
    `AudioBus` = one GainNode whose EffectChain leads to the parent's input (AudioBus.cs:76-91); `EffectChain.Rebuild`
    (EffectChain.cs:127-149) wires source -> effect[0].Input ... effect[n-1].Output -> destination; the only Kit effect is
    `ReverbEffect` (ReverbEffect.cs:63-81: inputSplit -> dry -> outputMerge ; inputSplit -> downmixer (explicit mono) -> convolver ->
    wet -> outputMerge).  Here: master (one reverb) <- sfx (two reverbs in series) <- {sfx/weapons (none), sfx/steps (one reverb,
    fading)} ; master <- music (none).  Voices: source -> StereoPanner -> a leaf bus."""
    from graphaudio_amd import ChannelCountMode, StereoPannerNode

    def reverb(ir_seed, ntaps, wet_level):
        split, merge, dry, wet, down = (GainNode(ctx) for _ in range(5))
        dry.Gain.Value = 1.0 - 0.5 * wet_level
        wet.Gain.Value = wet_level
        down.Inputs[0].SetChannelCount(1)
        down.Inputs[0].SetChannelCountMode(ChannelCountMode.Explicit)
        conv = ConvolverNode(ctx)
        conv.Buffer = PlayableAudioBuffer.FromChannelArrays([synth_ir(c, ntaps, seed0=ir_seed) for c in range(2)], sr)
        split.Connect(dry)
        dry.Connect(merge)
        split.Connect(down)
        down.Connect(conv)
        conv.Connect(wet)
        wet.Connect(merge)
        return split, merge

    def bus(parent_input, effects, gain=1.0):
        g = GainNode(ctx)
        g.Gain.Value = gain
        node = g
        for (inp, out) in effects:
            node.Connect(inp)
            node = out
        node.Connect(parent_input)
        return g

    master = bus(ctx.Destination, [reverb(7, taps[0], 0.3)], 0.8)
    sfx = bus(master, [reverb(40, taps[1], 0.25), reverb(60, taps[2], 0.4)], 0.9)
    music = bus(master, [])
    weapons = bus(sfx, [])
    steps = bus(sfx, [reverb(80, taps[2], 0.5)], 1.0)
    steps.Gain.SetValueAtTime(1.0, 0.0)                      # AudioBus.Fade
    steps.Gain.LinearRampToValueAtTime(0.2, frames / sr * 0.7)
    leaves = [weapons, steps, music]
    for v in range(voices):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(voice(v, frames, 0.25 / 4), sr)
        p = StereoPannerNode(ctx)
        p.Inputs[0].SetChannelCount(1)
        p.Pan.Value = -1.0 + 2.0 * v / max(voices - 1, 1)
        s.Connect(p)
        p.Connect(leaves[v % 3])
        s.Start(0.0 if v % 5 else 0.02)
    return 2


def render(ctx, channels, frames):
    out = np.zeros((channels, frames), dtype=np.float32)
    ctx.Render(out, frames)
    return out


def rms(a):
    return float(np.sqrt(np.mean(np.square(a.astype(np.float64)))))
