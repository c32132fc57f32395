"""GPU parity for the remaining pure-Core nodes: HIP path vs the CPU oracle, through the C ABI."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import (AudioBufferSourceNode, BiQuadFilterNode, ChannelMergerNode, ChannelSplitterNode, ConstantSourceNode,
                            GainNode, NotSupportedException, OfflineAudioContext, OscillatorNode, OscillatorType,
                            PlayableAudioBuffer, StereoPannerNode)
from tests import _graphs as G
from tests._oracle import OracleContext

SR = 48000


def pair(build, ch, frames, pieces=None, chunk=0):
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(SR)
        if chunk and mk is OfflineAudioContext:
            ctx.SetOption("max_chunk_blocks", chunk)
        ctx.Destination.SetChannelCount(ch)
        hold = build(ctx)
        out = np.zeros((ch, frames), np.float32)
        pos = 0
        for p in (pieces or [frames]):
            k = min(p, frames - pos)
            if k <= 0:
                break
            ctx.Render(out, k, pos)
            pos += k
        if pos < frames:
            ctx.Render(out, frames - pos, pos)
        outs.append(out)
        del hold
        ctx.Dispose()
    return outs


def _stereo(ctx, n, seed):
    rng = np.random.default_rng(seed)
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromStereoArrays((rng.standard_normal(n) * 0.25).astype(np.float32),
                                                   (rng.standard_normal(n) * 0.25).astype(np.float32), SR)
    return s


def test_splitter_merger_swap_and_processing_between():
    def build(ctx):
        s = _stereo(ctx, 128 * 30, 1)
        sp = ChannelSplitterNode(ctx, 3)
        mg = ChannelMergerNode(ctx, 3)
        g = GainNode(ctx)
        g.Gain.Value = 0.5
        g.Inputs[0].SetChannelCount(1)
        bq = BiQuadFilterNode(ctx)
        bq.Inputs[0].SetChannelCount(1)
        s.Connect(sp)
        sp.Connect(g, 0, 0)
        g.Connect(mg, 0, 1)      # L * 0.5 -> channel 1
        sp.Connect(bq, 1, 0)
        bq.Connect(mg, 0, 0)     # lowpass(R) -> channel 0
        sp.Connect(mg, 2, 2)     # the source has no third channel: silent
        mg.Connect(ctx.Destination)
        s.Start(0.01)
        return (s, sp, mg, g, bq)
    ref, got = pair(build, 3, 128 * 36, pieces=[700, 1300], chunk=9)
    assert G.rms(ref) > 1e-3 and np.abs(ref[2]).max() == 0.0
    assert np.array_equal(ref, got)


def test_constant_source_window_and_automation():
    def build(ctx):
        ctx.Destination.Inputs[0].SetChannelCount(1)
        cs = ConstantSourceNode(ctx)
        cs.Offset.SetValueAtTime(0.1, 0.0)
        cs.Offset.LinearRampToValueAtTime(0.9, 0.05)
        cs.Offset.SetTargetAtTime(0.2, 0.06, 0.01)
        cs.Connect(ctx.Destination)
        cs.Start(1234.5 / SR)
        cs.Stop(4000.25 / SR)
        c2 = ConstantSourceNode(ctx)       # constant value, stopped through the Start duration
        c2.Offset.Value = -0.25
        c2.Connect(ctx.Destination)
        c2.Start(0.0, 0.0, 2000.7 / SR)
        return (cs, c2)
    ref, got = pair(build, 1, 128 * 40, pieces=[1000, 300, 2900], chunk=7)
    assert np.count_nonzero(ref) > 3000
    assert np.abs(ref - got).max() <= 1e-6          # ramps: device pow/exp vs glibc


@pytest.mark.parametrize("typ", list(OscillatorType))
def test_oscillator_types_bit_exact(typ):
    def build(ctx):
        ctx.Destination.Inputs[0].SetChannelCount(1)
        o = OscillatorNode(ctx)
        o.Type = typ
        o.Frequency.Value = 997.0
        o.Connect(ctx.Destination)
        o.Start(300.2 / SR)
        o.Stop(128 * 70 + 17.5 / SR)
        o2 = OscillatorNode(ctx)
        o2.Type = typ
        o2.Frequency.Value = 31.0
        g = GainNode(ctx)
        g.Gain.Value = 0.25
        g.Inputs[0].SetChannelCount(1)
        o2.Connect(g)
        g.Connect(ctx.Destination)
        o2.Start(0.0)
        o2.Stop(5000.9 / SR)
        return (o, o2, g)
    ref, got = pair(build, 1, 128 * 80, pieces=[3000, 5000], chunk=11)
    assert G.rms(ref) > 0.1
    # sin(double) of the device math library vs glibc differ in the last bit of the DOUBLE result: after the cast to float
    # a sample can differ by one float ulp in rare cases
    assert np.abs(ref - got).max() <= 1.2e-7
    assert np.mean(ref != got) < 1e-3


def test_oscillator_frequency_automation():
    def build(ctx):
        ctx.Destination.Inputs[0].SetChannelCount(1)
        o = OscillatorNode(ctx)
        o.Frequency.SetValueAtTime(200.0, 0.0)
        o.Frequency.ExponentialRampToValueAtTime(4000.0, 0.1)
        o.Connect(ctx.Destination)
        o.Start(0.0)
        return (o,)
    ref, got = pair(build, 1, 128 * 50, chunk=13)
    # the frequency curve comes from device pow(): a 1-ulp float difference in f shifts the phase by ~1e-9 per sample
    assert G.rms(ref - got) <= 1e-5


@pytest.mark.parametrize("pan", [-1.0, -0.4, 0.0, 0.3, 1.0])
def test_stereo_panner_static_bit_exact(pan):
    def build(ctx):
        rng = np.random.default_rng(3)
        m = AudioBufferSourceNode(ctx)
        m.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(128 * 30) * 0.25).astype(np.float32), SR)
        p1 = StereoPannerNode(ctx)      # default input: block 0 stereo law on the up-mixed mono, then mono path with stale gains
        p1.Pan.Value = pan
        m.Connect(p1)
        p1.Connect(ctx.Destination)
        s = _stereo(ctx, 128 * 30, 5)
        p2 = StereoPannerNode(ctx)
        p2.Pan.Value = -pan
        s.Connect(p2)
        p2.Connect(ctx.Destination)
        m2 = AudioBufferSourceNode(ctx)
        m2.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(128 * 30) * 0.25).astype(np.float32), SR)
        p3 = StereoPannerNode(ctx)      # mono law from the first block
        p3.Inputs[0].SetChannelCount(1)
        p3.Pan.Value = pan
        m2.Connect(p3)
        p3.Connect(ctx.Destination)
        m.Start(0.0)
        s.Start(0.004)
        m2.Start(0.0)
        return (m, p1, s, p2, m2, p3)
    ref, got = pair(build, 2, 128 * 32, pieces=[1000], chunk=9)
    assert G.rms(ref) > 1e-2
    # the three panner outputs are summed on the bus in connection order: bit-exact
    assert np.array_equal(ref, got)


def test_stereo_panner_value_change_between_renders():
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(SR)
        s = _stereo(ctx, 128 * 30, 9)
        p = StereoPannerNode(ctx)
        p.Pan.Value = 0.25
        s.Connect(p)
        p.Connect(ctx.Destination)
        s.Start()
        out = np.zeros((2, 128 * 20), np.float32)
        ctx.Render(out, 128 * 8, 0)
        p.Pan.Value = -0.75
        ctx.Render(out, 128 * 12, 128 * 8)
        outs.append(out)
    assert np.array_equal(outs[0], outs[1])


def test_stereo_panner_automation_then_constant_again():
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(SR)
        if mk is OfflineAudioContext:
            ctx.SetOption("max_chunk_blocks", 9)
        rng = np.random.default_rng(21)
        s = _stereo(ctx, 128 * 70, 9)
        p = StereoPannerNode(ctx)
        p.Pan.SetValueAtTime(-0.8, 0.0)
        p.Pan.LinearRampToValueAtTime(0.9, 0.04)          # sweeps through the pan <= 0 / pan > 0 laws, then holds
        s.Connect(p)
        p.Connect(ctx.Destination)
        m = AudioBufferSourceNode(ctx)
        m.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(128 * 70) * 0.25).astype(np.float32), SR)
        p2 = StereoPannerNode(ctx)                         # mono path (after block 0), exponential-style curve via SetTarget
        p2.Pan.SetValueAtTime(0.7, 0.0)
        p2.Pan.SetTargetAtTime(-0.6, 0.01, 0.015)
        m.Connect(p2)
        p2.Connect(ctx.Destination)
        s.Start()
        m.Start(0.003)
        out = np.zeros((2, 128 * 64), np.float32)
        ctx.Render(out, 128 * 40 + 50, 0)
        p.Pan.Value = 0.9      # same value as the held ramp end: no recomputation, the gains of the automated run stay
        p2.Pan.Value = 0.1
        ctx.Render(out, 128 * 24 - 50, 128 * 40 + 50)
        outs.append(out)
    ref, got = outs
    assert G.rms(ref) > 1e-2
    # device cosf/sinf (gains) and exp (SetTarget curve) differ from glibc by an ulp
    assert G.rms(ref - got) <= 2e-7 and np.abs(ref - got).max() <= 2e-6


def test_delay_constant_and_tail_bit_exact():
    from graphaudio_amd import DelayNode

    def build(ctx):
        rng = np.random.default_rng(11)
        s = _stereo(ctx, 128 * 20, 4)                 # ends after 19 blocks: the delay tail outlives it
        d = DelayNode(ctx, 0.05)
        d.DelayTime.Value = 777.3 / SR
        s.Connect(d)
        d.Connect(ctx.Destination)
        m = AudioBufferSourceNode(ctx)
        m.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(128 * 40) * 0.25).astype(np.float32), SR)
        d2 = DelayNode(ctx, 1.0)                      # long line, short delay, mono input limited to one channel
        d2.Inputs[0].SetChannelCount(1)
        d2.DelayTime.Value = 0.011
        bq = BiQuadFilterNode(ctx)                    # a state-freezing node behind the delay: silent until the audio arrives
        m.Connect(d2)
        d2.Connect(bq)
        bq.Connect(ctx.Destination)
        s.Start(0.0)
        m.Start(0.013)
        return (s, d, m, d2, bq)
    ref, got = pair(build, 2, 128 * 48, pieces=[900, 2000, 128 * 20], chunk=7)
    assert G.rms(ref) > 1e-2
    assert np.array_equal(ref, got)


def test_delay_time_automation_and_value_change():
    from graphaudio_amd import DelayNode
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(SR)
        ctx.Destination.SetChannelCount(1)
        ctx.Destination.Inputs[0].SetChannelCount(1)
        rng = np.random.default_rng(12)
        m = AudioBufferSourceNode(ctx)
        m.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(128 * 60) * 0.25).astype(np.float32), SR)
        d = DelayNode(ctx, 0.02)
        d.Inputs[0].SetChannelCount(1)
        d.DelayTime.SetValueAtTime(0.001, 0.0)
        d.DelayTime.LinearRampToValueAtTime(0.019, 0.05)     # a-rate: the read position sweeps (:68)
        m.Connect(d)
        d.Connect(ctx.Destination)
        m.Start()
        out = np.zeros((1, 128 * 50), np.float32)
        ctx.Render(out, 128 * 30, 0)
        d.DelayTime.Value = 0.0033                             # cancels the timeline
        ctx.Render(out, 128 * 20, 128 * 30)
        outs.append(out)
    ref, got = outs
    assert G.rms(ref) > 1e-2
    # the delay in samples is (int)(float curve * sr): a 1-ulp difference of the device-made curve can move a read by one
    # sample at the instants where the product crosses an integer
    assert np.mean(ref != got) < 2e-3


def test_delay_ring_beyond_the_channel_count_stalls_and_resumes():
    """DelayNode.cs:62-94 writes only the rings of the input's current channels.  Block 1 sees a 3-channel input (lagged channel
    counts), so ring 2 takes 128 samples and then stalls; when the source is disposed the input has 3 channels again and ring 2
    plays those 128 samples back -- 14 blocks late."""
    from graphaudio_amd import ChannelCountMode, DelayNode

    def build(ctx):
        rng = np.random.default_rng(0)
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(1540) * 0.25).astype(np.float32), SR)
        g = GainNode(ctx)
        g.Gain.Value = 0.82
        g.Inputs[0].SetChannelCount(3)
        g.Inputs[0].SetChannelCountMode(ChannelCountMode.ClampedMax)
        d = DelayNode(ctx, 0.01)
        d.DelayTime.Value = 0.004412005290681458
        s.Connect(g)
        g.Connect(d)
        d.Connect(ctx.Destination)
        s.Start(0.00402)
        return (s, g, d)
    ref, got = pair(build, 4, 128 * 20, pieces=[174, 777, 479, 286, 81], chunk=11)
    assert np.count_nonzero(ref[2]) == 128
    assert np.array_equal(ref, got)


def test_kit_scene_buses_panners_and_post_mix_reverb():
    """bus hierarchy + panned voices + a ReverbEffect-shaped dry/wet split behind the mix."""
    frames = 128 * 120
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(SR)
        if mk is OfflineAudioContext:
            ctx.SetOption("max_chunk_blocks", 50)
        ch = G.kit_scene(ctx, voices=24, frames=frames, taps=6000)
        outs.append(G.render(ctx, ch, frames))
        ctx.Dispose()
    ref, got = outs
    assert G.rms(ref) > 1e-3
    err = G.rms(ref - got)
    assert err <= 1e-5 and err <= 2e-6 * G.rms(ref), (err, G.rms(ref))


@pytest.mark.parametrize("opts", [{}, {"max_chunk_blocks": 50}, {"coarse_min_blocks": 1, "max_chunk_blocks": 90}])
def test_kit_bus_hierarchy_with_effect_chains_per_bus(opts):
    """The other Kit shape (missing 7): AudioBus hierarchy (AudioBus.cs:76-91) with an EffectChain per bus
    (EffectChain.cs:127-149) -- reverbs in series on one bus, a reverb on a fading leaf bus, one on the master: five post-mix
    convolvers at four convolver depths, on the default route (one long chunk: formulation D), in chunks and with D forced."""
    frames = 128 * 300
    o = OracleContext(SR)
    ch = G.kit_bus_hierarchy(o, voices=24, frames=frames)
    ref = G.render(o, ch, frames)
    o.Dispose()
    h = OfflineAudioContext(SR)
    for k, v in opts.items():
        h.SetOption(k, v)
    G.kit_bus_hierarchy(h, voices=24, frames=frames)
    got = G.render(h, ch, frames)
    st = h.GetStats()
    h.Dispose()
    assert G.rms(ref) > 1e-3
    err = G.rms(ref - got)
    assert err <= 1e-5 and err <= 4e-6 * G.rms(ref), (opts, err, G.rms(ref))
    assert st["ref_order_rows"] == 0   # (gains and mixes behind the convolvers: nothing that amplifies the last bit)


def test_audio_rate_modulation_of_oscillator_panner_delay_biquad_and_offset():
    """AudioParam._input (AudioParam.cs:97-101,123-135,148-160): a ConstantSourceNode / an LFO buffer drives the parameters of
    the new nodes and of a biquad; values are clamp(intrinsic + modulation) while the modulator is non-silent."""
    from graphaudio_amd import DelayNode

    def build(ctx):
        rng = np.random.default_rng(31)
        lfo = AudioBufferSourceNode(ctx)                       # slow bipolar LFO, audio rate
        lfo.Buffer = PlayableAudioBuffer.FromMonoArray((0.4 * np.sin(2 * np.pi * np.arange(4800) / 1200.0)).astype(np.float32), SR)
        lfo.Loop = True
        cs = ConstantSourceNode(ctx)                           # a ramping control signal, starts late and stops early
        cs.Offset.SetValueAtTime(0.0, 0.0)
        cs.Offset.LinearRampToValueAtTime(300.0, 0.05)
        # FM: oscillator frequency = 440 + cs (0..300 Hz); clamped to [0, sr/2]
        osc = OscillatorNode(ctx)
        osc.Frequency.Value = 440.0
        cs.Connect(osc.Frequency)
        g = GainNode(ctx)
        g.Gain.Value = 0.2
        # auto-pan: pan = 0.1 + lfo
        pan = StereoPannerNode(ctx)
        pan.Pan.Value = 0.1
        lfo.Connect(pan.Pan)
        osc.Connect(g)
        g.Connect(pan)
        pan.Connect(ctx.Destination)
        # vibrato: delayTime = 0.004 + 0.005 * lfo ; wah: biquad frequency = 1200 + 2000 * lfo
        voice = AudioBufferSourceNode(ctx)
        voice.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(128 * 70) * 0.2).astype(np.float32), SR)
        lg1, lg2 = GainNode(ctx), GainNode(ctx)
        lg1.Gain.Value = 0.005
        lg2.Gain.Value = 2000.0
        lfo.Connect(lg1)
        lfo.Connect(lg2)
        d = DelayNode(ctx, 0.02)
        d.DelayTime.Value = 0.004
        lg1.Connect(d.DelayTime)
        bq = BiQuadFilterNode(ctx)
        bq.Frequency.Value = 1200.0
        bq.Q.Value = 2.0
        lg2.Connect(bq.Frequency)
        voice.Connect(d)
        d.Connect(bq)
        bq.Connect(ctx.Destination)
        # a constant source whose offset is itself modulated
        cs2 = ConstantSourceNode(ctx)
        cs2.Offset.Value = 0.05
        lfo.Connect(cs2.Offset)
        cs2.Connect(ctx.Destination)
        lfo.Start(0.0)
        cs.Start(0.005)
        cs.Stop(0.1)
        osc.Start(0.0)
        voice.Start(0.0)
        cs2.Start(0.01)
        return (lfo, cs, osc, g, pan, voice, lg1, lg2, d, bq, cs2)
    ref, got = pair(build, 2, 128 * 64, pieces=[3000, 128 * 20], chunk=13)
    assert G.rms(ref) > 1e-2
    err = G.rms(ref - got)
    # the curves are exact (float add + clamp); what differs is device sinf/cosf (pan gains, biquad coefficients), the
    # oscillator's sin(double), and -- at isolated samples -- a delay read one sample off where (int)(delayTime * sr) sits on an
    # integer boundary
    assert err <= 5e-4 * G.rms(ref), (err, G.rms(ref))
    assert np.mean(np.abs(ref - got) > 1e-4) < 5e-3


def test_unity_gains_hand_their_input_on():
    """A GainNode whose gain is the constant 1 (the default: buses, the splits and merges of effect chains) is not evaluated --
    its output IS its input, bit for bit what `x * 1.0f` gives.  Behind looping / late-starting / stopped sources (views that change
    from segment to segment), in front of a convolver, a biquad, a delay and a panner, chained, fanned out, with channel
    conversion at its input; a gain that is automated, modulated or set to 1 only later is evaluated as before."""
    from graphaudio_amd import ConvolverNode, DelayNode
    frames = 128 * 40

    def build(ctx):
        rng = np.random.default_rng(5)
        hold = []
        ir = PlayableAudioBuffer.FromChannelArrays([(rng.standard_normal(700) * 0.1).astype(np.float32) for _ in range(2)], SR)
        bus = GainNode(ctx)                      # unity bus with several inputs: the mix is the output
        bus.Connect(ctx.Destination)
        for v in range(6):
            s = AudioBufferSourceNode(ctx)
            nch = 1 + v % 2
            s.Buffer = PlayableAudioBuffer.FromChannelArrays([(rng.standard_normal(128 * 9 + 37 * v) * 0.2).astype(np.float32) for _ in range(nch)], SR)
            s.Loop = v % 3 != 1
            u1, u2 = GainNode(ctx), GainNode(ctx)   # two unity gains in a row
            s.Connect(u1)
            u1.Connect(u2)
            tail = u2
            if v == 0:
                cv = ConvolverNode(ctx); cv.Buffer = ir; tail.Connect(cv); tail = cv
            elif v == 1:
                bq = BiQuadFilterNode(ctx); bq.Frequency.Value = 900.0; tail.Connect(bq); tail = bq
            elif v == 2:
                dl = DelayNode(ctx, 0.05); dl.DelayTime.Value = 0.004; tail.Connect(dl); tail = dl
            elif v == 3:
                pn = StereoPannerNode(ctx); pn.Pan.Value = -0.3; tail.Connect(pn); tail = pn
            elif v == 4:
                u2.Inputs[0].SetChannelCount(2)       # channel conversion in front of a unity gain
                g = GainNode(ctx); g.Gain.SetValueAtTime(1.0, 0.0); g.Gain.LinearRampToValueAtTime(0.2, 0.08); tail.Connect(g); tail = g
            else:
                g = GainNode(ctx)                     # unity value, but modulated at audio rate: evaluated
                lfo = OscillatorNode(ctx); lfo.Frequency.Value = 7.0
                lg = GainNode(ctx); lg.Gain.Value = 0.25
                lfo.Connect(lg); lg.Connect(g.Gain); lfo.Start()
                tail.Connect(g); tail = g
                hold += [lfo, lg]
            tail.Connect(bus)
            if v % 2:
                tail.Connect(ctx.Destination)         # fan-out of a view
            s.Start(0.0 if v % 2 == 0 else 0.013 * v)
            if v == 2:
                s.Stop(0.07)
            hold += [s, u1, u2, tail]
        return hold

    for chunk in (0, 7):
        ref, got = pair(build, 2, frames, pieces=[128 * 13, 128 * 5], chunk=chunk)
        assert G.rms(ref) > 1e-3
        assert G.rms(ref - got) <= 2e-6 * max(G.rms(ref), 1.0), (chunk, G.rms(ref - got))
    # the same graph with the option off evaluates every gain: same result, more launches
    stats = []
    for opt in (1, 0):
        ctx = OfflineAudioContext(SR)
        ctx.SetOption("gain_pass_through", opt)
        ctx.SetOption("gain_fold", opt)          # (a unity gain that is not handed on would be folded as a constant one)
        ctx.Destination.SetChannelCount(2)
        hold = build(ctx)
        out = np.zeros((2, frames), np.float32)
        ctx.Render(out, frames)
        stats.append((out, ctx.GetStats()["kernel_launches"]))
        ctx.Dispose()
    assert np.array_equal(stats[0][0], stats[1][0])
    assert stats[0][1] < stats[1][1]


def test_delay_with_a_mixed_input_and_behind_constant_gains():
    """A DelayNode whose input has to be mixed (two connections, a channel conversion, a folded constant GainNode in front): the mix
    lands straight in the delay rings.  (Until round 3 the mixed slab was copied into the ring by a job of the SAME launch as the mix
    that produced it -- unordered -- and such a delay rendered garbage; the fuzz graphs only ever chained ONE node into a delay.)"""
    from graphaudio_amd import DelayNode
    frames = 128 * 34

    def build(ctx):
        rng = np.random.default_rng(11)
        hold = []

        def src(nch, when):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromChannelArrays([(rng.standard_normal(frames + 300) * 0.2).astype(np.float32) for _ in range(nch)], SR)
            s.Start(when)
            hold.append(s)
            return s

        d1 = DelayNode(ctx, 0.05); d1.DelayTime.Value = 0.004          # two mono sources, the second starting later
        src(1, 0.0).Connect(d1); src(1, 0.013).Connect(d1)
        d2 = DelayNode(ctx, 0.05); d2.DelayTime.Value = 0.0021         # mono + stereo: up-mix inside the delay's input
        src(1, 0.0).Connect(d2); src(2, 0.006).Connect(d2)
        d3 = DelayNode(ctx, 0.05); d3.DelayTime.Value = 0.0033         # a constant gain in front (folded into the ring's mix)
        g3 = GainNode(ctx); g3.Gain.Value = 0.7
        src(1, 0.002).Connect(g3); g3.Connect(d3)
        d4 = DelayNode(ctx, 0.05)                                      # two gains into one delay, automated delay time
        d4.DelayTime.SetValueAtTime(0.001, 0.0); d4.DelayTime.LinearRampToValueAtTime(0.008, 0.06)
        for k in range(2):
            g = GainNode(ctx); g.Gain.Value = 0.4 + 0.3 * k
            src(1 + k, 0.004 * k).Connect(g); g.Connect(d4); hold.append(g)
        d5 = DelayNode(ctx, 0.05); d5.DelayTime.Value = 0.002          # a delay feeding a delay
        d1.Connect(d5)
        for d in (d1, d2, d3, d4, d5):
            d.Connect(ctx.Destination)
        return hold + [d1, d2, d3, d4, d5, g3]

    for chunk in (0, 5):
        ref, got = pair(build, 2, frames, pieces=[128 * 9, 128 * 14], chunk=chunk)
        assert G.rms(ref) > 1e-2
        assert np.array_equal(ref, got), (chunk, G.rms(ref - got))


def test_constant_gains_are_folded_into_the_consumers_mix():
    """A GainNode with a constant gain and one consumer is not evaluated on its own: the consumer's mix (or down-mix, or the
    ring of a delay, or the single-term copy in front of a biquad / convolver / panner) multiplies the term -- fl(x * g), then the
    add: the reference's values bit for bit.  Voices -> volume gains -> bus gain -> master gain -> destination, with a stereo voice
    into a mono bus (down-mix of a folded term), a mono voice into a stereo bus (up-mix), chained constant gains, a gain with two
    consumers (evaluated: not folded), one automated gain."""
    frames = 128 * 30

    def build(ctx):
        rng = np.random.default_rng(23)
        hold = []
        master = GainNode(ctx); master.Gain.Value = 0.8
        master.Connect(ctx.Destination)
        stereo_bus = GainNode(ctx); stereo_bus.Gain.Value = 0.9
        mono_bus = GainNode(ctx); mono_bus.Gain.Value = 0.6
        mono_bus.Inputs[0].SetChannelCount(1)
        from graphaudio_amd import ChannelCountMode
        mono_bus.Inputs[0].SetChannelCountMode(ChannelCountMode.Explicit)
        stereo_bus.Connect(master); mono_bus.Connect(master)
        for v in range(7):
            s = AudioBufferSourceNode(ctx)
            nch = 1 + (v % 2)
            s.Buffer = PlayableAudioBuffer.FromChannelArrays([(rng.standard_normal(frames + 200) * 0.2).astype(np.float32) for _ in range(nch)], SR)
            vol = GainNode(ctx); vol.Gain.Value = 0.3 + 0.09 * v
            s.Connect(vol)
            tail = vol
            if v == 2:
                trim = GainNode(ctx); trim.Gain.Value = 1.7; tail.Connect(trim); tail = trim; hold.append(trim)      # chained
            if v == 3:
                vol.Gain.SetValueAtTime(0.2, 0.0); vol.Gain.LinearRampToValueAtTime(0.9, 0.05)                       # automated: evaluated
            tail.Connect(stereo_bus if v % 3 else mono_bus)
            if v == 4:
                tail.Connect(ctx.Destination)                                                                         # two consumers: evaluated
            if v == 5:
                bq = BiQuadFilterNode(ctx); bq.Frequency.Value = 1500.0; tail.Disconnect(); tail.Connect(bq); bq.Connect(stereo_bus); hold.append(bq)
            s.Start(0.0 if v % 2 else 0.009)
            hold += [s, vol]
        return hold + [master, stereo_bus, mono_bus]

    results = {}
    for fold in (1, 0):
        outs = []
        for mk in (OracleContext, OfflineAudioContext):
            ctx = mk(SR)
            ctx.Destination.SetChannelCount(2)
            if mk is OfflineAudioContext:
                ctx.SetOption("gain_fold", fold)
                ctx.SetOption("max_chunk_blocks", 7)
            hold = build(ctx)
            out = np.zeros((2, frames), np.float32)
            ctx.Render(out, 128 * 11)
            ctx.Render(out, frames - 128 * 11, 128 * 11)
            if mk is OfflineAudioContext:
                results[fold] = ctx.GetStats()["kernel_launches"]
            outs.append(out)
            ctx.Dispose()
        assert G.rms(outs[0]) > 1e-2
        assert np.array_equal(outs[0], outs[1]), (fold, G.rms(outs[0] - outs[1]))
    assert results[1] < results[0]


def test_parameter_modulation_that_falls_silent_inside_a_chunk():
    """The signal that modulates a panner's pan / a biquad's frequency ends (its source runs out) in the middle of a chunk: the node has
    been evaluated by its per-sample kernel, whose gains / coefficients live on the device -- the rest of the chunk has to stay on that
    kernel (with the constant parameter value), not fall back to the host-tracked state of the chunk's start.  (Found by the round-3
    generator, seed 30941: a panner modulated by a voice whose 882-frame source, played at 3.7 x, ends after two blocks.)"""
    frames = 128 * 20

    def build(ctx):
        rng = np.random.default_rng(17)
        hold = []
        carrier = AudioBufferSourceNode(ctx)
        carrier.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(frames + 64) * 0.2).astype(np.float32), SR)
        short = AudioBufferSourceNode(ctx)                        # the modulator: ends after ~2.6 blocks
        short.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(333) * 0.5).astype(np.float32), SR)
        pn = StereoPannerNode(ctx); pn.Pan.Value = 0.15
        bq = BiQuadFilterNode(ctx); bq.Frequency.Value = 1200.0; bq.Q.Value = 1.5
        dp = GainNode(ctx); dp.Gain.Value = 0.6
        df = GainNode(ctx); df.Gain.Value = 400.0
        short.Connect(dp); dp.Connect(pn.Pan)
        short.Connect(df); df.Connect(bq.Frequency)
        carrier.Connect(pn); pn.Connect(ctx.Destination)
        carrier.Connect(bq); bq.Connect(ctx.Destination)
        carrier.Start(); short.Start(0.0021)
        return hold + [carrier, short, pn, bq, dp, df]

    for chunk in (0, 9, 2):
        ref, got = pair(build, 2, frames, chunk=chunk)
        assert G.rms(ref) > 1e-2
        assert G.rms(ref - got) <= 2e-6 * G.rms(ref), (chunk, G.rms(ref - got))
