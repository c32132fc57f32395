"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerance: north_star asks for <= 1e-5 RMS per sample (float32).  Integer/elementwise paths (mix, gain with constant
value, biquad with constant coefficients, sources) are compared bit-exactly; the convolver (f32 MFMA accumulation
order differs from the reference's sequential unfused order) and the automation curves are compared to 1e-5 RMS abs
and to a much tighter bus-relative bound.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import (AudioBufferSourceNode, ConvolverNode, GainNode, OfflineAudioContext, PlayableAudioBuffer,
                            ArgumentOutOfRangeException)
from tests import _graphs as G
from tests._oracle import OracleContext

TOL_RMS = 1e-5


def both(builder, nrender, **kw):
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(48000)
        ch = builder(ctx, **kw)
        outs.append(G.render(ctx, ch, nrender))
        ctx.Dispose()
    return outs


def test_config1_plumbing_bit_exact():
    ref, got = both(G.config1_plumbing, 128 * 8, voices=8, frames=128 * 6)
    assert np.array_equal(ref, got)
    assert np.abs(ref[:, 128 * 5:]).max() == 0.0  # the final block of every one-shot is dropped


def test_biquad_chain_bit_exact():
    ref, got = both(G.config2_biquad, 128 * 40, voices=32, frames=128 * 50)
    assert G.rms(ref) > 1e-3
    assert np.array_equal(ref, got)


def test_biquad_default_stereo_upmix():
    ref, got = both(G.config2_biquad, 128 * 20, voices=8, frames=128 * 30, mono=False)
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("taps", [1, 100, 128, 129, 1000, 4096])
def test_convolver_small(taps):
    ref, got = both(G.config3_convolver, 128 * 48, voices=3, taps=taps, frames=128 * 48)
    err = G.rms(ref - got)
    assert G.rms(ref) > 1e-4
    assert err <= TOL_RMS, err
    assert err <= 2e-6 * max(G.rms(ref), 1e-3)


def test_convolver_shared_ir_many_voices():
    ref, got = both(G.config3_convolver, 128 * 96, voices=40, taps=8192, frames=128 * 96)
    err = G.rms(ref - got)
    assert err <= TOL_RMS, err
    assert err / G.rms(ref) < 2e-6


def test_convolver_state_persists_across_renders():
    frames = 128 * 64
    ctxs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(48000)
        ch = G.config3_convolver(ctx, voices=4, taps=3000, frames=frames)
        ctxs.append((ctx, ch))
    (o, ch), (h, _) = ctxs
    ref = G.render(o, ch, frames)
    # HIP: render in uneven pieces (partial blocks -> leftover cache, several chunks)
    parts = [100, 128 * 3 + 7, 1, 128 * 20, frames]
    got = np.zeros((ch, frames), np.float32)
    pos = 0
    for p in parts:
        n = min(p, frames - pos)
        if n <= 0:
            break
        h.Render(got, n, pos)
        pos += n
    assert pos == frames
    err = G.rms(ref - got)
    assert err <= TOL_RMS and err / G.rms(ref) < 2e-6


def test_config4_resample_eq_automation():
    ref, got = both(G.config4_eq, 128 * 300, voices=6, frames=128 * 300)
    err = G.rms(ref - got)
    assert G.rms(ref) > 1e-5
    assert err <= 1e-6, err


def test_render_too_many_channels_raises():
    ctx = OfflineAudioContext(48000)
    G.config1_plumbing(ctx, voices=2, frames=256)
    out = np.zeros((2, 256), np.float32)
    with pytest.raises(ArgumentOutOfRangeException):
        ctx.Render(out, 256)


def test_private_ir_per_voice():
    """Unique IR per voice (config 3 variant): served by the per-node formulation (time x IR-channels MFMA tiles)."""
    ref, got = both(G.config3_convolver, 128 * 40, voices=5, taps=3000, frames=128 * 40, shared=False)
    err = G.rms(ref - got)
    assert err <= TOL_RMS and err / G.rms(ref) < 2e-6


def _config5(ctx, sources=4, taps=2000, frames=128 * 30, ir_channels=16):
    """sources x 16-channel PartitionedConvolver (HRTF-style multi-IR), destination 16 ch (BASELINE.json configs[4])."""
    ctx.Destination.SetChannelCount(ir_channels)
    for v in range(sources):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), 48000)
        cv = ConvolverNode(ctx)
        cv.Buffer = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, taps, seed0=7 + 100 * v) for c in range(ir_channels)], 48000)
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()
    return ir_channels


def test_config5_multichannel_ir():
    ref, got = both(_config5, 128 * 30)
    assert ref.shape[0] == 16
    err = G.rms(ref - got)
    assert err <= TOL_RMS and err / G.rms(ref) < 2e-6


def test_private_ir_stereo_source_and_chunked_state():
    """Stereo source into a private 2-channel IR (distinct inputs per channel) rendered in uneven pieces."""
    frames = 128 * 50

    def build(ctx):
        l, r = G.voice(70, frames + 256), G.voice(71, frames + 256)
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromStereoArrays(l, r, 48000)
        cv = ConvolverNode(ctx)
        cv.Buffer = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 5000) for c in range(2)], 48000)
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()
        return 2

    o = OracleContext(48000)
    build(o)
    ref = G.render(o, 2, frames)
    h = OfflineAudioContext(48000)
    h.SetOption("max_chunk_blocks", 13)
    build(h)
    got = np.zeros_like(ref)
    pos = 0
    for n in (300, 128 * 7 + 5, 128 * 20, frames):
        n = min(n, frames - pos)
        if n > 0:
            h.Render(got, n, pos)
            pos += n
    err = G.rms(ref - got)
    assert err <= TOL_RMS and err / G.rms(ref) < 2e-6


def _automated_biquad(ctx):
    from graphaudio_amd import BiQuadFilterNode, FilterType
    n = 128 * 60
    for v, ft in enumerate([FilterType.Lowpass, FilterType.Peaking, FilterType.Highshelf]):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(300 + v, n + 256), 48000)
        bq = BiQuadFilterNode(ctx)
        bq.Type = ft
        bq.Frequency.SetValueAtTime(300.0, 0.0)
        bq.Frequency.ExponentialRampToValueAtTime(6000.0, 0.1)
        bq.Q.SetValueAtTime(0.7, 0.0)
        bq.Q.LinearRampToValueAtTime(4.0, 0.05)
        bq.Gain.SetValueAtTime(-6.0, 0.0)
        bq.Gain.SetValueAtTime(6.0, 0.06)
        s.Connect(bq).Connect(ctx.Destination)   # default input: mono source up-mixed to 2 channels (trigger state spans channels)
        s.Start()
    return 2


def test_biquad_parameter_automation():
    """a-rate frequency / Q with the per-sample coefficient refresh of BiQuadFilterNode.cs:123-134 (device sinf/cosf/powf)."""
    ref, got = both(_automated_biquad, 128 * 60)
    assert G.rms(ref) > 1e-3
    err = G.rms(ref - got)
    assert err <= TOL_RMS and err / G.rms(ref) < 2e-5, err


def _modulated_gain(ctx):
    n = 128 * 20
    ctx.Destination.SetChannelCount(1)
    carrier = AudioBufferSourceNode(ctx)
    carrier.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(400, n + 256), 48000)
    lfo = AudioBufferSourceNode(ctx)
    t = np.arange(n + 256) / 48000.0
    lfo.Buffer = PlayableAudioBuffer.FromStereoArrays((0.8 * np.sin(2 * np.pi * 50 * t)).astype(np.float32),
                                                      (0.8 * np.cos(2 * np.pi * 30 * t)).astype(np.float32), 48000)
    g = GainNode(ctx)
    g.Inputs[0].SetChannelCount(1)
    g.Gain.Value = 0.5
    lfo.Connect(g.Gain)            # stereo LFO -> 1-channel param input: (L + R) / sqrt(2), then clamp(intrinsic + mod)
    carrier.Connect(g).Connect(ctx.Destination)
    carrier.Start()
    lfo.Start(128 * 3 / 48000.0 + 1e-9)   # modulation starts three blocks in
    return 1


def test_gain_audio_rate_modulation_bit_exact():
    ref, got = both(_modulated_gain, 128 * 20)
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("taps,voices,time_fft", [
    (9000, 3, 1),      # P = 71   -> block-axis FFT, smallest kernel (N2 = 1024 although 4 * 128 = 512 would do)
    (20000, 3, 1),     # P = 157  -> block-axis FFT with N2 = 1024
    (100000, 2, 1),    # P = 782  -> N2 = 4096 (104 KB of LDS)
    (40000, 3, 0),     # P = 313  -> formulation B, two tap segments of 256
    (140000, 9, 0),    # P = 1094 -> formulation A (9 voices share the IR), two tap segments of 1024
    (140000, 2, 1),    # P = 1094 -> outside the block-axis FFT range: formulation B
])
def test_convolver_formulations_and_tap_ranges(taps, voices, time_fft):
    frames = 128 * 150
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(48000)
        ctx.SetOption("time_fft", time_fft)
        ch = G.config3_convolver(ctx, voices=voices, taps=taps, frames=frames)
        outs.append(G.render(ctx, ch, frames))
        ctx.Dispose()
    ref, got = outs
    err = G.rms(ref - got)
    assert G.rms(ref) > 1e-4
    assert err <= TOL_RMS and err / G.rms(ref) < 2e-6, (err, err / G.rms(ref))


def _chained_shared_ir(ctx, voices=10, taps=700, frames=128 * 40, hold=None):
    """`voices` sources, each through TWO convolvers in series that all share one impulse response (the shared-IR
    formulation then holds rows of one IR at two convolver depths)."""
    rng = np.random.default_rng(77)
    ctx.Destination.SetChannelCount(2)
    ir = PlayableAudioBuffer.FromChannelArrays([(rng.standard_normal(taps) * 0.05).astype(np.float32) for _ in range(2)], 48000)
    for v in range(voices):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromChannelArrays([(rng.standard_normal(frames) * 0.25).astype(np.float32)], 48000)
        c1 = ConvolverNode(ctx)
        c1.Buffer = ir
        c2 = ConvolverNode(ctx)
        c2.Buffer = ir
        s.Connect(c1)
        c1.Connect(c2)
        c2.Connect(ctx.Destination)
        s.Start(0.0)
        if hold is not None:
            hold.append((s, c1, c2))
    return 2


@pytest.mark.parametrize("chunk", [0, 7])
def test_shared_ir_convolvers_in_series(chunk):
    frames = 128 * 40
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(48000)
        ctx.SetOption("time_fft", 0)
        if chunk:
            ctx.SetOption("max_chunk_blocks", chunk)
        ch = _chained_shared_ir(ctx, frames=frames)
        outs.append(G.render(ctx, ch, frames))
        ctx.Dispose()
    ref, got = outs
    err = G.rms(ref - got)
    assert G.rms(ref) > 1e-4
    assert err <= TOL_RMS and err / G.rms(ref) < 2e-6, (err, err / G.rms(ref))


def test_shared_ir_row_follows_depth_change():
    """A graph edit between renders puts a convolver behind another one: its filter state must move with it."""
    frames = 128 * 30
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(48000)
        ctx.SetOption("time_fft", 0)
        hold = []
        ch = _chained_shared_ir(ctx, frames=frames, hold=hold)
        out = np.zeros((ch, frames), np.float32)
        ctx.Render(out, 128 * 12, 0)
        # voice 0: source -> c1 -> c2   becomes   source -> c2 -> c1 (c1 depth 0 -> 1, c2 depth 1 -> 0)
        s, c1, c2 = hold[0]
        s.Disconnect(c1)
        c1.Disconnect(c2)
        c2.Disconnect(ctx.Destination)
        s.Connect(c2)
        c2.Connect(c1)
        c1.Connect(ctx.Destination)
        ctx.Render(out, frames - 128 * 12, 128 * 12)
        outs.append(out)
        ctx.Dispose()
    ref, got = outs
    err = G.rms(ref - got)
    assert err <= TOL_RMS and err / G.rms(ref) < 2e-6, (err, err / G.rms(ref))


def test_convolver_buffer_swap_back_while_queued_is_ignored():
    """ConvolverNode.cs:30 compares with the buffer of the last EXECUTED swap: A -> B -> A between two blocks ends on B."""
    rng = np.random.default_rng(5)
    irA = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(300) * 0.1).astype(np.float32), 48000)
    irB = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(500) * 0.1).astype(np.float32), 48000)
    x = (rng.standard_normal(128 * 20) * 0.25).astype(np.float32)
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(48000)
        ctx.Destination.SetChannelCount(1)
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(x, 48000)
        c = ConvolverNode(ctx)
        c.Buffer = irA
        s.Connect(c)
        c.Connect(ctx.Destination)
        s.Start(0.0)
        out = np.zeros((1, 128 * 16), np.float32)
        ctx.Render(out, 128 * 6, 0)
        c.Buffer = irB
        c.Buffer = irA   # ignored: the executed buffer is still irA
        ctx.Render(out, 128 * 10, 128 * 6)
        outs.append(out)
        ctx.Dispose()
    ref, got = outs
    assert G.rms(ref[:, 128 * 8:]) > 1e-4
    assert G.rms(ref - got) <= 2e-6 * G.rms(ref)


_GSR_CASES = {
    # name: (loop at start, frame at which Loop is toggled (0 = never), time of the playbackRate event (None = constant
    #        rate), rate after the event, start offset, duration, buffer sample rate, channels)
    "timeline": (False, 0, 0.0367, 0.99, 0.0, float("inf"), 48000, 1),
    "timeline+loop": (True, 0, 0.0367, 0.99, 0.0, float("inf"), 48000, 2),
    "timeline+loop toggled on": (False, 1579, 0.0367, 0.99, 0.0, float("inf"), 48000, 1),
    "timeline+loop toggled on after the data ended": (False, 1579, 0.0367, 0.99, 0.00636, 0.088, 48000, 1),
    "timeline, offset+duration": (False, 0, 0.0367, 0.99, 0.00636, 0.088, 48000, 1),
    "44.1k loop": (True, 0, None, 1.0, 0.0, float("inf"), 44100, 2),
    "44.1k loop toggled off": (True, 2000, None, 1.0, 0.0, float("inf"), 44100, 1),
    "fast loop (rate 3.7)": (True, 0, None, 3.7, 0.0, float("inf"), 48000, 1),
    "rate crosses exactly 1.0": (True, 0, 0.01, 1.0, 0.0, float("inf"), 44100, 1),
}


@pytest.mark.parametrize("case", list(_GSR_CASES))
def test_source_general_replay_bit_exact(case):
    """Looping + resampling, a playbackRate that moves while the source plays, Loop toggled between renders: the host
    replays AudioBufferSourceNode.Process on indices, the device does the sample arithmetic -- bit-exact."""
    loop0, toggle_at, t_ev, rate_after, offset, duration, sr, nch = _GSR_CASES[case]
    rng = np.random.default_rng(3)
    data = [(rng.standard_normal(995 if duration != float("inf") else 3089) * 0.25).astype(np.float32) for _ in range(nch)]
    frames = 128 * 40
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(48000)
        ctx.Destination.SetChannelCount(nch)
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromChannelArrays(data, sr)
        s.Loop = loop0
        if loop0:
            s.LoopStart = 300 / sr
            s.LoopEnd = 900 / sr
        if t_ev is not None:
            s.PlaybackRate.SetValueAtTime(rate_after, t_ev)
            s.PlaybackRate.LinearRampToValueAtTime(1.0004, t_ev + 0.08)
        else:
            s.PlaybackRate.Value = rate_after
        s.Connect(ctx.Destination)
        s.Start(0.0, offset, duration)
        out = np.zeros((nch, frames), np.float32)
        if toggle_at:
            ctx.Render(out, toggle_at, 0)
            s.Loop = not s.Loop
            s.PlaybackRate.Value = 1.25 if t_ev is None else s.PlaybackRate.Value
            ctx.Render(out, frames - toggle_at, toggle_at)
        else:
            ctx.Render(out, frames)
        outs.append(out)
        ctx.Dispose()
    ref, got = outs
    assert G.rms(ref) > 1e-3
    assert np.array_equal(ref, got)


@pytest.mark.parametrize("option,value", [("fft64", 1), ("tconv_radix16", 0), ("time_fft", 0)])
def test_convolver_kernel_variants_behind_options(option, value):
    """The documented options select other kernels for the same arithmetic: double-precision 256-point transforms, the
    radix-8 block-axis kernel (one FFT length per chunk), the direct matrix-core formulation.  Several chunks, so that the
    history hand-over between the plane pairs runs with each of them."""
    frames = 128 * 150
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(48000)
        ctx.SetOption(option, value)
        ctx.SetOption("max_chunk_blocks", 64)
        ch = G.config3_convolver(ctx, voices=3, taps=20000, frames=frames)
        outs.append(G.render(ctx, ch, frames))
        ctx.Dispose()
    ref, got = outs
    err = G.rms(ref - got)
    assert G.rms(ref) > 1e-4
    assert err <= TOL_RMS and err / G.rms(ref) < 2e-6, (err, err / G.rms(ref))
