"""Pins the oracle's graph semantics (nodes, params, mixing rules, scheduling) against closed forms and scipy."""
import numpy as np
import pytest
import scipy.signal as sps

from graphaudio_amd import (ArgumentException, ArgumentOutOfRangeException, AudioBufferSourceNode, BiQuadFilterNode,
                            ChannelCountMode, ConvolverNode, FilterType, GainNode, InvalidOperationException,
                            PlayableAudioBuffer)
from tests import _graphs as G
from tests._oracle import OracleContext

SR = 48000


def mono_ctx():
    ctx = OracleContext(SR)
    ctx.Destination.SetChannelCount(1)
    return ctx


def src(ctx, x, sr=SR, connect=None, start=True, **kw):
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromMonoArray(np.asarray(x, np.float32), sr)
    if connect is not None:
        s.Connect(connect)
    if start:
        s.Start(**kw)
    return s


def render1(ctx, frames, ch=1):
    out = np.zeros((ch, frames), np.float32)
    ctx.Render(out, frames)
    return out


def test_config1_plumbing_last_block_dropped_and_dispose():
    ctx = mono_ctx()
    ch = G.config1_plumbing(ctx, voices=8, frames=128 * 6)
    out = render1(ctx, 128 * 8)
    exp = np.zeros(128 * 6, np.float32)
    for v in range(8):
        exp = exp + G.voice(v, 128 * 6)
    exp = exp * np.float32(0.125)
    assert np.array_equal(out[0, : 128 * 5], exp[: 128 * 5])
    assert np.abs(out[0, 128 * 5:]).max() == 0.0  # AudioBufferSourceNode.cs:360-368
    assert ctx.CurrentBlock == 8
    assert ctx.CurrentTime == pytest.approx(8 * 128 / SR, abs=1e-15)


def test_render_with_more_channels_than_destination_throws():
    ctx = mono_ctx()
    src(ctx, np.ones(1024), connect=ctx.Destination)
    with pytest.raises(ArgumentOutOfRangeException):
        ctx.Render(np.zeros((2, 256), np.float32), 256)  # GetChannelSpan(1) on a 1-channel buffer


def test_start_is_block_aligned():
    ctx = mono_ctx()
    g = GainNode(ctx)
    g.Inputs[0].SetChannelCount(1)
    g.Connect(ctx.Destination)
    x = np.arange(1, 2049, dtype=np.float32)
    src(ctx, x, connect=g, when=300 / SR)  # inside block 2 (frames 256..383)
    out = render1(ctx, 1024)
    assert np.abs(out[0, :256]).max() == 0
    assert np.array_equal(out[0, 256:1024], x[:768])  # plays from frame 0 of the first block with t1 > startTime


def test_default_input_upmixes_mono_to_stereo_and_lag_of_max_mode():
    ctx = OracleContext(SR)
    g = GainNode(ctx)  # default input: channelCount 2, mode Max -> mono source becomes 2 identical channels
    g.Connect(ctx.Destination)
    x = (np.random.default_rng(1).standard_normal(1024)).astype(np.float32)
    src(ctx, x, connect=g)
    out = render1(ctx, 512, ch=2)
    assert np.array_equal(out[0], x[:512]) and np.array_equal(out[1], x[:512])


def test_downmix_stereo_to_mono_rule():
    ctx = mono_ctx()
    l = np.random.default_rng(2).standard_normal(512).astype(np.float32)
    r = np.random.default_rng(3).standard_normal(512).astype(np.float32)
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromStereoArrays(np.concatenate([l, l]), np.concatenate([r, r]), SR)
    s.Connect(ctx.Destination)
    s.Start()
    ctx.Destination.Inputs[0].SetChannelCountMode(ChannelCountMode.Explicit)
    out = render1(ctx, 512)
    scale = np.float32(1.0) / np.sqrt(np.float32(2))
    assert np.array_equal(out[0], (l + r) * scale)  # (sum over channels) * 1/sqrt(N), AudioNodeInput.cs:214-228


def test_gain_param_automation_curves():
    ctx = mono_ctx()
    g = GainNode(ctx)
    g.Inputs[0].SetChannelCount(1)
    g.Connect(ctx.Destination)
    n = 48000
    src(ctx, np.ones(n + 256), connect=g)
    g.Gain.SetValueAtTime(0.0, 0.0)
    g.Gain.LinearRampToValueAtTime(1.0, 0.25)
    g.Gain.ExponentialRampToValueAtTime(0.25, 0.5)
    g.Gain.SetTargetAtTime(0.75, 0.6, 0.05)
    out = render1(ctx, n)[0]
    # block clock accumulates 128/sr; sample time = blockTime + i/sr (AudioParam.cs:116-120)
    bt = np.cumsum(np.full(n // 128, 128.0 / SR)) - 128.0 / SR
    t = (bt[:, None] + np.arange(128)[None, :] * (1.0 / SR)).reshape(-1)
    exp = np.empty(n)
    a = t < 0.25
    exp[a] = t[a] / 0.25
    b = (t >= 0.25) & (t < 0.5)
    exp[b] = 1.0 * (0.25 / 1.0) ** ((t[b] - 0.25) / 0.25)
    c = (t >= 0.5) & (t < 0.6)
    exp[c] = 0.25
    d = t >= 0.6
    exp[d] = 0.75 + (0.25 - 0.75) * np.exp(-(t[d] - 0.6) / 0.05)
    assert np.abs(out - exp).max() < 2e-7


def test_param_value_setter_cancels_events_and_ramp_after_settarget_uses_zero():
    ctx = mono_ctx()
    g = GainNode(ctx)
    g.Inputs[0].SetChannelCount(1)
    g.Connect(ctx.Destination)
    src(ctx, np.ones(4096), connect=g)
    g.Gain.LinearRampToValueAtTime(5.0, 0.01)
    g.Gain.Value = 0.5  # cancels the ramp (AudioParam.cs:37-48)
    out = render1(ctx, 1024)[0]
    assert np.all(out == np.float32(0.5))
    # LinearRamp after a SetTarget interpolates from prev.Value == 0 (AudioParam.cs:186-190)
    ctx2 = mono_ctx()
    g2 = GainNode(ctx2)
    g2.Inputs[0].SetChannelCount(1)
    g2.Connect(ctx2.Destination)
    src(ctx2, np.ones(4096), connect=g2)
    g2.Gain.SetTargetAtTime(1.0, 0.0, 0.01)
    g2.Gain.LinearRampToValueAtTime(2.0, 1024 / SR)
    out2 = render1(ctx2, 1024)[0]
    t = np.arange(1024) / SR
    exp = 2.0 * t / (1024 / SR)
    assert np.abs(out2[1:] - exp[1:]).max() < 1e-6


def cookbook(ftype, f0, q, gain_db, sr):
    """RBJ cookbook coefficients in float64, with the reference's shelf variant beta = sqrt(A)/q."""
    w0 = 2 * np.pi * f0 / sr
    c, s = np.cos(w0), np.sin(w0)
    al = s / (2 * q)
    A = 10 ** (gain_db / 40)
    be = np.sqrt(A) / q
    T = FilterType
    if ftype == T.Lowpass:
        b = [(1 - c) / 2, 1 - c, (1 - c) / 2]; a = [1 + al, -2 * c, 1 - al]
    elif ftype == T.Highpass:
        b = [(1 + c) / 2, -(1 + c), (1 + c) / 2]; a = [1 + al, -2 * c, 1 - al]
    elif ftype == T.Bandpass:
        b = [al, 0, -al]; a = [1 + al, -2 * c, 1 - al]
    elif ftype == T.Notch:
        b = [1, -2 * c, 1]; a = [1 + al, -2 * c, 1 - al]
    elif ftype == T.Allpass:
        b = [1 - al, -2 * c, 1 + al]; a = [1 + al, -2 * c, 1 - al]
    elif ftype == T.Peaking:
        b = [1 + al * A, -2 * c, 1 - al * A]; a = [1 + al / A, -2 * c, 1 - al / A]
    elif ftype == T.Lowshelf:
        b = [A * ((A + 1) - (A - 1) * c + be * s), 2 * A * ((A - 1) - (A + 1) * c), A * ((A + 1) - (A - 1) * c - be * s)]
        a = [(A + 1) + (A - 1) * c + be * s, -2 * ((A - 1) + (A + 1) * c), (A + 1) + (A - 1) * c - be * s]
    else:
        b = [A * ((A + 1) + (A - 1) * c + be * s), -2 * A * ((A - 1) + (A + 1) * c), A * ((A + 1) + (A - 1) * c - be * s)]
        a = [(A + 1) - (A - 1) * c + be * s, 2 * ((A - 1) - (A + 1) * c), (A + 1) - (A - 1) * c - be * s]
    return np.array(b) / a[0], np.array(a) / a[0]


@pytest.mark.parametrize("ftype", list(FilterType))
def test_biquad_types_match_scipy_lfilter(ftype):
    ctx = mono_ctx()
    bq = BiQuadFilterNode(ctx)
    bq.Inputs[0].SetChannelCount(1)
    bq.Type = ftype
    bq.Frequency.Value = 1500.0
    bq.Q.Value = 0.9
    bq.Gain.Value = 5.0
    bq.Connect(ctx.Destination)
    n = 128 * 60
    x = (np.random.default_rng(11).standard_normal(n + 256) * 0.25).astype(np.float32)
    src(ctx, x, connect=bq)
    out = render1(ctx, n)[0]
    b, a = cookbook(ftype, 1500.0, 0.9, 5.0, SR)
    ref = sps.lfilter(b, a, x[:n].astype(np.float64))
    assert np.sqrt(np.mean((out - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)) < 2e-5


def test_biquad_silent_input_freezes_state():
    ctx = mono_ctx()
    bq = BiQuadFilterNode(ctx)
    bq.Inputs[0].SetChannelCount(1)
    bq.Frequency.Value = 3000.0
    bq.Connect(ctx.Destination)
    x = np.ones(128 * 3 + 1, np.float32)  # plays 3 blocks, 4th block is the dropped end block
    src(ctx, x, connect=bq)
    x2 = np.ones(128 * 4, np.float32)
    s2 = src(ctx, x2, connect=bq, when=(128 * 6 + 1) / SR)  # resumes at block 7
    out = render1(ctx, 128 * 10)[0]
    assert np.abs(out[128 * 3: 128 * 6]).max() == 0.0  # silent input -> zeros, no decaying tail (:103-108)
    b, a = cookbook(FilterType.Lowpass, 3000.0, 1.0, 0.0, SR)
    ref = sps.lfilter(b, a, np.ones(128 * 6))
    # the filter continues from the frozen state as if the silence never happened
    got = np.concatenate([out[: 128 * 3], out[128 * 6: 128 * 9]])
    assert np.abs(got - ref).max() < 1e-4


def test_convolver_true_stereo_routing_and_sample_rate_check():
    ctx = OracleContext(SR)
    irs = [G.synth_ir(c, 500) for c in range(4)]
    l = (np.random.default_rng(21).standard_normal(128 * 12) * 0.25).astype(np.float32)
    r = (np.random.default_rng(22).standard_normal(128 * 12) * 0.25).astype(np.float32)
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromStereoArrays(l, r, SR)
    cv = ConvolverNode(ctx)
    cv.Normalize = False
    cv.Buffer = PlayableAudioBuffer.FromChannelArrays(irs, SR)
    s.Connect(cv).Connect(ctx.Destination)
    s.Start()
    n = 128 * 10
    out = render1(ctx, n, ch=2)
    conv = lambda x, h: np.convolve(x.astype(np.float64), h.astype(np.float64))[:n]
    expL = conv(l, irs[0]) + conv(r, irs[2])  # ConvolverNode.cs:127-144
    expR = conv(l, irs[1]) + conv(r, irs[3])
    assert np.abs(out[0] - expL).max() < 2e-5 * np.abs(expL).max()
    assert np.abs(out[1] - expR).max() < 2e-5 * np.abs(expR).max()
    cv2 = ConvolverNode(ctx)
    with pytest.raises(InvalidOperationException):
        cv2.Buffer = PlayableAudioBuffer.FromChannelArrays(irs[:1], 44100)  # :48-49


def test_convolver_keeps_tail_after_source_ends():
    ctx = OracleContext(SR)
    ir = np.zeros(1000, np.float32)
    ir[999] = 1.0
    x = np.ones(128 * 2 + 1, np.float32)
    s = src(ctx, x)
    cv = ConvolverNode(ctx)
    cv.Normalize = False
    cv.Buffer = PlayableAudioBuffer.FromMonoArray(ir, SR)
    s.Connect(cv).Connect(ctx.Destination)
    ctx.Destination.SetChannelCount(1)
    out = render1(ctx, 128 * 12)[0]
    exp = np.zeros(128 * 12)
    exp[999: 999 + 256] = 1.0  # the convolver keeps running on silent input (ConvolverNode.cs:102-155)
    assert np.abs(out - exp).max() < 1e-5


def test_resampled_source_matches_primitive_and_ends():
    ctx = mono_ctx()
    n_in = 2000
    x = np.sin(np.arange(n_in) * 0.01).astype(np.float32)
    s = src(ctx, x, sr=44100, connect=ctx.Destination)
    out = render1(ctx, 128 * 20)[0]
    from tests import _oracle as O
    prim, consumed, produced = O.resample(x, 128 * 20, 44100 / 48000.0)
    k = (produced // 128) * 128
    assert np.array_equal(out[: k - 128], prim[: k - 128])
    assert np.abs(out[128 * 18:]).max() == 0.0


def test_start_twice_errors_after_first_block_and_cycle_detection():
    ctx = mono_ctx()
    s = src(ctx, np.ones(4096), connect=ctx.Destination)
    render1(ctx, 128)
    with pytest.raises(InvalidOperationException):
        s.Start()
    # A cycle is NOT an error in the reference: ProcessInternal's memo check (Nodes/AudioNode.cs:154) runs before the
    # _isProcessing check (:157-160), so re-entering a node returns immediately and the loop closes with the node's
    # previous-block buffer -- an implicit one-block feedback delay.  The oracle restates that; so does the device path since
    # round 4 (one block per chunk, tests/test_gpu_cycles.py).
    g1, g2 = GainNode(ctx), GainNode(ctx)
    g1.Gain.Value = 0.5
    g1.Connect(g2)
    g2.Connect(g1)
    g2.Connect(ctx.Destination)
    out = render1(ctx, 128 * 3)
    assert np.isfinite(out).all()


def test_argument_validation():
    ctx = mono_ctx()
    with pytest.raises(ArgumentOutOfRangeException):
        ctx.Render(np.zeros((1, 8), np.float32), 0)
    with pytest.raises(ArgumentException):
        ctx.Render(np.zeros((1, 8), np.float32), 16)
    with pytest.raises(ArgumentOutOfRangeException):
        ctx.Destination.SetChannelCount(33)
    g = GainNode(ctx)
    with pytest.raises(ArgumentException):
        g.Gain.ExponentialRampToValueAtTime(0.0, 1.0)


def test_audio_rate_param_modulation_semantics():
    """value[i] = clamp(intrinsic + modulation[i]) when the modulation input is non-silent (AudioParam.cs:123-135)."""
    ctx = mono_ctx()
    g = GainNode(ctx)
    g.Inputs[0].SetChannelCount(1)
    g.Gain.Value = 0.25
    g.Connect(ctx.Destination)
    src(ctx, np.ones(2048), connect=g)
    mod = np.linspace(-1, 1, 2048).astype(np.float32)
    m = AudioBufferSourceNode(ctx)
    m.Buffer = PlayableAudioBuffer.FromMonoArray(mod, SR)
    m.Connect(g.Gain)
    m.Start()
    out = render1(ctx, 1024)[0]
    assert np.array_equal(out, np.float32(0.25) + mod[:1024])


def test_looping_resampled_source_is_seamless_and_playback_rate_is_k_rate():
    """Loop + resampling (AudioBufferSourceNode.cs:236-358, the 512-sample wrap buffer): a loop that holds whole
    periods of a sine, read at rate r, is that sine at r times the frequency -- across every loop wrap."""
    period, rate = 100, 0.5
    x = np.sin(2 * np.pi * np.arange(800) / period)
    ctx = mono_ctx()
    s = src(ctx, x, connect=ctx.Destination, start=False)
    s.Loop = True
    s.PlaybackRate.Value = rate
    s.Start(0.0)
    out = render1(ctx, 128 * 40)[0].astype(np.float64)
    # first output = in[1] (CubicResampler primes 4 samples, CubicResampler.cs:31-38): phase offset of one input sample
    want = np.sin(2 * np.pi * (1 + rate * np.arange(out.size)) / period)
    assert np.abs(out - want).max() < 2e-5   # Catmull-Rom error of a period-100 sine
    # the rate is sampled once per block (k-rate, :165): a ramp on it moves the pitch in 128-frame steps
    ctx = mono_ctx()
    s = src(ctx, x, connect=ctx.Destination, start=False)
    s.Loop = True
    s.PlaybackRate.SetValueAtTime(0.5, 0.0)
    s.PlaybackRate.LinearRampToValueAtTime(1.3, 128 * 20 / SR)   # (never exactly 1.0: that block would take the copy path)
    s.Start(0.0)
    out = render1(ctx, 128 * 20)[0].astype(np.float64)
    phase = 1.0
    want = np.zeros_like(out)
    for b in range(20):
        r = np.float32(0.5 + (1.3 - 0.5) * (b * 128 / SR) / (128 * 20 / SR))
        want[b * 128:(b + 1) * 128] = np.sin(2 * np.pi * (phase + float(r) * np.arange(128)) / period)
        phase += float(r) * 128
    assert np.abs(out - want).max() < 2e-5
