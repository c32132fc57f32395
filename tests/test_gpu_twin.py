"""Twin channels (option twin_channels, ga_stats.twin_rows): a mono signal in a stereo node or input is the SAME numbers on every
channel in the reference (AudioNodeInput.cs:182-244 copies the mono mix to all channels; BiQuadFilterNode.cs:117-146 then walks
every channel from its own -- equal -- state), so the device evaluates such channels once: one biquad job whose end state goes to
every channel's slot, one mix job that writes two rows.  The render has to be bit-identical with the option off, and the shortcut
has to let go the moment the channels stop being the same (another input joins, the states were never equal)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import (AudioBufferSourceNode, BiQuadFilterNode, DelayNode, FilterType, GainNode, OfflineAudioContext,
                            PlayableAudioBuffer)
from tests import _graphs as G
from tests._oracle import OracleContext

SR = 48000


def _render(make_ctx, build, frames, piece, edit=None, **opts):
    ctx = make_ctx(SR)
    for k, v in opts.items():
        ctx.SetOption(k, v)
    h = build(ctx)
    out = np.zeros((2, frames), np.float32)
    pos, k = 0, 0
    while pos < frames:
        n = min(piece, frames - pos)
        ctx.Render(out, n, pos)
        pos += n
        k += 1
        if edit:
            edit(ctx, h, k)
    st = ctx.GetStats() if make_ctx is OfflineAudioContext else None
    ctx.Dispose()
    return out, st


def _eq_chain(ctx, src, bands=((FilterType.Lowshelf, 100.0, 6.0), (FilterType.Peaking, 1000.0, -6.0), (FilterType.Highshelf, 8000.0, 3.0))):
    node = src
    made = []
    for ft, f, gdb in bands:
        bq = BiQuadFilterNode(ctx)
        bq.Type = ft
        bq.Frequency.Value = f
        bq.Gain.Value = gdb
        node = node.Connect(bq)
        made.append(bq)
    return node, made


def test_mono_voices_through_stereo_equalisers_are_walked_once():
    frames, piece = 128 * 90, 128 * 30

    def build(ctx):
        for v in range(12):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), SR)
            tail, _ = _eq_chain(ctx, s)
            g = GainNode(ctx)
            g.Gain.SetValueAtTime(0.0, 0.0)
            g.Gain.LinearRampToValueAtTime(0.25, 0.1)
            tail.Connect(g).Connect(ctx.Destination)
            s.Start(0.0 if v % 3 else 0.004 * v)
        return None

    a, sa = _render(OfflineAudioContext, build, frames, piece)
    b, sb = _render(OfflineAudioContext, build, frames, piece, twin_channels=0)
    o, _ = _render(OracleContext, build, frames, piece)
    assert sb["twin_rows"] == 0 and sa["twin_rows"] > 0
    assert G.rms(a) > 1e-3 and np.array_equal(a, b) and np.array_equal(a, o)
    assert np.array_equal(a[0], a[1])   # (mono material: both channels of the bus are the same row)


def test_a_stereo_source_joining_later_ends_the_twin_walk_with_the_right_states():
    frames, piece = 128 * 60, 128 * 10
    rng = np.random.default_rng(5)
    left = (rng.standard_normal(frames) * 0.2).astype(np.float32)
    right = (rng.standard_normal(frames) * 0.2).astype(np.float32)

    def build(ctx):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(3, frames + 256), SR)
        bq = BiQuadFilterNode(ctx)
        bq.Type = FilterType.Peaking
        bq.Frequency.Value = 700.0
        bq.Q.Value = 4.0
        bq.Gain.Value = 9.0
        bq2 = BiQuadFilterNode(ctx)
        bq2.Type = FilterType.Lowpass
        bq2.Frequency.Value = 3000.0
        s.Connect(bq).Connect(bq2).Connect(ctx.Destination)
        s.Start()
        st = AudioBufferSourceNode(ctx)
        st.Buffer = PlayableAudioBuffer.FromChannelArrays([left, right], SR)
        return bq, st

    def edit(ctx, h, k):
        bq, st = h
        if k == 2:
            st.Connect(bq)     # from here on the two channels of the filters differ
            st.Start()
        if k == 4:
            st.Disconnect()    # ... and stay different (their states are) after the stereo source has gone

    a, sa = _render(OfflineAudioContext, build, frames, piece, edit)
    b, _ = _render(OfflineAudioContext, build, frames, piece, edit, twin_channels=0)
    o, _ = _render(OracleContext, build, frames, piece, edit)
    assert sa["twin_rows"] > 0
    assert np.array_equal(a, b) and np.array_equal(a, o)
    assert not np.array_equal(a[0, 128 * 45:], a[1, 128 * 45:])


def test_channels_whose_states_were_never_equal_are_not_twins():
    """The filter runs on ONE channel first (explicit mono input): channel 1's state stays zero while channel 0's moves.  When the input
    goes stereo the same mono signal arrives on both channels, but from different states -- two walks."""
    frames, piece = 128 * 40, 128 * 10

    def build(ctx):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(9, frames + 256), SR)
        bq = BiQuadFilterNode(ctx)
        bq.Type = FilterType.Bandpass
        bq.Frequency.Value = 500.0
        bq.Q.Value = 8.0
        bq.Inputs[0].SetChannelCount(1)
        s.Connect(bq).Connect(ctx.Destination)
        s.Start()
        return bq

    def edit(ctx, bq, k):
        if k == 2:
            bq.Inputs[0].SetChannelCount(2)

    a, _ = _render(OfflineAudioContext, build, frames, piece, edit)
    b, _ = _render(OfflineAudioContext, build, frames, piece, edit, twin_channels=0)
    o, _ = _render(OracleContext, build, frames, piece, edit)
    assert np.array_equal(a, b) and np.array_equal(a, o)


def test_twin_rows_into_delay_rings_and_split_cascades():
    frames, piece = 128 * 400, 128 * 200

    def build(ctx):
        for v in range(3):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(20 + v, frames + 256), SR)
            bq = BiQuadFilterNode(ctx)
            bq.Type = FilterType.Lowpass
            bq.Frequency.Value = 900.0 + 400.0 * v
            d = DelayNode(ctx, 0.5)
            d.DelayTime.Value = 0.011 * (v + 1)
            s.Connect(bq).Connect(d).Connect(ctx.Destination)
            bq.Connect(ctx.Destination)
            s.Start()
        return None

    a, sa = _render(OfflineAudioContext, build, frames, piece)                         # (long chunks: the cascades are split along time)
    b, sb = _render(OfflineAudioContext, build, frames, piece, twin_channels=0)
    assert sa["twin_rows"] > 0 and sa["biquad_split_cascades"] > 0
    assert np.array_equal(a, b)
    c, _ = _render(OfflineAudioContext, build, frames, piece, biquad_time_split=0)
    o, _ = _render(OracleContext, build, frames, piece)
    assert np.array_equal(c, o)


@pytest.mark.parametrize("voices", [256, 289, 330])
def test_a_bus_of_hundreds_of_terms_takes_the_wide_kernel_in_term_order(voices):
    """mix_wide_kernel (ga_kernels.hip): 64 descriptors per load, 32 terms in flight, a ragged last batch; plain terms, folded constant
    gains and folded gain curves side by side; voices that start inside the render (several segments, unaligned first frames)"""
    frames, piece = 128 * 24, 128 * 12

    def build(ctx):
        for v in range(voices):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, 128 * (10 + v % 17) + 13 * (v % 5)), SR)
            if v % 3 == 0:
                s.Connect(ctx.Destination)
            else:
                g = GainNode(ctx)
                if v % 3 == 1:
                    g.Gain.Value = 0.1 + 0.001 * v
                else:
                    g.Gain.SetValueAtTime(0.05, 0.0)
                    g.Gain.LinearRampToValueAtTime(0.3, 0.03 + 0.0001 * v)
                s.Connect(g).Connect(ctx.Destination)
            s.Start(0.0 if v % 4 else 0.0007 * (v % 29))
        return None

    a, sa = _render(OfflineAudioContext, build, frames, piece)
    o, _ = _render(OracleContext, build, frames, piece)
    assert G.rms(a) > 1e-3 and np.array_equal(a, o)
