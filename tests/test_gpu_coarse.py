"""Formulation D of the convolver (coarse partitions of 8192 samples, consumer sums fused in the frequency domain,
graphaudio_amd/csrc/ga_coarse.hip) against the CPU oracle.  By default a convolver takes formulation D only when its first
chunk is long (option coarse_min_blocks = 256); these tests force it (coarse_min_blocks = 1) so that short renders, uneven
pieces and graph edits exercise its state handling: the state of a node is the last P' x 8192 input samples per channel.

Tolerance: north_star's 1e-5 RMS absolute, plus a bus-relative bound (float32 transforms of 4096 complex points: ~4e-7).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import (AudioBufferSourceNode, BiQuadFilterNode, ChannelCountMode, ConvolverNode, FilterType, GainNode,
                            OfflineAudioContext, PlayableAudioBuffer)
from tests import _graphs as G
from tests._oracle import OracleContext

SR = 48000
TOL_RMS = 1e-5
REL = 2e-6


def hip(**opts):
    ctx = OfflineAudioContext(SR)
    ctx.SetOption("coarse_min_blocks", 1)
    for k, v in opts.items():
        ctx.SetOption(k, v)
    return ctx


def check(ref, got, rel=REL):
    err, sig = G.rms(ref - got), G.rms(ref)
    assert sig > 1e-5
    assert err <= TOL_RMS, err
    assert err <= rel * sig, (err, sig, err / sig)


def pair(builder, frames, pieces=None, **opts):
    o = OracleContext(SR)
    ch = builder(o)
    ref = G.render(o, ch, frames)
    h = hip(**opts)
    builder(h)
    got = np.zeros_like(ref)
    pos = 0
    for n in (pieces or [frames]):
        n = min(n, frames - pos)
        if n > 0:
            h.Render(got, n, pos)
            pos += n
    if pos < frames:
        h.Render(got, frames - pos, pos)
    st = h.GetStats()
    h.Dispose()
    o.Dispose()
    return ref, got, st


def used_coarse(st):
    return st["stage_launches"][5] > 0 and st["stage_launches"][3] == 0   # coarse_fwd ran, no A/B/C partition sum


@pytest.mark.parametrize("taps", [8193, 8200, 16384, 20000, 40000, 65536])
def test_tap_counts(taps):
    """P' = 2 .. 8 coarse partitions, last partition short; 3 voices fused into one sum per channel."""
    frames = 128 * 600
    ref, got, st = pair(lambda c: G.config3_convolver(c, voices=3, taps=taps, frames=frames), frames)
    assert used_coarse(st)
    check(ref, got)


def test_131072_taps_sixteen_partitions():
    frames = 128 * 1100
    ref, got, st = pair(lambda c: G.config3_convolver(c, voices=2, taps=131072, frames=frames), frames)
    assert used_coarse(st)
    check(ref, got)
    tail = slice(128 * 1030, None)   # every partition populated
    assert G.rms(ref[:, tail] - got[:, tail]) <= REL * G.rms(ref[:, tail])


def test_state_across_uneven_pieces():
    """Partial blocks (leftover cache), one-block chunks, chunks shorter and longer than the history, many chunks."""
    frames = 128 * 500
    pieces = [100, 128 * 3 + 7, 1, 128 * 70, 128, 128 * 200 - 5, 77]
    ref, got, st = pair(lambda c: G.config3_convolver(c, voices=4, taps=30000, frames=frames), frames, pieces, max_chunk_blocks=96)
    assert used_coarse(st) and st["chunks"] > 8
    check(ref, got)


def test_more_than_64_coarse_blocks_in_one_chunk():
    """A chunk of 9,000 blocks is 141 coarse blocks: the multiply-accumulate jobs are split along time (<= 64 blocks each)."""
    frames = 128 * 9000
    ref, got, st = pair(lambda c: G.config3_convolver(c, voices=1, taps=9000, frames=frames), frames, max_chunk_blocks=32768)
    assert used_coarse(st) and st["chunks"] == 1
    check(ref, got)


def test_unique_ir_per_voice():
    frames = 128 * 400
    ref, got, st = pair(lambda c: G.config3_convolver(c, voices=5, taps=20000, frames=frames, shared=False), frames)
    assert used_coarse(st)
    check(ref, got)


def _true_stereo(ctx, frames):
    irs = [G.synth_ir(c, 17000) for c in range(4)]
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromStereoArrays(G.voice(50, frames + 256), G.voice(51, frames + 256), SR)
    cv = ConvolverNode(ctx)
    cv.Buffer = PlayableAudioBuffer.FromChannelArrays(irs, SR)
    s.Connect(cv).Connect(ctx.Destination)
    s.Start()
    return 2


def test_true_stereo_four_channel_ir():
    """outL = L * h0 + R * h2, outR = L * h1 + R * h3 (ConvolverNode.cs:127-151): four terms, two accumulators."""
    frames = 128 * 400
    ref, got, st = pair(lambda c: _true_stereo(c, frames), frames, [128 * 150, 128 * 250])
    assert used_coarse(st)
    check(ref, got)


def _stereo_discrete(ctx, frames):
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromStereoArrays(G.voice(60, frames + 256), G.voice(61, frames + 256), SR)
    cv = ConvolverNode(ctx)
    cv.EnableTrueStereo = False
    cv.Buffer = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 12000) for c in range(2)], SR)
    s.Connect(cv).Connect(ctx.Destination)
    s.Start()
    return 2


def test_stereo_input_discrete_channels():
    """Two different input channels, each against its own IR channel: two signals, one column each."""
    frames = 128 * 300
    ref, got, st = pair(lambda c: _stereo_discrete(c, frames), frames)
    assert used_coarse(st)
    check(ref, got)


def _sixteen(ctx, frames, sources=3):
    ctx.Destination.SetChannelCount(16)
    for v in range(sources):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), SR)
        cv = ConvolverNode(ctx)
        cv.Buffer = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 10000, seed0=7 + 100 * v) for c in range(16)], SR)
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()
    return 16


def test_sixteen_channel_ir_long_chunk():
    """16 columns per input = four jobs of 4 columns; 40 coarse blocks > 32: the 4-column jobs are split along time."""
    frames = 128 * 2560
    ref, got, st = pair(lambda c: _sixteen(c, frames), frames)
    assert used_coarse(st)
    check(ref, got)


def _not_fused(ctx, frames):
    """conv -> gain -> destination, and a second convolver whose output feeds TWO inputs: nothing to fuse."""
    ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 9000) for c in range(2)], SR)
    for v in range(2):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), SR)
        cv = ConvolverNode(ctx)
        cv.Buffer = ir
        g = GainNode(ctx)
        g.Gain.Value = 0.5 + 0.25 * v
        s.Connect(cv)
        cv.Connect(g)
        g.Connect(ctx.Destination)
        if v == 1:
            g2 = GainNode(ctx)
            g2.Gain.Value = -0.3
            cv.Connect(g2)
            g2.Connect(ctx.Destination)
        s.Start()
    return 2


def test_outputs_that_cannot_be_fused():
    frames = 128 * 300
    ref, got, st = pair(lambda c: _not_fused(c, frames), frames)
    assert used_coarse(st)
    check(ref, got)


def _fused_with_other_terms(ctx, frames):
    """The destination sums a fused group of convolvers AND plain voices AND a convolver of another length (second group)."""
    ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 20000) for c in range(2)], SR)
    ir2 = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 9000, seed0=40) for c in range(2)], SR)
    for v in range(6):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), SR)
        if v == 2:
            g = GainNode(ctx)
            g.Gain.Value = 0.01
            s.Connect(g).Connect(ctx.Destination)
        else:
            cv = ConvolverNode(ctx)
            cv.Buffer = ir2 if v == 4 else ir
            s.Connect(cv).Connect(ctx.Destination)
        s.Start()
    return 2


def test_fused_group_next_to_other_terms():
    frames = 128 * 300
    ref, got, st = pair(lambda c: _fused_with_other_terms(c, frames), frames)
    assert used_coarse(st)
    check(ref, got)


def _mono_ir_into_stereo_destination(ctx, frames):
    """A 1-channel IR: the convolver's output is mono and the destination (2 channels) copies it to both (1 -> N rule)."""
    ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(0, 9000)], SR)
    for v in range(3):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), SR)
        cv = ConvolverNode(ctx)
        cv.Buffer = ir
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()
    return 2


def test_fused_mono_outputs_upmixed_by_the_consumer():
    frames = 128 * 200
    ref, got, st = pair(lambda c: _mono_ir_into_stereo_destination(c, frames), frames)
    assert used_coarse(st)
    check(ref, got)
    assert np.array_equal(got[0], got[1])


def _downmixed(ctx, frames):
    """Stereo convolver outputs into a mono (explicit) input: the N -> 1 rule (sum of channels / sqrt N) applied to the group sum."""
    ctx.Destination.SetChannelCount(1)
    bus = GainNode(ctx)
    bus.Inputs[0].SetChannelCount(1)
    bus.Inputs[0].SetChannelCountMode(ChannelCountMode.Explicit)
    bus.Connect(ctx.Destination)
    ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 9000) for c in range(2)], SR)
    for v in range(3):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), SR)
        cv = ConvolverNode(ctx)
        cv.Buffer = ir
        s.Connect(cv).Connect(bus)
        s.Start()
    return 1


def test_fused_group_downmixed_by_the_consumer():
    frames = 128 * 200
    ref, got, st = pair(lambda c: _downmixed(c, frames), frames)
    assert used_coarse(st)
    check(ref, got)


def _chain(ctx, frames):
    """Two convolvers in series (convolver depth 0 and 1) with a biquad in between."""
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(3, frames + 256), SR)
    a = ConvolverNode(ctx)
    a.Buffer = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(0, 9000)], SR)
    bq = BiQuadFilterNode(ctx)
    bq.Type = FilterType.Highpass
    bq.Frequency.Value = 900.0
    b = ConvolverNode(ctx)
    b.Buffer = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 12000, seed0=90) for c in range(2)], SR)
    s.Connect(a).Connect(bq).Connect(b).Connect(ctx.Destination)
    s.Start()
    return 2


def test_convolvers_in_series():
    frames = 128 * 300
    # (conv_reference_order = 0: this test is about formulation D at two convolver depths; by default the planner evaluates the FIRST
    # convolver in the reference's own order, because the biquad behind it amplifies last-bit differences -- second half of the test)
    ref, got, st = pair(lambda c: _chain(c, frames), frames, [128 * 100, 128 * 200], conv_reference_order=0)
    assert used_coarse(st)
    check(ref, got, rel=2e-5)   # the high-pass recursion amplifies the 4e-7 of the first convolver
    ref, got2, st2 = pair(lambda c: _chain(c, frames), frames, [128 * 100, 128 * 200])
    assert st2["ref_order_rows"] > 0 and st2["stage_launches"][5] > 0   # first convolver: formulation R, second: D
    check(ref, got2)
    assert G.rms(ref - got2) < G.rms(ref - got)


def _scheduled(ctx, frames):
    """Voices that start late, stop early and end inside the chunk: segments; the convolver keeps ringing on silent input."""
    ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 10000) for c in range(2)], SR)
    for v in range(4):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, 128 * (40 + 30 * v)), SR)
        cv = ConvolverNode(ctx)
        cv.Buffer = ir
        s.Connect(cv).Connect(ctx.Destination)
        s.Start(0.01 * v)
        if v == 1:
            s.Stop(0.2)
    return 2


def test_sources_ending_inside_the_chunk():
    frames = 128 * 300
    ref, got, st = pair(lambda c: _scheduled(c, frames), frames)
    assert used_coarse(st) and st["segments"] > 2
    check(ref, got)


def test_unplugged_convolver_keeps_its_state():
    """A convolver disconnected from the destination for a while is not processed (pull model): its input history stays."""
    frames = 128 * 360

    def run(ctx):
        ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 9000) for c in range(2)], SR)
        cvs = []
        for v in range(2):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), SR)
            cv = ConvolverNode(ctx)
            cv.Buffer = ir
            s.Connect(cv).Connect(ctx.Destination)
            s.Start()
            cvs.append(cv)
        out = np.zeros((2, frames), np.float32)
        ctx.Render(out, 128 * 120, 0)
        cvs[1].Disconnect()
        ctx.Render(out, 128 * 100, 128 * 120)
        cvs[1].Connect(ctx.Destination)
        ctx.Render(out, 128 * 140, 128 * 220)
        return out

    ref = run(OracleContext(SR))
    got = run(hip())
    check(ref, got)


def test_ir_swap_resets_the_state():
    frames = 128 * 300

    def run(ctx):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(7, frames + 256), SR)
        cv = ConvolverNode(ctx)
        cv.Buffer = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 9000) for c in range(2)], SR)
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()
        out = np.zeros((2, frames), np.float32)
        ctx.Render(out, 128 * 150, 0)
        cv.Buffer = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 14000, seed0=70) for c in range(2)], SR)
        ctx.Render(out, 128 * 150, 128 * 150)
        return out

    ref = run(OracleContext(SR))
    got = run(hip())
    check(ref, got)


def test_default_policy_long_first_chunk_takes_the_coarse_path():
    frames = 128 * 300
    o = OracleContext(SR)
    G.config3_convolver(o, voices=2, taps=20000, frames=frames)
    ref = G.render(o, 2, frames)
    h = OfflineAudioContext(SR)   # defaults: coarse = 1, coarse_min_blocks = 256
    G.config3_convolver(h, voices=2, taps=20000, frames=frames)
    got = G.render(h, 2, frames)
    assert used_coarse(h.GetStats())
    check(ref, got)
    h2 = OfflineAudioContext(SR)
    G.config3_convolver(h2, voices=2, taps=20000, frames=frames)
    got2 = np.zeros_like(ref)
    h2.Render(got2, 128 * 100, 0)          # a short first chunk: the block-axis FFT formulation, kept for the node's life
    h2.Render(got2, 128 * 200, 128 * 100)
    st = h2.GetStats()
    assert st["stage_launches"][5] == 0 and st["stage_launches"][3] > 0
    check(ref, got2)


# ---- the two multiply-accumulate kernels: terms that share their impulse response are summed before the multiply
#      (coarse_sum_kernel), terms with their own go through coarse_mac_kernel ----
@pytest.mark.parametrize("voices", [1, 4, 5, 33, 70])
def test_shared_ir_term_counts(voices):
    """jobs of 32 terms + a rest; the reduction's four-terms-in-flight loop and its tail."""
    frames = 128 * 300
    ref, got, st = pair(lambda c: G.config3_convolver(c, voices=voices, taps=20000, frames=frames), frames, [128 * 130, 128 * 170])
    assert used_coarse(st)
    check(ref, got)


@pytest.mark.parametrize("ir_channels", [1, 4])
def test_shared_ir_one_and_four_columns(ir_channels):
    """mono IR (1 column) and a 4-channel IR on mono voices (4 columns -> the 4-column instance of both kernels)"""
    frames = 128 * 300
    ref, got, st = pair(lambda c: G.config3_convolver(c, voices=6, taps=17000, frames=frames, ir_channels=ir_channels), frames)
    assert used_coarse(st)
    check(ref, got)


def _shared_and_private(ctx, frames):
    shared = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 25000) for c in range(2)], SR)
    ctx.Destination.SetChannelCount(2)
    for v in range(9):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), SR)
        cv = ConvolverNode(ctx)
        # voices 0-5 share one buffer, 6-8 have their own (same length: same partition count, same fused group)
        cv.Buffer = shared if v < 6 else PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 25000, seed0=900 + 10 * v) for c in range(2)], SR)
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()
    return 2


def test_shared_and_private_impulse_responses_in_one_sum():
    frames = 128 * 300
    ref, got, st = pair(lambda c: _shared_and_private(c, frames), frames, [128 * 100, 128 * 200])
    assert used_coarse(st)
    check(ref, got)


def test_history_carried_by_the_forward_kernel_equals_the_copy_kernel():
    """option coarse_carry: the next chunk's history written by coarse_fwd_kernel or by coarse_hist_kernel -- same bits"""
    frames = 128 * 700
    pieces = [128 * 300, 128 * 50, 128 * 350]   # longer than the history (carried), shorter (copied), longer again
    outs = []
    for carry in (1, 0):
        h = hip(coarse_carry=carry)
        G.config3_convolver(h, voices=3, taps=30000, frames=frames)
        got = np.zeros((2, frames), np.float32)
        pos = 0
        for n in pieces:
            h.Render(got, n, pos)
            pos += n
        st = h.GetStats()
        assert used_coarse(st)
        assert (st["stage_launches"][8] > 0) if carry == 0 else True
        outs.append(got)
        h.Dispose()
    assert np.array_equal(outs[0], outs[1])


# ---- carried output tails (option coarse_tail): a group that stays the same from chunk to chunk renders its first blocks from
#      the tail the previous chunk left, not from the members' input histories; any change falls back to the histories ----
def test_a_convolver_on_its_own_keeps_to_its_input_history():
    """a tail costs P' inverse transforms per output channel and saves P' - 2 forward transforms per input: no gain for one voice"""
    frames = 128 * 600
    ref, got, st = pair(lambda c: G.config3_convolver(c, voices=1, taps=65536, frames=frames), frames, [128 * 300, 128 * 300])
    assert used_coarse(st) and st["coarse_carried_outputs"] == 0
    check(ref, got)


def test_steady_chunks_render_from_carried_tails():
    frames = 128 * 900
    pieces = [128 * 300, 128 * 100, 128 * 37, 128 * 263, 128 * 200]   # longer and shorter than the 4-partition tail
    ref, got, st = pair(lambda c: G.config3_convolver(c, voices=5, taps=30000, frames=frames), frames, pieces)
    assert used_coarse(st)
    assert st["coarse_carried_outputs"] >= 2 * (len(pieces) - 1)   # two output channels, every chunk but the first
    check(ref, got)
    ref2, got2, st2 = pair(lambda c: G.config3_convolver(c, voices=5, taps=30000, frames=frames), frames, pieces, coarse_tail=0)
    assert st2["coarse_carried_outputs"] == 0
    check(ref2, got2)
    assert G.rms(got - got2) <= 4e-7 * G.rms(ref)   # the two routes differ by rounding only


def test_group_changes_between_chunks_fall_back_to_the_histories():
    """a voice joins the fused group, one leaves it, one comes back, one is disposed -- every time the tail of the old group is
    dropped and the members' input histories take over (both kept up to date in every chunk)"""
    def script(ctx, render):
        shared = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 26000) for c in range(2)], SR)
        ctx.Destination.SetChannelCount(2)
        voices = []

        def add(v):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, 128 * 1300), SR)
            cv = ConvolverNode(ctx)
            cv.Buffer = shared
            s.Connect(cv).Connect(ctx.Destination)
            s.Start()
            voices.append((s, cv))
        for v in range(6):
            add(v)
        render(128 * 200)
        render(128 * 150)                       # same group: carried
        add(6)                                  # a voice joins
        render(128 * 150)
        render(128 * 100)                       # carried again
        voices[1][1].Disconnect()               # a member leaves the sum (its convolver stops being pulled)
        render(128 * 120)
        voices[1][1].Connect(ctx.Destination)   # ... and comes back with the state it had when it left
        render(128 * 130)
        voices[0][1].Dispose()                  # a member is disposed
        render(128 * 150)
        render(128 * 100)

    total = 128 * 1100

    def run(ctx):
        out = np.zeros((2, total), np.float32)
        pos = [0]

        def render(n):
            ctx.Render(out, n, pos[0])
            pos[0] += n
        script(ctx, render)
        assert pos[0] == total
        return out

    o = OracleContext(SR)
    ref = run(o)
    h = hip()
    got = run(h)
    st = h.GetStats()
    assert used_coarse(st) and st["coarse_carried_outputs"] >= 4
    check(ref, got)
    h.Dispose()
    o.Dispose()


@pytest.mark.parametrize("taps", [8193, 12000, 20000, 32768, 40000])
@pytest.mark.parametrize("mfma", [1, 0])
def test_sixteen_column_jobs_at_every_partition_count(taps, mfma):
    """16-channel private impulse responses (config 5's shape) with P' = 2, 2, 3, 4 partitions on the matrix-core kernel (rows of
    partitions that do not exist are zero rows of its LDS image, never loaded) and P' = 5 on the register-tiled 16-column kernel;
    rendered in uneven pieces so that jobs start with and without windows in front; `coarse_mfma = 0`: the register-tiled kernel
    for all of them."""
    frames = 128 * 520
    ref, got, st = pair(lambda c: G.config5_ambisonic(c, sources=3, taps=taps, frames=frames), frames, pieces=[128 * 70, 128 * 333, 128 * 117],
                        coarse_mfma=mfma)
    assert used_coarse(st)
    kernels = " ".join(st["stage_kernel"])
    assert ("mfma16" in kernels) == (mfma == 1 and taps <= 32768), kernels
    check(ref, got)


# ---- exact zeros in front of an onset (the round-3 open defect: edit session 42867) ----------------------------------------
def _onset_scene(ctx, onset_block, taps=9000, frames=128 * 40, mono_mod=False, through_delay=False):
    """A mono voice -> StereoPannerNode(pan 0) -> destination, the pan modulated (through a depth gain) by a true-stereo
    convolver whose stereo source starts at `onset_block`.  Until then the reference's convolver puts out exact zeros, so
    `pan != _lastPan` (StereoPannerNode.cs:92-99) never fires and the panner keeps the gains of its FIRST block (stereo law:
    the first block's input is the up-mixed 2-channel buffer, AudioNodeInput.cs:140-168); any rounding noise in front of the
    onset re-derives them with the mono law and the voice comes out ~1e7 times louder on the left."""
    from graphaudio_amd import StereoPannerNode, DelayNode
    ctx.Destination.SetChannelCount(2)
    v = AudioBufferSourceNode(ctx)
    v.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(3, frames + 256), SR)
    pn = StereoPannerNode(ctx)
    pn.Pan.Value = 0.0
    v.Connect(pn).Connect(ctx.Destination)
    v.Start()
    m = AudioBufferSourceNode(ctx)
    data = [G.voice(11, frames), G.voice(12, frames)]
    m.Buffer = PlayableAudioBuffer.FromChannelArrays(data[:1] if mono_mod else data, SR)
    cv = ConvolverNode(ctx)
    cv.Buffer = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, taps) for c in range(4)], SR)
    depth = GainNode(ctx)
    depth.Gain.Value = 0.48
    m.Connect(cv).Connect(depth)
    if through_delay:
        d = DelayNode(ctx, 0.05)
        d.DelayTime.Value = 0.001
        depth.Connect(d)
        d.Connect(pn.Pan)
    else:
        depth.Connect(pn.Pan)
    m.Start(onset_block * 128 / SR + 1e-4)   # (block-granular start: the first block whose end lies behind this time)
    return cv


@pytest.mark.parametrize("mono_mod", [False, True])
@pytest.mark.parametrize("pieces", [[128 * 10, 128 * 6, 128 * 24], [128 * 40], [128 * 3 + 77, 128 * 9, 128 * 5 + 51, 128 * 30]])
def test_exact_zeros_in_front_of_an_onset_inside_the_chunk(pieces, mono_mod):
    """Session 42867 as a deterministic case: the convolver's input is silent, then its two channels differ from the middle
    of a chunk on (shared -> per-channel rows of formulation D in the same chunk)."""
    frames = 128 * 40
    # conv_reference_order = 0: since round 4 the planner evaluates a convolver that feeds a PARAMETER in the reference's own order
    # (formulation R, exact anyway); this test is about the transform formulations' zeros, so the route is switched off.
    # D forced in short chunks / D in one chunk / formulation C / and the default (R)
    for opts in ({"max_chunk_blocks": 11, "conv_reference_order": 0}, {"conv_reference_order": 0},
                 {"coarse_min_blocks": 1 << 30, "conv_reference_order": 0}, {}):
        ref, got, st = pair(lambda c: (_onset_scene(c, 15, mono_mod=mono_mod), 2)[1], frames, pieces, **opts)
        assert (st["ref_order_rows"] > 0) == (not opts)
        assert G.rms(ref - got) <= 1e-6, (opts, G.rms(ref - got), G.rms(ref))


def test_exact_zeros_in_front_of_an_onset_after_an_impulse_response_swap():
    """The same with the impulse response swapped before the onset: the new PartitionedConvolver instances start from an
    empty delay line (ConvolverNode.cs:51-77), so the zeros stay exact -- and a swap AFTER the onset does too."""
    frames = 128 * 40
    for swap_at, onset in ((128 * 8, 15), (128 * 20, 5)):
        outs = []
        for ctx in (OracleContext(SR), hip(max_chunk_blocks=11, conv_reference_order=0)):
            cv = _onset_scene(ctx, onset)
            out = np.zeros((2, frames), np.float32)
            ctx.Render(out, swap_at, 0)
            cv.Buffer = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 12000, seed0=40) for c in range(4)], SR)
            ctx.Render(out, frames - swap_at, swap_at)
            outs.append(out)
            ctx.Dispose()
        assert G.rms(outs[0]) > 1e-3
        assert G.rms(outs[0] - outs[1]) <= 1e-6, (swap_at, G.rms(outs[0] - outs[1]))


def test_default_policy_onset_in_the_middle_of_a_long_chunk():
    """The default policy (no option set): one render of 300 blocks takes formulation D, the onset sits in block 15."""
    frames = 128 * 300

    def scene(c):
        _onset_scene(c, 15, frames=frames)
        return 2
    o = OracleContext(SR)
    scene(o)
    ref = G.render(o, 2, frames)
    for ref_order in (0, 1):   # formulation D (the route of round 3, where the defect lived) / the planner's choice now: R
        h = OfflineAudioContext(SR)
        h.SetOption("conv_reference_order", ref_order)
        scene(h)
        got = G.render(h, 2, frames)
        st = h.GetStats()
        assert (st["stage_launches"][5] > 0) == (ref_order == 0) and (st["ref_order_rows"] > 0) == (ref_order == 1)
        assert G.rms(ref - got) <= 1e-6, (ref_order, G.rms(ref - got))
