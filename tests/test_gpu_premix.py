"""Time-domain pre-mix of fused groups (option coarse_premix, graphaudio_amd/csrc/ga_coarse.hip::coarse_premix_kernel).

Convolvers that hold the same impulse response and feed the same sum are a single convolution of the SUM of their inputs:
sum_v (x_v * h) = (sum_v x_v) * h.  The planner (ga_chunk.cpp, CoarseStage) adds the members' inputs up in the time domain and
transforms one signal per input channel of the group; members keep their own input histories, so groups can re-form.  These tests
cover the shapes the pre-mix has to get right, each against the CPU oracle and against the per-voice route (coarse_premix = 0):
member counts around the kernel's wave split and batches, stereo and true-stereo members, members that are silent for part of
the chunk, inputs that are not 16-byte aligned, chunks shorter than the history, groups that are not uniform (no pre-mix).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import AudioBufferSourceNode, ConvolverNode, GainNode, OfflineAudioContext, PlayableAudioBuffer
from tests import _graphs as G
from tests._oracle import OracleContext

SR = 48000


def hip(**opts):
    ctx = OfflineAudioContext(SR)
    ctx.SetOption("coarse_min_blocks", 1)
    for k, v in opts.items():
        ctx.SetOption(k, v)
    return ctx


def run(ctx, builder, frames, pieces):
    ch = builder(ctx)
    out = np.zeros((ch, frames), np.float32)
    pos = 0
    for n in pieces:
        n = min(n, frames - pos)
        if n > 0:
            ctx.Render(out, n, pos)
            pos += n
    if pos < frames:
        ctx.Render(out, frames - pos, pos)
    return out


def three_way(builder, frames, pieces, expect_premix=True, rel=2e-6, **opts):
    o = OracleContext(SR)
    ref = run(o, builder, frames, [frames])
    o.Dispose()
    outs = []
    for premix in (1, 0):
        h = hip(coarse_premix=premix, **opts)
        outs.append(run(h, builder, frames, pieces))
        st = h.GetStats()
        h.Dispose()
        assert st["stage_launches"][5] > 0
        assert (st["coarse_premixed_signals"] > 0) == bool(premix and expect_premix), st["coarse_premixed_signals"]
        assert (st["stage_launches"][10] > 0) == bool(premix and expect_premix)
    sig = G.rms(ref)
    assert sig > 1e-5
    for got in outs:
        err = G.rms(ref - got)
        assert err <= 1e-5 and err <= rel * sig, (err, sig)
    assert G.rms(outs[0] - outs[1]) <= 1e-6 * sig   # the two routes differ by rounding only
    return ref, outs


@pytest.mark.parametrize("voices", [2, 3, 7, 9, 31, 33, 70, 300])
def test_member_counts(voices):
    """terms per wave (a quarter, rounded up to the batch of 8), the batch loop's rest, descriptor reloads every 64 terms"""
    frames = 128 * 300
    three_way(lambda c: G.config3_convolver(c, voices=voices, taps=20000, frames=frames), frames, [128 * 130, 128 * 170])


def _stereo_members(ctx, frames, true_stereo):
    ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 17000) for c in range(4 if true_stereo else 2)], SR)
    ctx.Destination.SetChannelCount(2)
    for v in range(5):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromStereoArrays(G.voice(2 * v, frames + 256), G.voice(2 * v + 1, frames + 256), SR)
        cv = ConvolverNode(ctx)
        cv.EnableTrueStereo = true_stereo
        cv.Buffer = ir
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()
    return 2


@pytest.mark.parametrize("true_stereo", [False, True])
def test_stereo_members_one_mixed_signal_per_input_channel(true_stereo):
    frames = 128 * 400
    three_way(lambda c: _stereo_members(c, frames, true_stereo), frames, [128 * 150, 128 * 90, 128 * 160])


def _late_and_early(ctx, frames):
    """members that are silent for part of the render: sources that start late (their chunk input is materialised from
    segment views) and one-shots that end early"""
    ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 30000) for c in range(2)], SR)
    ctx.Destination.SetChannelCount(2)
    for v in range(6):
        s = AudioBufferSourceNode(ctx)
        n = frames + 256 if v % 2 == 0 else 128 * (60 + 37 * v)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, n), SR)
        cv = ConvolverNode(ctx)
        cv.Buffer = ir
        s.Connect(cv).Connect(ctx.Destination)
        s.Start(0.0 if v < 3 else (128 * 41 * v + 5) / SR)
    return 2


def test_members_silent_for_part_of_the_render():
    frames = 128 * 500
    three_way(lambda c: _late_and_early(c, frames), frames, [128 * 200, 128 * 30, 128 * 270])


def _unaligned(ctx, frames):
    """playback offsets that are not multiples of four samples: the convolver inputs alias the buffers at unaligned addresses"""
    ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 20000) for c in range(2)], SR)
    ctx.Destination.SetChannelCount(2)
    for v in range(6):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 1024), SR)
        cv = ConvolverNode(ctx)
        cv.Buffer = ir
        s.Connect(cv).Connect(ctx.Destination)
        s.Start(0.0, (v * 37 + 1) / SR)
    return 2


def test_inputs_that_are_not_16_byte_aligned():
    frames = 128 * 300
    three_way(lambda c: _unaligned(c, frames), frames, [128 * 100, 128 * 200])


def test_chunks_shorter_than_the_history():
    """short pieces: the members' histories are copied by coarse_hist_kernel, the group renders from carried tails or from the
    mixed histories"""
    frames = 128 * 420
    pieces = [128 * 7, 128 * 200, 128 * 3, 128 * 11, 128 * 150, 128 * 49]
    three_way(lambda c: G.config3_convolver(c, voices=6, taps=40000, frames=frames), frames, pieces)
    for tail in (1, 0):   # ... and without carried tails: every chunk pre-mixes the histories as well
        h = hip(coarse_tail=tail)
        got = run(h, lambda c: G.config3_convolver(c, voices=6, taps=40000, frames=frames), frames, pieces)
        st = h.GetStats()
        h.Dispose()
        assert st["coarse_premixed_signals"] > 0 and (st["coarse_carried_outputs"] > 0) == bool(tail)
        o = OracleContext(SR)
        ref = run(o, lambda c: G.config3_convolver(c, voices=6, taps=40000, frames=frames), frames, [frames])
        o.Dispose()
        assert G.rms(ref - got) <= 2e-6 * G.rms(ref)


def test_history_written_by_the_premix_equals_the_copy_kernel():
    frames = 128 * 700
    pieces = [128 * 300, 128 * 50, 128 * 350]
    outs = []
    for carry in (1, 0):
        h = hip(coarse_carry=carry, coarse_tail=0)   # (without tails every chunk reads the histories back)
        outs.append(run(h, lambda c: G.config3_convolver(c, voices=5, taps=30000, frames=frames), frames, pieces))
        assert h.GetStats()["coarse_premixed_signals"] > 0
        h.Dispose()
    assert np.array_equal(outs[0], outs[1])


def _mixed_layouts(ctx, frames):
    """a mono and a stereo source through the same impulse response into the same sum: one fused group, not uniform"""
    ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 17000) for c in range(2)], SR)
    ctx.Destination.SetChannelCount(2)
    for v in range(4):
        s = AudioBufferSourceNode(ctx)
        if v == 2:
            s.Buffer = PlayableAudioBuffer.FromStereoArrays(G.voice(20, frames + 256), G.voice(21, frames + 256), SR)
        else:
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), SR)
        cv = ConvolverNode(ctx)
        cv.EnableTrueStereo = False
        cv.Buffer = ir
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()
    return 2


def test_groups_that_are_not_uniform_are_not_premixed():
    frames = 128 * 300
    three_way(lambda c: _mixed_layouts(c, frames), frames, [128 * 100, 128 * 200], expect_premix=False)


def test_group_behind_gains_and_into_a_bus():
    """the members' inputs are node outputs (slabs), the consumer is a bus gain, not the destination"""
    frames = 128 * 300

    def build(ctx):
        ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 25000) for c in range(2)], SR)
        ctx.Destination.SetChannelCount(2)
        bus = GainNode(ctx)
        bus.Gain.Value = 0.5
        bus.Connect(ctx.Destination)
        for v in range(10):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), SR)
            g = GainNode(ctx)
            g.Gain.Value = 0.25 + 0.1 * v   # (none of them 1: a unity gain hands the source's buffer view on, and a group whose
                                            #  members mix buffer views and slabs is not pre-mixed -- covered below)
            cv = ConvolverNode(ctx)
            cv.Buffer = ir
            s.Connect(g).Connect(cv).Connect(bus)
            s.Start()
        return 2

    three_way(build, frames, [128 * 100, 128 * 200])


def test_members_behind_unity_and_other_gains_are_summed_as_spectra():
    """One member's gain is exactly 1: both channels of its convolver's input are the source's buffer view -- ONE signal to transform
    -- while the others' inputs are products.  With every channel's product written on its own (twin_channels = 0) those members
    have two signals each: the group is not uniform and is not pre-mixed, every member is transformed and the spectra are summed.
    By default a product that is the same on both channels is written once (tests/test_gpu_twin.py): every member has one signal,
    the group is pre-mixed.  Same result either way."""
    frames = 128 * 300

    def build(ctx):
        ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 25000) for c in range(2)], SR)
        ctx.Destination.SetChannelCount(2)
        for v in range(6):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), SR)
            g = GainNode(ctx)
            g.Gain.Value = 1.0 if v == 2 else 0.4 + 0.1 * v
            cv = ConvolverNode(ctx)
            cv.Buffer = ir
            s.Connect(g).Connect(cv).Connect(ctx.Destination)
            s.Start()
        return 2

    three_way(build, frames, [128 * 100, 128 * 200], expect_premix=False, twin_channels=0)
    three_way(build, frames, [128 * 100, 128 * 200], expect_premix=True)
