"""Feedback cycles.  The reference does not refuse them: ProcessInternal's memo check (Nodes/AudioNode.cs:153-156) returns before
the "cycle detected" test (:157-160) can fire, so a node that is pulled while it is being processed hands out the buffer its output
STILL holds -- its previous block.  A loop therefore renders with an implicit one-block delay on the edge that closes it (which
edge that is follows from the traversal order: parameters first, then inputs, connections in order).  The oracle restates that;
since round 4 the device path renders such graphs too, one block per chunk (the reference's own granularity), with the stale
block of every such producer kept on the device -- and has to match bit for bit where the node arithmetic is bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import (AudioBufferSourceNode, BiQuadFilterNode, ChannelMergerNode, ChannelSplitterNode, ConvolverNode, DelayNode,
                            FilterType, GainNode, OfflineAudioContext, PlayableAudioBuffer, StereoPannerNode)
from tests import _graphs as G
from tests._oracle import OracleContext

SR = 48000


def both(build, frames, pieces=None, edit=None, channels=2):
    outs = []
    for ctx in (OracleContext(SR), OfflineAudioContext(SR)):
        h = build(ctx)
        out = np.zeros((channels, frames), np.float32)
        pos, k = 0, 0
        for n in (pieces or [frames]):
            n = min(n, frames - pos)
            if n <= 0:
                break
            ctx.Render(out, n, pos)
            pos += n
            k += 1
            if edit:
                edit(ctx, h, k)
        if pos < frames:
            ctx.Render(out, frames - pos, pos)
        outs.append(out)
        ctx.Dispose()
    return outs


def src(ctx, seed, frames, stereo=False):
    s = AudioBufferSourceNode(ctx)
    data = [G.voice(seed, frames), G.voice(seed + 50, frames)]
    s.Buffer = PlayableAudioBuffer.FromChannelArrays(data if stereo else data[:1], SR)
    s.Start()
    return s


def test_delay_gain_delay_echo_is_bit_exact():
    """VERDICT r3 item 7's bar: the classic echo, DelayNode -> GainNode -> DelayNode (Nodes/DelayNode.cs:43-100)."""
    frames = 128 * 120

    def build(ctx):
        s = src(ctx, 1, 128 * 20)   # a short burst, then the echoes ring on
        d = DelayNode(ctx, 0.05)
        d.DelayTime.Value = 0.0123
        fb = GainNode(ctx)
        fb.Gain.Value = 0.6
        s.Connect(d)
        d.Connect(fb).Connect(d)
        d.Connect(ctx.Destination)
    ref, got = both(build, frames, pieces=[128 * 7 + 5, 128 * 40, 128 * 3 + 100])
    assert G.rms(ref[:, 128 * 60:]) > 1e-4   # (echoes long after the source ended)
    assert np.array_equal(ref, got)


def test_two_gains_feeding_each_other_one_block_feedback():
    frames = 128 * 40

    def build(ctx):
        s = src(ctx, 2, frames, stereo=True)
        g1, g2 = GainNode(ctx), GainNode(ctx)
        g1.Gain.Value = 0.5
        g2.Gain.Value = 0.9
        s.Connect(g1)
        g1.Connect(g2)
        g2.Connect(g1)
        g2.Connect(ctx.Destination)
    ref, got = both(build, frames)
    assert G.rms(ref) > 1e-3 and np.array_equal(ref, got)


def test_cycle_through_a_parameter_and_through_splitter_merger_panner():
    frames = 128 * 30

    def build(ctx):
        s = src(ctx, 3, frames)
        g = GainNode(ctx)
        g.Gain.Value = 0.7
        depth = GainNode(ctx)
        depth.Gain.Value = 0.3
        s.Connect(g)
        g.Connect(depth)
        depth.Connect(g.Gain)                  # the gain's own output modulates its gain (one block late)
        pn = StereoPannerNode(ctx)
        pn.Pan.Value = 0.25
        sp, mg = ChannelSplitterNode(ctx, 2), ChannelMergerNode(ctx, 2)
        g.Connect(pn).Connect(sp)
        sp.Connect(mg, 0, 1)
        sp.Connect(mg, 1, 0)
        loop = GainNode(ctx)
        loop.Gain.Value = 0.4
        mg.Connect(loop).Connect(pn)           # panner -> splitter -> merger -> gain -> panner
        mg.Connect(ctx.Destination)
    ref, got = both(build, frames, pieces=[128 * 11, 128 * 2 + 3])
    assert G.rms(ref) > 1e-3 and np.array_equal(ref, got)


def test_cycle_with_a_biquad_and_a_convolver():
    frames = 128 * 60

    def build(ctx):
        s = src(ctx, 4, 128 * 25)
        bq = BiQuadFilterNode(ctx)
        bq.Type = FilterType.Lowpass
        bq.Frequency.Value = 3000.0
        cv = ConvolverNode(ctx)
        cv.Buffer = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(0, 500)], SR)
        fb = GainNode(ctx)
        fb.Gain.Value = 0.5
        s.Connect(bq).Connect(cv).Connect(fb).Connect(bq)
        cv.Connect(ctx.Destination)
    ref, got = both(build, frames, pieces=[128 * 9, 128 * 30])
    assert G.rms(ref) > 1e-5
    assert G.rms(ref - got) <= 1e-6 * max(G.rms(ref), 1e-3), G.rms(ref - got)   # (the convolver's partition sum is not bit-exact)


def test_an_edit_closes_a_loop_and_opens_it_again():
    """The first block after the edit mixes what the new stale producer put out BEFORE the edit (its output buffer still holds it)."""
    frames = 128 * 50

    def build(ctx):
        s = src(ctx, 5, frames)
        a, b = GainNode(ctx), GainNode(ctx)
        a.Gain.Value = 0.8
        b.Gain.Value = 0.5
        s.Connect(a).Connect(b).Connect(ctx.Destination)
        return a, b

    def edit(ctx, h, k):
        a, b = h
        if k == 2:
            b.Connect(a)        # closes the loop a -> b -> a
        if k == 4:
            b.Disconnect(a)     # and opens it again
    ref, got = both(build, frames, pieces=[128 * 6, 128 * 7 + 9, 128 * 5, 128 * 3 - 9, 128 * 20], edit=edit)
    assert G.rms(ref) > 1e-3 and np.array_equal(ref, got)


def test_master_echo_behind_sixteen_voices():
    frames = 128 * 200

    def build(ctx):
        bus = GainNode(ctx)
        bus.Gain.Value = 0.5
        for v in range(16):
            s = src(ctx, 10 + v, frames)
            bq = BiQuadFilterNode(ctx)
            bq.Type = FilterType.Peaking
            bq.Frequency.Value = 400.0 + 150 * v
            bq.Gain.Value = 3.0
            s.Connect(bq).Connect(bus)
        d = DelayNode(ctx, 0.5)
        d.DelayTime.Value = 0.11
        fb = GainNode(ctx)
        fb.Gain.Value = 0.45
        bus.Connect(d)
        d.Connect(fb).Connect(d)
        bus.Connect(ctx.Destination)
        d.Connect(ctx.Destination)
    ref, got = both(build, frames)
    assert G.rms(ref) > 1e-3 and np.array_equal(ref, got)
