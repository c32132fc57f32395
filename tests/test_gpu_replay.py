"""Steady renders skip the control-plane traversal of a chunk's first block (Context::lastSegNodes, option sim_replay): the records
of the previous chunk's last segment are taken over when nothing can have moved -- no API call, no queued command, the same graph,
no time-dependent node, every source in the same phase.  The render must be bit-identical to the one that traverses every chunk,
and anything that CAN move control state has to switch the shortcut off for that chunk."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import (AudioBufferSourceNode, BiQuadFilterNode, ConvolverNode, DelayNode, FilterType, GainNode,
                            OfflineAudioContext, PlayableAudioBuffer, StereoPannerNode)
from tests import _graphs as G
from tests._oracle import OracleContext

SR = 48000


def _render(build, frames, piece, replay, edit=None, **opts):
    ctx = OfflineAudioContext(SR)
    ctx.SetOption("sim_replay", replay)
    for k, v in opts.items():
        ctx.SetOption(k, v)
    handles = build(ctx)
    ch = handles[0] if isinstance(handles, tuple) else handles
    out = np.zeros((ch, frames), np.float32)
    pos, k = 0, 0
    while pos < frames:
        n = min(piece, frames - pos)
        ctx.Render(out, n, pos)
        pos += n
        k += 1
        if edit:
            edit(ctx, handles, k)
    st = ctx.GetStats()
    ctx.Dispose()
    return out, st


def test_config4_steady_chunks_are_replayed_bit_identically():
    frames, piece = 128 * 160, 128 * 20
    build = lambda c: G.config4_eq(c, voices=48, frames=frames)
    a, sa = _render(build, frames, piece, 1)
    b, sb = _render(build, frames, piece, 0)
    assert sb["sim_replays"] == 0
    assert sa["sim_replays"] >= sa["chunks"] - 3   # all but the first chunks (the second one confirms the fixpoint)
    assert G.rms(a) > 1e-4 and np.array_equal(a, b)
    o = OracleContext(SR)
    build(o)
    assert np.array_equal(G.render(o, 2, frames), a)   # config 4 is bit-exact on the device (one-walk cascades)


def test_headline_shape_and_kit_scene_replayed():
    frames, piece = 128 * 600, 128 * 150
    for build in (lambda c: G.config3_convolver(c, voices=24, taps=20000, frames=frames),
                  lambda c: G.kit_scene(c, voices=16, frames=frames + 256, taps=20000)):
        a, sa = _render(build, frames, piece, 1, coarse_min_blocks=1)
        b, sb = _render(build, frames, piece, 0, coarse_min_blocks=1)
        assert sa["sim_replays"] >= 1 and sb["sim_replays"] == 0
        assert np.array_equal(a, b)


def test_anything_that_can_move_control_state_switches_the_shortcut_off():
    frames, piece = 128 * 120, 128 * 10

    def build(ctx):
        ctx.Destination.SetChannelCount(2)
        srcs, gains = [], []
        for v in range(6):
            s = AudioBufferSourceNode(ctx)
            # (voices of different lengths: some END in the middle of the render -- a phase change at a chunk's first block or inside it)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, 128 * (35 + 17 * v) + 40 * v), SR)
            bq = BiQuadFilterNode(ctx)
            bq.Type = FilterType.Peaking
            bq.Frequency.Value = 500.0 + 300 * v
            g = GainNode(ctx)
            g.Gain.Value = 0.5
            s.Connect(bq).Connect(g).Connect(ctx.Destination)
            s.Start(0.0 if v % 2 else 0.013 * v)
            srcs.append(s)
            gains.append(g)
        return 2, srcs, gains

    def edit(ctx, handles, k):
        _, srcs, gains = handles
        if k == 3:
            gains[1].Gain.Value = 0.25                      # a parameter write
        if k == 5:
            gains[2].Gain.LinearRampToValueAtTime(0.1, ctx.CurrentTime + 0.05)
        if k == 7:
            srcs[5].Stop(ctx.CurrentTime + 0.004)           # a stop that takes effect inside the next chunk
        if k == 8:
            gains[3].Disconnect()                           # a graph edit
        if k == 9:
            gains[3].Connect(ctx.Destination)

    a, sa = _render(build, frames, piece, 1, edit)
    b, sb = _render(build, frames, piece, 0, edit)
    assert np.array_equal(a, b)
    assert 0 < sa["sim_replays"] < sa["chunks"] - 5
    o = OracleContext(SR)
    h = build(o)
    ref = np.zeros((2, frames), np.float32)
    pos, k = 0, 0
    while pos < frames:
        o.Render(ref, piece, pos)
        pos += piece
        k += 1
        edit(o, h, k)
    assert np.array_equal(ref, a)


def test_delay_nodes_are_replayed_only_once_their_output_flag_is_up():
    """A DelayNode's control state is its sticky output flag (DelayNode.cs:96-97): until delayed audio has arrived its ring model is
    walked block by block; afterwards (constant delay time) it is a node like any other.  A delay time on a timeline never is."""
    frames, piece = 128 * 80, 128 * 10

    def build(automated):
        def b(ctx):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(1, frames + 256), SR)
            d = DelayNode(ctx, 0.5)
            d.DelayTime.Value = 0.03       # 1440 samples: audible from block 11 on
            if automated:
                d.DelayTime.LinearRampToValueAtTime(0.01, frames / SR)
            s.Connect(d).Connect(ctx.Destination)
            s.Start()
            return 2
        return b
    a, sa = _render(build(False), frames, piece, 1)
    b, sb = _render(build(False), frames, piece, 0)
    assert np.array_equal(a, b) and G.rms(a) > 1e-3
    assert 0 < sa["sim_replays"] < sa["chunks"] - 1, (sa["sim_replays"], sa["chunks"])   # (not the chunks in which the delayed audio arrives)
    c, sc = _render(build(True), frames, piece, 1)
    d, sd = _render(build(True), frames, piece, 0)
    assert sc["sim_replays"] == 0 and np.array_equal(c, d)
    o = OracleContext(SR)
    build(True)(o)
    assert np.array_equal(G.render(o, 2, frames), c)
