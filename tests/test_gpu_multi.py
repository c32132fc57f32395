"""ga_render_reduce with a LIVE communicator: two ranks as two threads of this process, one context per GPU, RCCL between them.
Needs two MI355X in one process -- skipped on the one-GPU boxes of this build (the round's driver runs the N = 2, 4, 8 scaling
bench on a whole node: bench.py --gpus N goes through exactly this path, and checks its sum against float64 mathematics)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import InvalidOperationException, OfflineAudioContext
from graphaudio_amd._capi import product_api
from graphaudio_amd.distributed import shard_range
from tests import _f64model as M
from tests import _graphs as G

SR = 48000


def _two_gpus():
    if product_api().device_count() < 2:
        pytest.skip("needs two GPUs in one process")


def _rank(rank, world, uid, frames, voices, results, errors, steps=3, async_=True):
    try:
        ctx = OfflineAudioContext(SR, device=rank)
        v0, v1 = shard_range(voices, world, rank)
        irbuf = None
        from graphaudio_amd import AudioBufferSourceNode, ConvolverNode, PlayableAudioBuffer
        irbuf = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 30000) for c in range(2)], SR)
        ctx.Destination.SetChannelCount(2)
        for v in range(v0, v1):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames * steps + 256), SR)
            cv = ConvolverNode(ctx)
            cv.Buffer = irbuf
            s.Connect(cv).Connect(ctx.Destination)
            s.Start()
        ctx.CommInit(uid, world, rank)
        info = ctx.CommInfo()   # ncclCommCount / ncclCommUserRank: what the SCALE record's `comm` field is read from
        assert info == {"ranks": world, "rank": rank, "uses_rccl": True}, info
        if async_:
            ctx.SetOption("async", 1)
        outs = []
        for k in range(steps):
            out = np.zeros((2, frames), np.float32)
            ctx.RenderReduce(out, frames)
            outs.append(out)
        ctx.Synchronize()
        results[rank] = outs
        ctx.CommDestroy()
        ctx.Dispose()
    except Exception as e:   # noqa: BLE001 -- reported by the test thread
        errors[rank] = e


def test_two_ranks_reduce_equals_the_unsharded_render_and_float64():
    _two_gpus()
    frames, voices, steps = 128 * 300, 24, 3
    boot = OfflineAudioContext(SR, device=0)
    uid = boot.CommUniqueId()
    boot.Dispose()
    results, errors = {}, {}
    ts = [threading.Thread(target=_rank, args=(r, 2, uid, frames, voices, results, errors, steps)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(300)
    assert not errors, errors
    got = np.concatenate(results[0], axis=1)   # rank 0 is the root
    xs = np.zeros(frames * steps)
    for v in range(voices):
        xs += G.voice(v, frames * steps + 256)[:frames * steps]
    truth = np.stack([M.linear_conv(xs, M.scaled_ir64(G.synth_ir(c, 30000)), frames * steps) for c in range(2)])
    err, sig = M.rms(got - truth), M.rms(truth)
    assert err <= 1e-5 and err / sig < 2e-6, (err, sig)


def test_a_rank_that_fails_does_not_leave_its_peer_waiting():
    """rank 1 renders into too few frames' worth of rows (an argument error inside ga_render_reduce's render): it aborts the
    communicator; rank 0, whose collective is already enqueued, gets an error code from ga_synchronize instead of a hang"""
    _two_gpus()
    boot = OfflineAudioContext(SR, device=0)
    uid = boot.CommUniqueId()
    boot.Dispose()
    outcome = {}

    def rank(r):
        ctx = OfflineAudioContext(SR, device=r)
        G.config3_convolver(ctx, voices=4, taps=20000, frames=128 * 64)
        ctx.CommInit(uid, 2, r)
        ctx.SetOption("comm_timeout_s", 20)
        ctx.SetOption("async", 1)
        try:
            if r == 1:
                ctx.Destination.SetChannelCount(1)          # the root asks for 2 channels: "channelIndex" inside the render
            ctx.RenderReduce(np.zeros((2, 128 * 64), np.float32), 128 * 64)
            ctx.Synchronize()
            outcome[r] = "ok"
        except Exception as e:   # noqa: BLE001
            outcome[r] = type(e).__name__
        ctx.CommDestroy()
        ctx.Dispose()

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    assert all(not t.is_alive() for t in ts), "a rank is still waiting"
    assert outcome[1] != "ok" and outcome[0] != "ok", outcome
