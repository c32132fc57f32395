"""AudioContextBase.ProcessBlocks / ProcessBlockInterleaved (AudioContextBase.cs:88-186): SURVEY.md 8(f) rank 2.
CPU: the oracle against Render; GPU: the HIP library against the oracle."""
import numpy as np
import pytest

from graphaudio_amd import (ArgumentException, ArgumentOutOfRangeException, AudioBufferSourceNode, ConvolverNode, GainNode,
                            PlayableAudioBuffer)
from tests import _graphs as G
from tests._oracle import OracleContext

SR = 48000


def _graph(ctx):
    """A stereo voice that starts late (the destination buffer has 2 channels all along), a convolver and a mono voice."""
    rng = np.random.default_rng(8)
    ctx.Destination.SetChannelCount(2)
    s = AudioBufferSourceNode(ctx)
    s.Buffer = PlayableAudioBuffer.FromStereoArrays((rng.standard_normal(128 * 30) * 0.25).astype(np.float32),
                                                   (rng.standard_normal(128 * 30) * 0.25).astype(np.float32), SR)
    c = ConvolverNode(ctx)
    c.Buffer = PlayableAudioBuffer.FromStereoArrays((rng.standard_normal(300) * 0.1).astype(np.float32),
                                                   (rng.standard_normal(300) * 0.1).astype(np.float32), SR)
    s.Connect(c)
    c.Connect(ctx.Destination)
    m = AudioBufferSourceNode(ctx)
    m.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(128 * 12) * 0.25).astype(np.float32), SR)
    g = GainNode(ctx)
    g.Gain.Value = 0.5
    m.Connect(g)
    g.Connect(ctx.Destination)
    s.Start(0.004)
    m.Start(0.0)
    return (s, c, m, g)


def _run(mk, how, nblocks=24, channels=2):
    ctx = mk(SR)
    hold = _graph(ctx)
    if how == "render":
        out = np.zeros((2, nblocks * 128), np.float32)
        ctx.Render(out, nblocks * 128)
        res = out
    elif how == "blocks":
        outs = [np.full(nblocks * 128, 7.0, np.float32) for _ in range(channels)]
        skip1 = (lambda lst: lst[:1] + [None] + lst[2:]) if channels > 2 else (lambda lst: lst)   # a null channel entry is skipped
        ctx.ProcessBlocks(skip1(outs), 10)                                                  # first 10 blocks ...
        ctx.ProcessBlocks(skip1([o[10 * 128:] for o in outs]), nblocks - 10)                # ... then the remaining ones
        res = np.stack(outs)
    else:
        buf = np.full(nblocks * 128 * channels, 7.0, np.float32)
        ctx.ProcessBlockInterleaved(buf, channels)                                          # the reference call: one block
        ctx.ProcessBlocksInterleaved(buf[128 * channels:], channels, nblocks - 1)
        res = buf.reshape(nblocks * 128, channels).T.copy()
    del hold
    ctx.Dispose()
    return res


def test_oracle_process_blocks_and_interleaved_match_render():
    ref = _run(OracleContext, "render")
    assert G.rms(ref) > 1e-3
    planar = _run(OracleContext, "blocks")
    assert np.array_equal(planar, ref)
    inter = _run(OracleContext, "interleaved")
    assert np.array_equal(inter, ref)
    # more output channels than the destination buffer has: planar leaves them untouched (:173), interleaved zero-fills (:142-153)
    planar4 = _run(OracleContext, "blocks", channels=4)
    assert np.array_equal(planar4[0], ref[0]) and np.all(planar4[1] == 7.0) and np.all(planar4[2:] == 7.0)
    inter4 = _run(OracleContext, "interleaved", channels=4)
    assert np.array_equal(inter4[:2], ref) and np.abs(inter4[2:]).max() == 0.0


def test_oracle_process_blocks_argument_checks():
    ctx = OracleContext(SR)
    with pytest.raises(ArgumentOutOfRangeException):
        ctx.ProcessBlocks([np.zeros(128, np.float32)], -1)
    with pytest.raises(ArgumentOutOfRangeException):
        ctx.ProcessBlockInterleaved(np.zeros(128 * 33, np.float32), 33)
    with pytest.raises(ArgumentException):
        ctx.ProcessBlockInterleaved(np.zeros(100, np.float32), 2)
    ctx.ProcessBlocks([], 3)                 # no output buffers: the blocks are still processed (:167-185)
    assert ctx.CurrentBlock == 3


@pytest.mark.gpu
@pytest.mark.parametrize("how,channels", [("blocks", 2), ("blocks", 4), ("interleaved", 2), ("interleaved", 4), ("interleaved", 1)])
def test_gpu_process_blocks_match_oracle(how, channels):
    from graphaudio_amd import OfflineAudioContext

    def mk(sr):
        ctx = OfflineAudioContext(sr)
        ctx.SetOption("max_chunk_blocks", 7)
        return ctx
    ref = _run(OracleContext, how, channels=channels)
    got = _run(mk, how, channels=channels)
    assert G.rms(ref[:1]) > 1e-3
    scale = G.rms(ref[: min(channels, 2)])
    assert G.rms(ref - got) <= 2e-6 * max(scale, 1e-3)
    assert np.array_equal(ref == 7.0, got == 7.0) and np.array_equal(ref == 0.0, got == 0.0)   # untouched / zero-filled alike
