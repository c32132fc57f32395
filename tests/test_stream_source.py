"""AudioStreamNodeBase (GraphAudio.IO/AudioStreamSourceNodeBase.cs:19-329) with the queue filled by the host -- SURVEY.md 8(f)
rank 3, the second call site of the CubicResampler.  CPU: the oracle's restatement against an independent numpy model of the
queue / resampler arithmetic; GPU: the HIP path against the oracle, bit-exact (sources are index work + the per-sample float32
polynomial)."""
import numpy as np
import pytest

from graphaudio_amd import (AudioStreamSourceNode, GainNode, OfflineAudioContext, PlayableAudioBuffer, StreamState)
from tests import _graphs as G
from tests._oracle import OracleContext

SR = 48000


def _buffers(spec, seed=0):
    rng = np.random.default_rng(seed)
    out = []
    for (n, ch, rate) in spec:
        out.append(PlayableAudioBuffer.FromChannelArrays([(rng.standard_normal(n) * 0.3).astype(np.float32) for _ in range(ch)], rate))
    return out


def _render(mk, spec, frames, pieces=None, rate=None, script=None):
    ctx = mk(SR)
    s = AudioStreamSourceNode(ctx)
    bufs = _buffers(spec)
    for b in bufs:
        s.QueueBuffer(b)
    if rate is not None:
        s.PlaybackRate.Value = rate
    s.Connect(ctx.Destination)
    s.Play()
    out = np.zeros((2, frames), np.float32)
    pos = 0
    for i, n in enumerate(pieces or [frames]):
        if script:
            script(i, s, bufs)
        n = min(n, frames - pos)
        if n > 0:
            ctx.Render(out, n, pos)
            pos += n
    info = (s.QueuedBufferCount, s.ProcessedBufferCount)
    ctx.Dispose()
    return out, info


def model_copy_path(spec, frames, seed=0):
    """rate == 1: the buffers played back to back, mono copied to both destination channels"""
    rng = np.random.default_rng(seed)
    parts = [[(rng.standard_normal(n) * 0.3).astype(np.float32) for _ in range(ch)] for (n, ch, rate) in spec]
    out = np.zeros((2, frames), np.float32)
    pos = 0
    for p in parts:
        n = min(len(p[0]), frames - pos)
        if n <= 0:
            break
        out[0, pos:pos + n] = p[0][:n]
        out[1, pos:pos + n] = (p[1] if len(p) > 1 else p[0])[:n]
        pos += n
    return out


def test_oracle_copy_path_concatenates_buffers():
    spec = [(300, 1, SR), (1000, 1, SR), (77, 1, SR), (5000, 1, SR)]
    frames = 128 * 40
    got, info = _render(OracleContext, spec, frames)
    want = model_copy_path(spec, frames)
    assert np.array_equal(got, want)
    assert info == (0, 3)     # the fourth buffer is still being played (it is the current buffer, not queued)


def test_oracle_resampled_stream_model():
    """44.1 kHz buffers at 48 kHz: Catmull-Rom over the concatenated stream, minus the <= 4 samples dropped at each buffer end
    (`_currentBufferPosition >= Length - 4` retires the buffer, :278-283)."""
    spec = [(2000, 1, 44100), (3000, 1, 44100)]
    frames = 128 * 30
    got, _ = _render(OracleContext, spec, frames)
    rng = np.random.default_rng(0)
    a, b = [(rng.standard_normal(n) * 0.3).astype(np.float32) for (n, ch, r) in spec]
    # independent model of the reference's loop: feed samples, retire a buffer once fewer than 5 unread samples are left
    rate = 44100 / 48000.0
    S = [np.float32(0)] * 4
    Pos, ready = 0.0, 0
    out = []
    stream = []
    for buf in (a, b):
        pos = 0
        while True:
            avail = len(buf) - pos
            consumed = 0
            while ready < 4 and consumed < avail:
                S = S[1:] + [buf[pos + consumed]]
                consumed += 1
                ready += 1
            blocked = False
            if ready == 4:
                while len(out) % 128 != 0 or consumed == 0 or True:
                    if len(out) >= frames:
                        break
                    c = int(Pos)
                    if consumed + c > avail:
                        blocked = True
                        break
                    for _ in range(c):
                        S = S[1:] + [buf[pos + consumed]]
                        consumed += 1
                    Pos -= c
                    t = np.float32(Pos)
                    S0, S1, S2, S3 = S
                    y = S1 + t * (np.float32(0.5) * (S2 - S0) + t * ((S0 - np.float32(2.5) * S1 + np.float32(2) * S2 - np.float32(0.5) * S3)
                                                                    + t * (np.float32(0.5) * (S3 - S0) + np.float32(1.5) * (S1 - S2))))
                    out.append(np.float32(y))
                    Pos += rate
                    if len(out) % 128 == 0:
                        break
            pos += consumed
            if pos >= len(buf) - 4 or len(out) >= frames:
                break
            if consumed == 0:
                out.extend([np.float32(0)] * ((-len(out)) % 128))   # rest of the block cleared
        if len(out) >= frames:
            break
    out = np.array(out[:frames] + [np.float32(0)] * max(0, frames - len(out)), np.float32)
    n_cmp = int(0.9 * (4990 / rate)) // 128 * 128   # well inside the streamed samples
    assert np.array_equal(got[0, :n_cmp], out[:n_cmp])


def test_oracle_states_and_processed_queue():
    spec = [(128 * 3, 1, SR), (128 * 3, 1, SR)]
    frames = 128 * 12

    def script(i, s, bufs):
        if i == 1:
            s.Pause()
        if i == 2:
            s.Play()
        if i == 3:
            s.Stop()          # flushes the rest to the processed list

    got, info = _render(OracleContext, spec, frames, pieces=[128 * 2, 128 * 2, 128 * 2, 128 * 6], script=script)
    want = model_copy_path(spec, frames)
    exp = np.zeros_like(want)
    exp[:, :256] = want[:, :256]              # played
    exp[:, 512:768] = want[:, 256:512]        # resumed after the pause where it stopped
    assert np.array_equal(got, exp)
    assert info == (0, 2)


CASES = {
    "copy": dict(spec=[(300, 1, SR), (1000, 2, SR)], frames=128 * 12),                       # channel-count change: back of the queue
    "resample": dict(spec=[(2000, 1, 44100), (3000, 1, 44100), (500, 1, 44100)], frames=128 * 50),
    "mixed_rates": dict(spec=[(2000, 2, 44100), (1500, 2, SR), (4000, 2, 22050), (700, 2, SR)], frames=128 * 110),
    "rate_param": dict(spec=[(6000, 1, SR), (6000, 1, SR)], frames=128 * 70, rate=1.37),
    "tiny_buffers": dict(spec=[(3, 1, 44100), (5, 1, 44100), (9, 1, 44100), (4000, 1, 44100)], frames=128 * 40),
}


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(CASES))
def test_hip_matches_oracle(name):
    kw = CASES[name]
    ref, ri = _render(OracleContext, **kw)
    got, gi = _render(OfflineAudioContext, **kw)
    assert G.rms(ref) > 1e-3
    assert np.array_equal(ref, got)
    assert ri == gi


@pytest.mark.gpu
def test_hip_pieces_states_and_requeue():
    """uneven render pieces (state across chunks: window on the device, position on the host), pause / resume, buffers queued
    while playing, a k-rate playbackRate ramp"""
    def script(i, s, bufs):
        if i == 2:
            s.Pause()
        if i == 3:
            s.Play()
            s.QueueBuffer(bufs[0])       # a processed buffer goes round again
        if i == 4:
            s.PlaybackRate.SetValueAtTime(1.0, 0.05)
            s.PlaybackRate.LinearRampToValueAtTime(0.6, 0.12)

    kw = dict(spec=[(3000, 2, 44100), (2500, 2, 44100)], frames=128 * 90, pieces=[100, 128 * 7 + 3, 128 * 4, 128 * 20, 128 * 60], script=script)
    ref, ri = _render(OracleContext, **kw)
    got, gi = _render(OfflineAudioContext, **kw)
    assert np.array_equal(ref, got)
    assert ri == gi


@pytest.mark.gpu
def test_hip_stream_into_graph():
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(SR)
        s = AudioStreamSourceNode(ctx)
        for b in _buffers([(5000, 1, 44100), (5000, 1, 32000)]):
            s.QueueBuffer(b)
        g = GainNode(ctx)
        g.Gain.SetValueAtTime(0.0, 0.0)
        g.Gain.LinearRampToValueAtTime(1.0, 0.1)
        s.Connect(g).Connect(ctx.Destination)
        s.Play()
        outs.append(G.render(ctx, 2, 128 * 100))
    assert G.rms(outs[0]) > 1e-3
    assert np.array_equal(outs[0], outs[1])
