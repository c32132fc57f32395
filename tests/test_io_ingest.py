"""IO ingest: WAV -> PlayableAudioBuffer, mirroring GraphAudio.IO.AudioDecoder."""
import io
import struct
import wave

import numpy as np
import pytest

from graphaudio_amd import AudioBufferSourceNode, ConvolverNode, InvalidOperationException
from graphaudio_amd.io import AudioDecoder
from tests import _graphs as G
from tests._oracle import OracleContext


def wav_pcm(data_i, sr, width):
    """data_i: int array [frames, channels]"""
    bio = io.BytesIO()
    with wave.open(bio, "wb") as w:
        w.setnchannels(data_i.shape[1])
        w.setsampwidth(width)
        w.setframerate(sr)
        if width == 2:
            w.writeframes(data_i.astype("<i2").tobytes())
        elif width == 3:
            b = data_i.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3]
            w.writeframes(b.tobytes())
        else:
            w.writeframes(data_i.astype("<i4").tobytes())
    bio.seek(0)
    return bio


def wav_float32(x, sr):
    frames, ch = x.shape
    data = x.astype("<f4").tobytes()
    fmt = struct.pack("<HHIIHH", 3, ch, sr, sr * ch * 4, ch * 4, 32)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 3) + b"abc\0" + b"data" + struct.pack("<I", len(data)) + data
    return io.BytesIO(b"RIFF" + struct.pack("<I", len(body)) + body)


def test_pcm16_24_32_and_float_are_scaled_like_sf_readf_float():
    rng = np.random.default_rng(0)
    fr = 48000   # a whole second: Duration is exact, no frame lost
    for width, full in ((2, 32768), (3, 8388608), (4, 2147483648)):
        xi = rng.integers(-full, full, size=(fr, 2))
        buf = AudioDecoder.LoadFromStream(wav_pcm(xi, 48000, width))
        assert buf.NumberOfChannels == 2 and buf.Length == fr and buf.SampleRate == 48000
        for c in range(2):
            assert np.array_equal(buf.GetChannelData(c), (xi[:, c] / full).astype(np.float32))
    x = (rng.standard_normal((4410, 3)) * 0.3).astype(np.float32)
    buf = AudioDecoder.LoadFromStream(wav_float32(x, 44100))
    assert buf.NumberOfChannels == 3 and buf.SampleRate == 44100 and buf.Length == 4410
    assert np.array_equal(np.stack([buf.GetChannelData(c) for c in range(3)], 1), x)


def test_duration_in_ticks_drops_the_last_frame_of_most_lengths():
    """`(long)(Duration.TotalSeconds * SampleRate)` with Duration in 100 ns ticks (LibsndfileDecoder.cs:57-59,199)."""
    x = np.ones((1540, 1), np.float32)
    buf = AudioDecoder.LoadFromStream(wav_float32(x, 48000))
    assert buf.Length == 1539
    with pytest.raises(InvalidOperationException):
        AudioDecoder.LoadFromStream(io.BytesIO(b"RIFFxxxxJUNK"))


def test_decoder_streaming_decode_and_planar():
    rng = np.random.default_rng(1)
    x = (rng.standard_normal((1000, 2)) * 0.2).astype(np.float32)
    dec = AudioDecoder(wav_float32(x, 48000))
    a = np.zeros(600, np.float32)
    assert dec.Decode(a) == 300 and np.array_equal(a.reshape(300, 2), x[:300])
    ch = [np.zeros(800, np.float32), np.zeros(800, np.float32)]
    assert dec.DecodePlanar(ch) == 700                      # only 700 frames left
    assert np.array_equal(ch[0][:700], x[300:, 0]) and np.array_equal(ch[1][:700], x[300:, 1])


@pytest.mark.gpu
def test_wav_impulse_response_and_voice_through_the_device_path():
    from graphaudio_amd import OfflineAudioContext
    rng = np.random.default_rng(2)
    ir = (rng.standard_normal((4800, 2)) * 0.05 * np.exp(-np.arange(4800) / 900.0)[:, None]).astype(np.float32)
    voice_i = rng.integers(-20000, 20000, size=(48000, 1))
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(48000)
        s = AudioBufferSourceNode(ctx)
        s.Buffer = AudioDecoder.LoadFromStream(wav_pcm(voice_i, 48000, 2))
        c = ConvolverNode(ctx)
        c.Buffer = AudioDecoder.LoadFromStream(wav_float32(ir, 48000))
        s.Connect(c)
        c.Connect(ctx.Destination)
        s.Start()
        out = np.zeros((2, 128 * 100), np.float32)
        ctx.Render(out, 128 * 100)
        outs.append(out)
    ref, got = outs
    assert G.rms(ref) > 1e-3
    assert G.rms(ref - got) <= 2e-6 * G.rms(ref)


# ---- AIFF / AU containers (the other uncompressed formats libsndfile would decode) and the decoded stream feed ----
class _KeepOpen(io.BytesIO):
    """aifc / sunau close the file object they were given"""
    def close(self):
        pass


def test_aiff_and_au_pcm16_decode_like_wav():
    import aifc
    import sunau
    rng = np.random.default_rng(5)
    data = rng.integers(-30000, 30000, size=(777, 2)).astype(np.int16)
    a = _KeepOpen()
    w = aifc.open(a, "wb")
    w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100)
    w.writeframes(data.astype(">i2").tobytes())
    w.close()
    a = io.BytesIO(a.getvalue())
    u = _KeepOpen()
    w = sunau.open(u, "wb")
    w.setcomptype("NONE", "not compressed")
    w.setnchannels(2); w.setsampwidth(2); w.setframerate(44100)
    w.writeframes(data.astype(">i2").tobytes())
    w.close()
    u = io.BytesIO(u.getvalue())
    want = (data.astype(np.float32) / np.float32(32768.0)).T
    for s in (a, u):
        dec = AudioDecoder(s)
        assert (dec.Channels, dec.SampleRate) == (2, 44100)
        ch = [np.zeros(777, np.float32) for _ in range(2)]
        assert dec.DecodePlanar(ch) == 777
        assert np.array_equal(np.stack(ch), want)


def test_aiff_24_bit_and_au_float():
    import aifc
    rng = np.random.default_rng(6)
    v = rng.integers(-(1 << 23), 1 << 23, size=(300, 1)).astype(np.int32)
    a = _KeepOpen()
    w = aifc.open(a, "wb")
    w.setnchannels(1); w.setsampwidth(3); w.setframerate(48000)
    w.writeframes(v.astype(">i4").view(np.uint8).reshape(-1, 4)[:, 1:].tobytes())
    w.close()
    dec = AudioDecoder(io.BytesIO(a.getvalue()))
    ch = [np.zeros(300, np.float32)]
    assert dec.DecodePlanar(ch) == 300
    assert np.array_equal(ch[0], (v[:, 0].astype(np.float32) / np.float32(8388608.0)))
    x = rng.standard_normal((200, 2)).astype(np.float32)
    au = struct.pack(">4sIIIII", b".snd", 24, x.size * 4, 6, 32000, 2) + x.astype(">f4").tobytes()
    dec = AudioDecoder(io.BytesIO(au))
    ch = [np.zeros(200, np.float32) for _ in range(2)]
    assert dec.DecodePlanar(ch) == 200 and dec.SampleRate == 32000
    assert np.array_equal(np.stack(ch), x.T)


def test_decoded_stream_feed_plays_the_file_through_the_stream_node():
    """queue_decoded_stream + AudioStreamSourceNode on the oracle: at the context's sample rate the stream node plays the file's
    samples back to back (copy path, AudioStreamSourceNodeBase.cs:222-240)."""
    from graphaudio_amd import AudioStreamSourceNode
    from graphaudio_amd.io import queue_decoded_stream
    rng = np.random.default_rng(7)
    data = rng.integers(-20000, 20000, size=(10000, 1)).astype(np.int16)
    ctx = OracleContext(48000)
    s = AudioStreamSourceNode(ctx)
    assert queue_decoded_stream(s, wav_pcm(data, 48000, 2), bufferSize=4096) == 3
    s.Connect(ctx.Destination)
    s.Play()
    out = G.render(ctx, 2, 128 * 80)
    want = data[:, 0].astype(np.float32) / np.float32(32768.0)
    assert np.array_equal(out[0, :10000], want) and np.array_equal(out[1, :10000], want)
    assert np.abs(out[:, 10000:]).max() == 0.0
