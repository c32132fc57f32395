"""IO ingest: the step BEFORE the hot path -- on-disk audio -> planar float32 `PlayableAudioBuffer`.

Mirrors `GraphAudio.IO.AudioDecoder` (GraphAudio.IO/LibsndfileDecoder.cs:20-222).  The reference decodes through libsndfile
(`sf_open_virtual` + `sf_readf_float`); that library is not in this image, so the uncompressed container formats are parsed
here: RIFF/WAVE (PCM 8/16/24/32 bit, IEEE float 32/64, WAVE_FORMAT_EXTENSIBLE), AIFF / AIFF-C (`NONE`, `twos`, `sowt`, `fl32`,
`fl64`) and Sun/NeXT AU (linear 8/16/24/32, float 32/64), converted to float exactly as libsndfile's `sf_readf_float` does
for those encodings (integer PCM divided by 2^(bits-1); WAV's unsigned 8-bit re-centred).  Compressed formats (FLAC, Ogg, MP3)
would need libsndfile's codecs and are refused.  Everything after the decode -- de-interleave, the `Duration`-based frame
count, `PlayableAudioBuffer` construction -- follows the reference.  `queue_decoded_stream` feeds an AudioStreamSourceNode the
way the reference's AudioDecoderStreamNode does (fixed-size buffers), without the decoder thread.
"""
from __future__ import annotations

import io
import struct
from typing import BinaryIO, List

import numpy as np

from ._capi import ArgumentException, InvalidOperationException
from .core import PlayableAudioBuffer

_TICKS_PER_SECOND = 10_000_000  # System.TimeSpan


class AudioDecoder:
    """AudioDecoder(Stream) (LibsndfileDecoder.cs:24-61): Channels, SampleRate, Duration (TimeSpan ticks), Decode*."""

    def __init__(self, stream: BinaryIO):
        if not stream.readable():
            raise ArgumentException("Stream must be readable.")
        if not stream.seekable():
            raise ArgumentException("Stream must be seekable.")
        self._stream = stream
        try:
            self._parse()
        except (struct.error, ValueError) as e:
            raise InvalidOperationException(f"failed to open stream: {e}") from e
        # Duration = TimeSpan.FromSeconds((double)frames / samplerate) (:57-59): whole 100 ns ticks, truncated
        self.DurationTicks = int((self._frames / self.SampleRate) * _TICKS_PER_SECOND) if self._frames > 0 and self.SampleRate > 0 else 0
        self._pos = 0

    @property
    def DurationTotalSeconds(self) -> float:
        return self.DurationTicks / _TICKS_PER_SECOND

    def _parse(self):
        f = self._stream
        f.seek(0)
        magic = f.read(4)
        f.seek(0)
        self._big = False        # big-endian samples
        self._u8 = True          # 8-bit samples are unsigned (WAV) / signed (AIFF, AU)
        if magic == b"FORM":
            return self._parse_aiff()
        if magic == b".snd":
            return self._parse_au()
        riff, _, wave = struct.unpack("<4sI4s", f.read(12))
        if riff != b"RIFF" or wave != b"WAVE":
            raise ValueError("not a RIFF/WAVE, AIFF or AU stream")
        fmt = None
        self._data_off = self._data_len = -1
        while True:
            hdr = f.read(8)
            if len(hdr) < 8:
                break
            cid, size = struct.unpack("<4sI", hdr)
            if cid == b"fmt ":
                fmt = f.read(size)
            elif cid == b"data":
                self._data_off, self._data_len = f.tell(), size
                f.seek(size, io.SEEK_CUR)
            else:
                f.seek(size, io.SEEK_CUR)
            if size & 1:
                f.seek(1, io.SEEK_CUR)
        if fmt is None or self._data_off < 0:
            raise ValueError("missing fmt or data chunk")
        tag, ch, sr, _, align, bits = struct.unpack("<HHIIHH", fmt[:16])
        if tag == 0xFFFE and len(fmt) >= 26:  # WAVE_FORMAT_EXTENSIBLE: the sub-format GUID starts with the real tag
            tag = struct.unpack("<H", fmt[24:26])[0]
        if tag not in (1, 3) or ch < 1 or sr < 1:
            raise ValueError(f"unsupported WAVE format tag {tag}")
        self._float = tag == 3
        self._bits = bits
        self.Channels, self.SampleRate = ch, sr
        self._bytes_per_frame = align if align else ch * bits // 8
        self._frames = self._data_len // self._bytes_per_frame

    def _parse_aiff(self):
        f = self._stream
        form, _, kind = struct.unpack(">4sI4s", f.read(12))
        if kind not in (b"AIFF", b"AIFC"):
            raise ValueError("not an AIFF stream")
        comm = None
        self._data_off = self._data_len = -1
        while True:
            hdr = f.read(8)
            if len(hdr) < 8:
                break
            cid, size = struct.unpack(">4sI", hdr)
            if cid == b"COMM":
                comm = f.read(size)
            elif cid == b"SSND":
                offset, _ = struct.unpack(">II", f.read(8))
                self._data_off, self._data_len = f.tell() + offset, size - 8 - offset
                f.seek(size - 8, io.SEEK_CUR)
            else:
                f.seek(size, io.SEEK_CUR)
            if size & 1:
                f.seek(1, io.SEEK_CUR)
        if comm is None or self._data_off < 0:
            raise ValueError("missing COMM or SSND chunk")
        ch, frames, bits = struct.unpack(">hIh", comm[:8])
        expo, mant = struct.unpack(">HQ", comm[8:18])          # 80-bit IEEE extended sample rate
        sr = int(round(mant * 2.0 ** ((expo & 0x7FFF) - 16383 - 63))) if mant else 0
        comp = comm[18:22] if kind == b"AIFC" and len(comm) >= 22 else b"NONE"
        self._float = comp in (b"fl32", b"FL32", b"fl64", b"FL64")
        if comp in (b"fl64", b"FL64"):
            bits = 64
        elif self._float:
            bits = 32
        elif comp not in (b"NONE", b"twos", b"sowt"):
            raise ValueError(f"unsupported AIFF-C compression {comp!r}")
        if ch < 1 or sr < 1:
            raise ValueError("bad AIFF COMM chunk")
        self._big = comp != b"sowt"
        self._u8 = False
        self._bits = (bits + 7) // 8 * 8
        self.Channels, self.SampleRate = ch, sr
        self._bytes_per_frame = ch * self._bits // 8
        self._frames = min(frames, self._data_len // self._bytes_per_frame)

    def _parse_au(self):
        f = self._stream
        magic, off, size, enc, sr, ch = struct.unpack(">4sIIIII", f.read(24))
        widths = {2: (8, False), 3: (16, False), 4: (24, False), 5: (32, False), 6: (32, True), 7: (64, True)}
        if enc not in widths or ch < 1 or sr < 1:
            raise ValueError(f"unsupported AU encoding {enc}")
        self._bits, self._float = widths[enc]
        self._big, self._u8 = True, False
        self.Channels, self.SampleRate = ch, sr
        self._bytes_per_frame = ch * self._bits // 8
        f.seek(0, io.SEEK_END)
        avail = f.tell() - off
        self._data_off = off
        self._data_len = avail if size == 0xFFFFFFFF else min(size, avail)
        self._frames = self._data_len // self._bytes_per_frame

    def _read_float(self, frames: int) -> np.ndarray:
        """sf_readf_float: interleaved float32, `frames` frames from the current position."""
        frames = max(0, min(frames, self._frames - self._pos))
        self._stream.seek(self._data_off + self._pos * self._bytes_per_frame)
        raw = self._stream.read(frames * self._bytes_per_frame)
        n = frames * self.Channels
        e = ">" if self._big else "<"
        if self._float:
            x = np.frombuffer(raw, e + ("f4" if self._bits == 32 else "f8"), n).astype(np.float32)
        elif self._bits == 8 and self._u8:
            x = ((np.frombuffer(raw, np.uint8, n).astype(np.float32) - 128.0) / 128.0).astype(np.float32)
        elif self._bits == 8:
            x = (np.frombuffer(raw, np.int8, n).astype(np.float32) / np.float32(128.0)).astype(np.float32)
        elif self._bits == 16:
            x = (np.frombuffer(raw, e + "i2", n).astype(np.float32) / np.float32(32768.0)).astype(np.float32)
        elif self._bits == 24:
            b = np.frombuffer(raw, np.uint8, n * 3).reshape(n, 3).astype(np.int32)
            v = (b[:, 2] | (b[:, 1] << 8) | (b[:, 0] << 16)) if self._big else (b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16))
            v = np.where(v & 0x800000, v - 0x1000000, v)
            x = (v.astype(np.float32) / np.float32(8388608.0)).astype(np.float32)
        elif self._bits == 32:
            x = (np.frombuffer(raw, e + "i4", n).astype(np.float64) / 2147483648.0).astype(np.float32)
        else:
            raise InvalidOperationException(f"unsupported PCM width {self._bits}")
        self._pos += frames
        return x

    def Decode(self, buffer: np.ndarray) -> int:  # :73-84, interleaved
        if self.Channels == 0:
            return 0
        frames = buffer.size // self.Channels
        if frames <= 0:
            return 0
        x = self._read_float(frames)
        buffer.reshape(-1)[: x.size] = x
        return x.size // self.Channels

    def DecodePlanar(self, channels: List[np.ndarray]) -> int:  # :93-140 + DeinterleaveFrames
        if self.Channels == 0 or len(channels) == 0:
            return 0
        frames = channels[0].shape[0]
        if frames <= 0:
            return 0
        if any(c.shape[0] != frames for c in channels):
            raise ArgumentException("All channel buffers must have the same length.")
        if len(channels) != self.Channels:
            raise ArgumentException(f"Expected {self.Channels} channels, but got {len(channels)}.")
        x = self._read_float(frames)
        got = x.size // self.Channels
        inter = x.reshape(got, self.Channels)
        for ch in range(self.Channels):
            channels[ch][:got] = inter[:, ch]
        return got

    @staticmethod
    def LoadFromStream(stream: BinaryIO) -> PlayableAudioBuffer:  # :195-222
        dec = AudioDecoder(stream)
        # `(long)(decoder.Duration.TotalSeconds * decoder.SampleRate)` (:199): Duration is in whole 100 ns ticks, so a length
        # that is not a multiple of 100 ns loses its last frame (e.g. 1540 frames @ 48 kHz -> 1539)
        total = int(dec.DurationTotalSeconds * dec.SampleRate)
        if total <= 0 or total > 2**31 - 1:
            raise InvalidOperationException(f"Invalid audio duration or frame count: {total}")
        chans = [np.zeros(total, np.float32) for _ in range(dec.Channels)]
        dec.DecodePlanar(chans)
        return PlayableAudioBuffer.FromChannelArrays(chans, dec.SampleRate)

    @staticmethod
    def LoadFromFile(path: str) -> PlayableAudioBuffer:  # LoadFromFileAsync, :227-236 (synchronous here)
        with open(path, "rb") as f:
            return AudioDecoder.LoadFromStream(io.BytesIO(f.read()))


def queue_decoded_stream(node, stream: BinaryIO, bufferSize: int = 4096) -> int:
    """Feed an `AudioStreamSourceNode` the way `AudioDecoderStreamNode` does (GraphAudio.IO/AudioDecoderStreamNode.cs:47-88:
    planar buffers of `bufferSize` frames at the file's sample rate), but ahead of time instead of from a decoder thread: the
    whole stream is decoded and queued, so an offline render never waits for the decoder.  Returns the number of buffers."""
    if bufferSize <= 0:
        raise ArgumentException("bufferSize")
    dec = AudioDecoder(stream)
    count = 0
    while True:
        chans = [np.zeros(bufferSize, np.float32) for _ in range(dec.Channels)]
        got = dec.DecodePlanar(chans)
        if got <= 0:
            break
        node.QueueBuffer(PlayableAudioBuffer.FromChannelArrays([c[:got].copy() for c in chans], dec.SampleRate))
        count += 1
        if got < bufferSize:
            break
    return count
