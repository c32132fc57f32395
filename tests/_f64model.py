"""Float64 models of the convolver graphs -- mathematical truth, written from the definition of the operation, not from the
oracle or the product.

The reference holds no tests or vectors, so the strongest pin available for BOTH the CPU oracle and the HIP
path is the linear convolution itself evaluated in double precision:

    bus[c] = sum_v  x_v * (scale_{v,c} h_{v,c})          (PartitionedConvolver.cs:104-223 is a zero-latency evaluation of this)

`normalization_scale` restates PartitionedConvolver.cs:93-102 in numpy (float32 products, double accumulation) independently of
oracle/ga_oracle.cpp.  Convolutions run through numpy's float64 FFT (error ~1e-15 relative -- ten orders below the float32
paths they are compared with).
"""
import numpy as np

from tests import _graphs as G


def normalization_scale(ir):
    """PartitionedConvolver.cs:93-102: float products accumulated in double, power and scale in float."""
    r = np.ascontiguousarray(ir, dtype=np.float32)
    sq = (r * r).astype(np.float32)                       # response[i] * response[i] is a float product
    power = np.float32(np.sqrt(np.sum(sq.astype(np.float64)) / len(r)))
    if not np.isfinite(power) or power < np.float32(0.000125):
        power = np.float32(0.000125)
    cal = np.float32(np.float32(-58) * np.float32(0.05))  # GainCalibration * 0.05f (float), promoted inside Math.Pow
    return np.float32(np.float32(1.0) / power) * np.float32(np.power(10.0, float(cal)))


def scaled_ir64(ir, normalize=True):
    """the taps the convolver multiplies with: float32 ir x float32 scale (PartitionedConvolver.cs:75-80), as float64 values"""
    ir = np.ascontiguousarray(ir, dtype=np.float32)
    s = normalization_scale(ir) if normalize else np.float32(1.0)
    return (ir * s).astype(np.float64)


def linear_conv(x64, h64, nout):
    """first `nout` samples of the linear convolution x * h, float64 (FFT)"""
    n = len(x64) + len(h64) - 1
    nfft = 1 << (n - 1).bit_length()
    y = np.fft.irfft(np.fft.rfft(x64, nfft) * np.fft.rfft(h64, nfft), nfft)
    return y[:nout]


def circular_conv(x64, h64):
    """circular convolution of period len(x64) (len(h64) <= len(x64)): the steady state of a looping voice"""
    n = len(x64)
    assert len(h64) <= n
    return np.fft.irfft(np.fft.rfft(x64, n) * np.fft.rfft(h64, n), n)


def voices_sum64(v0, v1, n, scale=0.25):
    acc = np.zeros(n, np.float64)
    for v in range(v0, v1):
        acc += G.voice(v, n, scale)
    return acc


def config3_shared(voices, taps, frames, ir_channels=2, voice_len=None):
    """tests/_graphs.py::config3_convolver(shared=True): every voice through the same `ir_channels`-channel impulse response"""
    vlen = voice_len or (frames + 256)
    xs = np.zeros(frames, np.float64)
    for v in range(voices):
        xs += G.voice(v, vlen)[:frames]
    return np.stack([linear_conv(xs, scaled_ir64(G.synth_ir(c, taps)), frames) for c in range(ir_channels)])


def config3_private(voices, taps, frames, ir_channels=2, voice_len=None):
    """config3_convolver(shared=False): voice v through its own impulse response (seed0 = 7 + 100 (v + 1))"""
    vlen = voice_len or (frames + 256)
    n = frames + taps - 1
    nfft = 1 << (n - 1).bit_length()
    acc = np.zeros((ir_channels, nfft // 2 + 1), np.complex128)
    for v in range(voices):
        X = np.fft.rfft(G.voice(v, vlen)[:frames].astype(np.float64), nfft)
        for c in range(ir_channels):
            acc[c] += X * np.fft.rfft(scaled_ir64(G.synth_ir(c, taps, seed0=7 + 100 * (v + 1))), nfft)
    return np.stack([np.fft.irfft(acc[c], nfft)[:frames] for c in range(ir_channels)])


def config5(sources, taps, frames, ir_channels=16, v0=0):
    """tests/test_gpu_atsize.py::_config5: source v through its own 16-channel impulse response (seed0 = 7 + 100 v)"""
    n = frames + taps - 1
    nfft = 1 << (n - 1).bit_length()
    acc = np.zeros((ir_channels, nfft // 2 + 1), np.complex128)
    for v in range(v0, v0 + sources):
        X = np.fft.rfft(G.voice(v, frames + 256)[:frames].astype(np.float64), nfft)
        for c in range(ir_channels):
            acc[c] += X * np.fft.rfft(scaled_ir64(G.synth_ir(c, taps, seed0=7 + 100 * v)), nfft)
    return np.stack([np.fft.irfft(acc[c], nfft)[:frames] for c in range(ir_channels)])


def rms(a):
    return float(np.sqrt(np.mean(np.square(np.asarray(a, dtype=np.float64)))))
