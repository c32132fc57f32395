"""Pins the CPU oracle's DSP primitives against independent analytic models (numpy / scipy, float64).

The reference ships no tests or golden vectors for this path, so these known-answer tests are what
anchors the restatement: FFT convention, zero-latency partitioned convolution, normalisation scale, Catmull-Rom
resampling.
"""
import numpy as np
import pytest

from tests import _oracle as O


def test_rfft256_matches_numpy_convention():
    rng = np.random.default_rng(0)
    for _ in range(8):
        x = rng.standard_normal(256)
        X = O.rfft256(x)
        ref = np.fft.rfft(x)  # forward e^{-i w n}, unscaled (RealFourierTransform.cs:62-88)
        assert np.abs(X - ref).max() < 1e-12
        assert X[0].imag == 0.0 and X[128].imag == 0.0
        assert np.abs(O.irfft256(ref) - x).max() < 1e-13  # inverse carries the full 1/N scale (:46,129)


@pytest.mark.parametrize("taps", [1, 100, 128, 129, 1000, 4096])
def test_convolver_is_zero_latency_linear_convolution(taps):
    rng = np.random.default_rng(taps)
    ir = rng.standard_normal(taps).astype(np.float32)
    x = (rng.standard_normal(128 * 40) * 0.25).astype(np.float32)
    y = O.convolve(ir, x, normalize=False)
    ref = np.convolve(x.astype(np.float64), ir.astype(np.float64))[: len(x)]
    rel = np.sqrt(np.mean((y - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))
    assert rel < 5e-7, rel


def test_convolver_impulse_returns_scaled_ir():
    ir = np.random.default_rng(3).standard_normal(300).astype(np.float32)
    x = np.zeros(128 * 4, np.float32)
    x[0] = 1.0
    y = O.convolve(ir, x, normalize=True)
    scale = np.float32(O.normalization_scale(ir))
    assert np.allclose(y[:300], ir * scale, rtol=2e-6, atol=1e-9)
    assert np.abs(y[300:]).max() < 1e-8


def test_normalization_scale_known_answers():
    # rms = 0.5 -> (1/0.5) * 10^(-58*0.05)   (PartitionedConvolver.cs:93-102)
    ir = np.full(1000, 0.5, np.float32)
    expect = np.float32(1.0 / 0.5) * np.float32(10.0 ** float(np.float32(-58) * np.float32(0.05)))
    assert O.normalization_scale(ir) == pytest.approx(float(expect), rel=1e-7)
    # all-zero IR -> MinPower branch
    z = np.zeros(64, np.float32)
    expect0 = np.float32(1.0 / np.float32(0.000125)) * np.float32(10.0 ** float(np.float32(-58) * np.float32(0.05)))
    assert O.normalization_scale(z) == pytest.approx(float(expect0), rel=1e-7)
    # shape independence: scaled rms is 10^(-58/20) whatever the IR
    r = np.random.default_rng(5).standard_normal(5000).astype(np.float32) * 3
    s = O.normalization_scale(r)
    assert np.sqrt(np.mean((r.astype(np.float64) * s) ** 2)) == pytest.approx(10 ** (-58 / 20), rel=1e-5)


def test_resampler_first_output_is_second_input_and_linear_is_exact():
    x = np.arange(200, dtype=np.float32)
    out, consumed, produced = O.resample(x, 128, 44100 / 48000.0)
    assert produced == 128 and consumed == 120  # SURVEY 8c sanity value
    assert out[0] == 1.0  # first four inputs only prime the window (CubicResampler.cs:31-38)
    expect = 1.0 + np.arange(128) * (44100 / 48000.0)
    assert np.allclose(out, expect, rtol=0, atol=2e-4)


def test_resampler_sine_accuracy_and_exhaustion():
    n = 4000
    f = 440.0
    x = np.sin(2 * np.pi * f * np.arange(n) / 44100.0).astype(np.float32)
    rate = 44100 / 48000.0
    out, consumed, produced = O.resample(x, 4096, rate)
    t = (1.0 + np.arange(produced) * rate) / 44100.0
    assert np.abs(out - np.sin(2 * np.pi * f * t)).max() < 2e-4  # cubic interpolation bound at 440 Hz
    # asks for more output than the input can feed: stops early, never reads past the end
    out2, consumed2, produced2 = O.resample(x[:50], 128, rate)
    assert produced2 < 128 and consumed2 <= 50


def test_resampler_needs_four_samples_to_prime():
    out, consumed, produced = O.resample(np.ones(3, np.float32), 16, 0.5)
    assert produced == 0 and consumed == 3
