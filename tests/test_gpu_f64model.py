"""The HIP path held to FLOAT64 MATHEMATICS at the sizes BASELINE.json quotes, at full partition depth (VERDICT r2 "Next" 1b, 1d).

The reference holds nothing that pins parity, so besides the oracle comparisons of tests/test_gpu_atsize.py every
formulation of the convolver is compared here with the linear convolution itself, evaluated in double precision by
tests/_f64model.py (independent of both the oracle and the product):

  * config 3, all 1024 voices x 65,536-tap stereo IR, 640 blocks (> 512: all 512 fine and all 8 coarse partitions live):
    formulation D with the time-domain pre-mix (default), D per voice (spectra summed: coarse_sum_kernel), C (block-axis FFT);
    one call, and two calls (the second from the carried tails);
  * 256 voices x PRIVATE 65,536-tap stereo IRs, 640 blocks (the general multiply-accumulate kernel, coarse_mac_kernel);
  * config 5: one GPU's shard (64 sources x 16-channel 32,768-tap private IRs) and the whole configuration (512 sources) on one
    GPU, 400 blocks, plus chunk invariance at full size.

Tolerance: north_star's 1e-5 RMS per sample absolute; the bus-relative figure is asserted at 2e-6.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import OfflineAudioContext
from tests import _f64model as M
from tests import _graphs as G
from tests._report import note

SR = 48000
TOL = 1e-5

_truth = {}


def _config3_truth(frames):
    if frames not in _truth:
        _truth[frames] = M.config3_shared(1024, 65536, frames)
    return _truth[frames]


def _check(name, got, truth, rel=2e-6):
    err, sig = M.rms(got - truth), M.rms(truth)
    note(f"[f64] {name}: bus rms {sig:.4f}  abs rms err {err:.3e}  relative {err / sig:.3e}")
    assert err <= TOL, (name, err)
    assert err / sig < rel, (name, err / sig)
    return err, sig


FORMS = {
    "D, pre-mixed group (default)": {},
    "D, per-voice transforms, spectra summed": {"coarse_premix": 0},
    "C, block-axis FFT": {"coarse": 0},
}


@pytest.mark.parametrize("form", list(FORMS))
def test_config3_1024_voices_full_partition_depth_against_float64(form):
    blocks = 640
    frames = blocks * 128
    truth = _config3_truth(frames)
    h = OfflineAudioContext(SR)
    for k, v in FORMS[form].items():
        h.SetOption(k, v)
    G.config3_convolver(h, voices=1024, taps=65536, frames=frames)
    got = G.render(h, 2, frames)
    st = h.GetStats()
    h.Dispose()
    assert (st["stage_launches"][5] > 0) == form.startswith("D")
    assert (st["coarse_premixed_signals"] > 0) == form.startswith("D, pre-mixed")
    err, sig = _check(f"config 3, 1024 voices x 65,536 taps, {blocks} blocks, {form}", got, truth)
    assert sig > 2.0
    last = slice(576 * 128, None)   # all partitions live
    _check(f"  last 64 blocks, {form}", got[:, last], truth[:, last])


@pytest.mark.parametrize("premix", [1, 0])
def test_config3_1024_voices_in_three_calls_against_float64(premix):
    """the steady state of the benchmark's steps: calls 2 and 3 start from the carried tails"""
    frames = 640 * 128
    truth = _config3_truth(frames)
    h = OfflineAudioContext(SR)
    h.SetOption("coarse_premix", premix)
    G.config3_convolver(h, voices=1024, taps=65536, frames=frames)
    got = np.zeros((2, frames), np.float32)
    pos = 0
    for n in (300 * 128, 260 * 128, 80 * 128):
        h.Render(got, n, pos)
        pos += n
    st = h.GetStats()
    h.Dispose()
    assert st["coarse_carried_outputs"] == 4
    _check(f"config 3, 1024 voices, 300 + 260 + 80 blocks, premix={premix}", got, truth)


def test_config3_256_voices_private_impulse_responses_against_float64():
    blocks = 640
    frames = blocks * 128
    truth = M.config3_private(256, 65536, frames)
    for pieces in ([frames], [280 * 128, 360 * 128]):
        h = OfflineAudioContext(SR)
        G.config3_convolver(h, voices=256, taps=65536, frames=frames, shared=False)
        got = np.zeros((2, frames), np.float32)
        pos = 0
        for n in pieces:
            h.Render(got, n, pos)
            pos += n
        st = h.GetStats()
        h.Dispose()
        assert st["stage_launches"][6] > 0 and st["coarse_premixed_signals"] == 0
        _check(f"256 voices x private 65,536-tap stereo IRs, {[p // 128 for p in pieces]} blocks", got, truth)


def test_config5_64_source_shard_against_float64():
    frames = 400 * 128
    truth = M.config5(64, 32768, frames)
    h = OfflineAudioContext(SR)
    ch = G.config5_ambisonic(h, sources=64, taps=32768, frames=frames)
    got = G.render(h, ch, frames)
    h.Dispose()
    _check("config 5 shard, 64 sources x 16 ch x 32,768 taps, 400 blocks", got, truth)
    for c in range(16):
        assert M.rms(got[c] - truth[c]) / M.rms(truth[c]) < 2e-6, c


def test_config5_all_512_sources_on_one_gpu_against_float64_and_chunk_invariance():
    frames = 400 * 128
    truth = M.config5(512, 32768, frames)
    outs = []
    for pieces in ([frames], [150 * 128, 130 * 128, 120 * 128]):
        h = OfflineAudioContext(SR)
        ch = G.config5_ambisonic(h, sources=512, taps=32768, frames=frames)
        got = np.zeros((ch, frames), np.float32)
        pos = 0
        for n in pieces:
            h.Render(got, n, pos)
            pos += n
        h.Dispose()
        outs.append(got)
        _check(f"config 5, 512 sources x 16 ch x 32,768 taps, {[p // 128 for p in pieces]} blocks", got, truth)
    # any chunking of a render gives the same result to float32 rounding
    assert M.rms(outs[0] - outs[1]) / M.rms(truth) < 1e-6
    # superposition: the bus of all 512 sources is the sum of the eight 64-source shards' buses (SURVEY.md 8e partitioning)
    acc = np.zeros((16, frames), np.float64)
    for r in range(8):
        h = OfflineAudioContext(SR)
        G.config5_ambisonic(h, sources=64, taps=32768, frames=frames, v0=64 * r)
        acc += G.render(h, 16, frames)
        h.Dispose()
    assert M.rms(acc - outs[0]) / M.rms(truth) < 1e-6
