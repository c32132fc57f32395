"""Holds the CPU ORACLE to float64 mathematics at the sizes BASELINE.json quotes (VERDICT r2 "Next" 1c).

tests/test_oracle_dsp.py stops at 4,096 taps x 1 voice; here the oracle renders the headline graph itself -- all 1024 voices
through the shared 65,536-tap stereo impulse response, every one of the 512 partitions live at the end -- and one GPU's shard of
config 5 (64 sources x 16-channel 32,768-tap private impulse responses), and both are compared with the linear convolution
evaluated in double precision (tests/_f64model.py: written from the definition, independent of oracle/ga_oracle.cpp).

What the figure means: the oracle IS float32 where the reference is (spectra truncated to float before the multiply-accumulate,
float accumulators over 512 partitions, a sequential float sum of 1024 voices at the destination), so its distance to the
float64 truth is the reference algorithm's own rounding noise -- SURVEY.md 8(d) predicts 1-2e-6 absolute on this bus.  A
restatement bug (partition order, FDL indexing, scale, overlap-add) would show up at the 1e-2 .. 1 level.

Cost: ~100 s for config 3 on one host thread, ~15 s for the config 5 shard.
"""
import numpy as np

from tests import _f64model as M
from tests import _graphs as G
from tests._oracle import OracleContext
from tests._report import note

SR = 48000


def test_oracle_config3_1024_voices_65536_taps_against_float64_convolution():
    blocks = 520                       # > P = 512: every partition of the impulse response multiplies non-zero spectra
    frames = blocks * 128
    o = OracleContext(SR)
    G.config3_convolver(o, voices=1024, taps=65536, frames=frames)
    got = G.render(o, 2, frames)
    o.Dispose()
    truth = M.config3_shared(1024, 65536, frames)
    err, sig = M.rms(got - truth), M.rms(truth)
    last = slice((blocks - 8) * 128, None)          # the blocks in which all 512 partitions are live
    err_last, sig_last = M.rms(got[:, last] - truth[:, last]), M.rms(truth[:, last])
    note(f"[oracle vs f64] config 3, 1024 voices x 65,536 taps, {blocks} blocks: bus rms {sig:.4f}, abs rms err {err:.3e} "
          f"(relative {err / sig:.3e}); last 8 blocks: bus {sig_last:.4f}, err {err_last:.3e}")
    assert sig_last > 2.0                           # sigma ~ 2.6 once the tail has built up
    assert err <= 1e-5 and err_last <= 1e-5         # north_star's tolerance, absolute
    assert err / sig < 3e-6 and err_last / sig_last < 3e-6


def test_oracle_config5_64_source_shard_against_float64_convolution():
    blocks = 272                       # > P = 256
    frames = blocks * 128
    o = OracleContext(SR)
    ch = G.config5_ambisonic(o, sources=64, taps=32768, frames=frames)
    got = G.render(o, ch, frames)
    o.Dispose()
    truth = M.config5(64, 32768, frames)
    err, sig = M.rms(got - truth), M.rms(truth)
    note(f"[oracle vs f64] config 5 shard, 64 sources x 16 ch x 32,768 taps, {blocks} blocks: bus rms {sig:.4f}, abs rms err {err:.3e} "
          f"(relative {err / sig:.3e})")
    assert got.shape[0] == 16 and sig > 0.2
    assert err <= 1e-5
    assert err / sig < 3e-6
    for c in range(16):                # per channel (every column of the 16-channel routing, ConvolverNode.cs:145-151)
        assert M.rms(got[c] - truth[c]) / M.rms(truth[c]) < 3e-6, c
