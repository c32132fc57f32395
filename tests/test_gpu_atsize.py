"""Parity against the CPU oracle AT THE SIZES BASELINE.json quotes (the 1 s short form of every configuration, SURVEY.md 8(d)
"Parity check alongside timing"), not on scaled-down stand-ins:

  * config 3: all 1024 voices -> per-voice ConvolverNode (shared 65,536-tap stereo IR) -> 2-channel destination, 375 blocks.
    This is the measurement of the 1024-term destination sum (AudioNodeInput.cs:118-132,195-198) that the round-1 design
    note only extrapolated from 4 voices.
  * config 5: 16-channel 32,768-tap private IRs (P = 256, 16 columns per input; Nodes/ConvolverNode.cs:145-151).
  * config 2 at 256 voices, config 4 at 512 voices (one GPU's shard of the 4096): the multi-workgroup indexing of the
    single-section biquad, the 5-section pipeline and the resampler at the job counts they run with.
  * a level with more than 65,535 jobs (gridDim.y windows of the job-table launchers).

Tolerance (north_star): <= 1e-5 RMS per sample, absolute, float32; the bus-relative error is asserted as well.
Oracle cost on the GPU box's host: ~45 s for config 3, seconds for the others.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import (AudioBufferSourceNode, ConvolverNode, GainNode, InvalidOperationException, OfflineAudioContext,
                            PlayableAudioBuffer)
from tests import _graphs as G
from tests._oracle import OracleContext
from tests._report import note

SR = 48000
TOL_RMS = 1e-5


def both(builder, nrender, options=None, **kw):
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(SR)
        if mk is OfflineAudioContext:
            for k, v in (options or {}).items():
                ctx.SetOption(k, v)
        ch = builder(ctx, **kw)
        outs.append(G.render(ctx, ch, nrender))
        ctx.Dispose()
    return outs


def report(name, ref, got):
    err, sig = G.rms(ref - got), G.rms(ref)
    note(f"[atsize] {name}: bus rms {sig:.4e}  abs rms err {err:.3e}  relative {err / sig:.3e}")
    return err, sig


_cfg3_ref = {}


def _config3_reference(frames):
    """the oracle's 1024-voice render (~45 s of host CPU) is shared by the formulations checked against it"""
    if frames not in _cfg3_ref:
        o = OracleContext(SR)
        G.config3_convolver(o, voices=1024, taps=65536, frames=frames)
        _cfg3_ref[frames] = G.render(o, 2, frames)
        o.Dispose()
    return _cfg3_ref[frames]


@pytest.mark.parametrize("formulation", ["coarse partitions (D, default)", "block-axis FFT (C)"])
def test_config3_all_1024_voices_short_form(formulation):
    frames = 375 * 128
    ref = _config3_reference(frames)
    h = OfflineAudioContext(SR)
    h.SetOption("coarse", 1 if formulation.startswith("coarse") else 0)
    G.config3_convolver(h, voices=1024, taps=65536, frames=frames)
    got = G.render(h, 2, frames)
    st = h.GetStats()
    h.Dispose()
    assert (st["stage_launches"][5] > 0) == formulation.startswith("coarse")
    err, sig = report(f"config 3, 1024 voices x 65,536-tap stereo IR, 375 blocks, {formulation}", ref, got)
    assert sig > 0.5            # an incoherent bus of 1024 voices (sigma ~ 2.6 in steady state, less while the tail builds up)
    assert err <= TOL_RMS, err
    assert err / sig < 2e-6


def test_config3_all_1024_voices_second_call_from_the_carried_tails():
    """the steady state of the benchmark's steps: the second render call of the 1024-voice graph starts from the output tails
    the first one left, not from the voices' input histories"""
    frames = 375 * 128
    ref = _config3_reference(frames)
    h = OfflineAudioContext(SR)
    G.config3_convolver(h, voices=1024, taps=65536, frames=frames)
    got = np.zeros((2, frames), np.float32)
    h.Render(got, 256 * 128, 0)
    h.Render(got, frames - 256 * 128, 256 * 128)
    st = h.GetStats()
    h.Dispose()
    assert st["coarse_carried_outputs"] == 2
    err, sig = report("config 3, 1024 voices, 256 + 119 blocks (second call from the carried tails)", ref, got)
    assert err <= TOL_RMS, err
    assert err / sig < 2e-6


@pytest.mark.parametrize("tail_private", [1, 0])
def test_config3_256_voices_with_private_impulse_responses_in_two_calls(tail_private):
    """the general multiply-accumulate kernel at scale (8 jobs of 32 terms, every term its own 65,536-tap stereo IR), rendered in
    two calls: the second one starts from the carried output tails (default since round 3: the kernel skips the partition blocks
    behind the chunk's end) or, with `coarse_tail_private = 0`, from the input histories"""
    frames = 375 * 128
    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(SR)
        if mk is OfflineAudioContext:
            ctx.SetOption("coarse_tail_private", tail_private)
        G.config3_convolver(ctx, voices=256, taps=65536, frames=frames, shared=False)
        out = np.zeros((2, frames), np.float32)
        ctx.Render(out, 260 * 128, 0)
        ctx.Render(out, frames - 260 * 128, 260 * 128)
        if mk is OfflineAudioContext:
            st = ctx.GetStats()
            assert st["stage_launches"][5] > 0 and st["coarse_carried_outputs"] == (2 if tail_private else 0)
        outs.append(out)
        ctx.Dispose()
    err, sig = report(f"config 3 variant, 256 voices x private 65,536-tap stereo IRs, 260 + 115 blocks, tails carried: {tail_private}", outs[0], outs[1])
    assert err <= TOL_RMS, err
    assert err / sig < 2e-6


_config5 = G.config5_ambisonic


@pytest.mark.parametrize("coarse", [1, 0])
def test_config5_16_channel_32768_tap_private_irs(coarse):
    frames = 400 * 128          # > P = 256 blocks: every partition of every column is active at the end
    ref, got = both(_config5, frames, options={"coarse": coarse}, sources=6, taps=32768, frames=frames)
    assert ref.shape[0] == 16
    err, sig = report(f"config 5, 6 sources x 16-channel 32,768-tap private IRs, 400 blocks, coarse={coarse}", ref, got)
    assert err <= TOL_RMS, err
    assert err / sig < 2e-6
    tail = slice(300 * 128, None)
    assert G.rms(ref[:, tail] - got[:, tail]) / G.rms(ref[:, tail]) < 2e-6


def test_config2_256_voices_bit_exact():
    frames = 375 * 128
    ref, got = both(G.config2_biquad, frames, options={"biquad_time_split": 0}, voices=256, frames=frames)   # (the one-walk evaluation)
    report("config 2, 256 voices -> lowpass -> gain -> mono mix, 375 blocks", ref, got)
    assert G.rms(ref) > 1e-3
    assert np.array_equal(ref, got)


def test_config4_512_voice_shard():
    frames = 375 * 128
    ref, got = both(G.config4_eq, frames, options={"biquad_time_split": 0}, voices=512, frames=frames)   # (split: tests/test_gpu_biquad_split.py)
    err, sig = report("config 4, 512 voices (resampler + 5-band EQ + gain automation), 375 blocks", ref, got)
    assert sig > 1e-4
    assert err <= 1e-6, err     # constant-coefficient cascades and the resampler are bit-exact; the gain curve is f64 on both sides


def test_more_than_65535_jobs_in_one_level():
    """4,200 voices, each starting in its own block of a 40-block chunk: the gain level of the chunk holds
    voices x segments jobs, far beyond gridDim.y = 65,535 (launch_mix / gain / loop_source windows)."""
    voices, blocks = 4200, 40
    frames = blocks * 128

    def build(ctx):
        ctx.Destination.SetChannelCount(1)
        bus = GainNode(ctx)
        bus.Inputs[0].SetChannelCount(1)
        bus.Gain.Value = 1.0 / 64
        bus.Connect(ctx.Destination)
        for v in range(voices):
            s = AudioBufferSourceNode(ctx)
            s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v % 64, 128 * 24), SR)
            g = GainNode(ctx)
            g.Inputs[0].SetChannelCount(1)
            g.Gain.Value = 0.5 + 0.001 * (v % 100)
            s.Connect(g).Connect(bus)
            s.Start(((v % 32) * 128 + 1) / SR)
        return 1

    outs = []
    for mk in (OracleContext, OfflineAudioContext):
        ctx = mk(SR)
        build(ctx)
        outs.append(G.render(ctx, 1, frames))
        if mk is OfflineAudioContext:
            st = ctx.GetStats()
            note(f"[atsize] many-jobs graph: {st['segments']} segments, {st['kernel_launches']} launches")
        ctx.Dispose()
    ref, got = outs
    assert G.rms(ref) > 1e-3
    assert np.array_equal(ref, got)


def test_unsupported_fft_length_is_an_error_code_and_faults_the_context():
    """A planner bug of the class fixed in round 1 (a block-axis FFT length without a kernel) must surface as an error code
    through the C ABI -- never abort() the host -- and, since control state has moved, leave the context faulted."""
    ctx = OfflineAudioContext(SR)
    ctx.SetOption("coarse", 0)
    G.config3_convolver(ctx, voices=2, taps=128 * 100, frames=128 * 64)
    ctx.SetOption("debug_tconv_n2", 512)
    out = np.zeros((2, 128 * 8), np.float32)
    with pytest.raises(InvalidOperationException):
        ctx.Render(out, 128 * 8)
    ctx.SetOption("debug_tconv_n2", 0)
    with pytest.raises(InvalidOperationException, match="faulted"):
        ctx.Render(out, 128 * 8)
    ctx.Dispose()
    # the process and the device are fine: a new context renders
    ctx = OfflineAudioContext(SR)
    G.config3_convolver(ctx, voices=2, taps=128 * 100, frames=128 * 64)
    assert G.rms(G.render(ctx, 2, 128 * 8)) > 0
    ctx.Dispose()
