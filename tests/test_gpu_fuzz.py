"""Randomised graphs: HIP path vs CPU oracle (control plane: channel counts, silence, scheduling, dispose; data plane:
every node type).  The render is split into uneven pieces on the HIP side to exercise chunk / state carry-over."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import NotSupportedException, OfflineAudioContext
from tests import _graphs as G
from tests._fuzz import build_random_graph
from tests._oracle import OracleContext


# seeds that once failed (resampler end-of-input block, loop offset beyond the loop end, shared-IR rows at two convolver
# depths) stay in the list; 1186/1763/3027/3639/4787 are the worst numeric cases of a 6000-seed sweep (a convolver in
# front of a low-frequency biquad, whose f32 direct-form recursion amplifies 1e-7 input differences ~100x)
REGRESSION_SEEDS = [215, 219, 357, 430, 459, 749, 1232, 1273, 1476, 2550, 2577, 2916, 3371, 1186, 1763, 3027, 3639, 4787,
                    # round 3 (connections between voice chains): a delay fed by a convolver that has not seen input yet (flagged non-silent,
                    # exact zeros) is one of two connections of a biquad whose other source ended -- the delay's output flag
                    20256, 22316,
                    # a panner whose pan modulation falls silent inside a chunk (seeds >= 30000: chain signals into parameters of later chains)
                    30941,
                    # round 4: documented above the 1e-5 contract in round 3 (2.6e-5: a peaking filter at 104 Hz, Q 2.1, behind a convolver;
                    # 5.7e-5: a DelayNode's delay time modulated by a convolver's output) -- the planner now evaluates such convolvers in
                    # the reference's own order (formulation R, tests/test_gpu_reforder.py); 23930 / 24413 / 24797 / 11679: the same class at 2e-6 .. 6e-6
                    22879, 40542, 23930, 24413, 24797, 11679] + list(range(30000, 30012)) + list(range(40000, 40010))   # (>= 40000: splitter outputs / merger inputs cross-connected by index)


# coarse = 1: convolvers with more than 64 partitions are forced onto formulation D (coarse partitions) even though the pieces are
# short -- every piece re-transforms the input history, which is exactly the state handling to stress
# 2: formulation D without carried output tails (input histories only); 3: D without the time-domain pre-mix of shared-IR groups
# 4: D with carried tails also for groups of private impulse responses (option coarse_tail_private, off by default)
@pytest.mark.parametrize("coarse", [0, 1, 2, 3, 4])
# seeds >= 20000: graphs with GainNodes at exactly 1 (buses and chain gains: their input views are handed on, no kernel)
# seeds >= 50000 (round 4): feedback loops inside the voice chains (rendered one block per chunk, the stale block of the loop's producer kept)
@pytest.mark.parametrize("seed", list(range(120)) + REGRESSION_SEEDS + list(range(20000, 20024)) + list(range(50000, 50016)))
def test_random_graph_matches_oracle(seed, coarse):
    frames = 128 * 36
    o = OracleContext(48000)
    ch = build_random_graph(o, seed, frames)
    ref = np.zeros((ch, frames), np.float32)
    try:
        o.Render(ref, frames)
    except Exception as e:  # e.g. destination narrower than requested: must fail the same way on the device
        h = OfflineAudioContext(48000)
        build_random_graph(h, seed, frames)
        with pytest.raises(type(e)):
            h.Render(np.zeros((ch, frames), np.float32), frames)
        return
    h = OfflineAudioContext(48000)
    h.SetOption("max_chunk_blocks", 11)
    h.SetOption("coarse_min_blocks", 1 if coarse else 1 << 30)
    h.SetOption("coarse_tail", 0 if coarse == 2 else 1)
    h.SetOption("coarse_premix", 0 if coarse == 3 else 1)
    h.SetOption("coarse_tail_private", 1 if coarse == 4 else 0)
    build_random_graph(h, seed, frames)
    got = np.zeros_like(ref)
    pos = 0
    rng = np.random.default_rng(1000 + seed)
    try:
        while pos < frames:
            n = int(min(frames - pos, rng.integers(1, 128 * 9)))
            h.Render(got, n, pos)
            pos += n
    except NotSupportedException as e:
        pytest.skip(f"graph uses a feature outside the device path: {e}")
    assert o.CurrentBlock == h.CurrentBlock or pos == frames
    if seed >= 50000 and (not np.isfinite(ref).all() or G.rms(ref) > 50.0):
        pytest.skip("a feedback loop with a gain above 1: the reference's output is not finite, or grows without bound and every last-bit difference with it")
    err = G.rms(ref - got)
    scale = max(G.rms(ref), 1e-3)
    # north_star: <= 1e-5 RMS per sample (seeds >= 50000: feedback loops may grow without bound -- the bound is taken at the signal's level)
    assert err <= 1e-5 * max(1.0, scale if seed >= 50000 else 1.0) and err <= 2e-5 * scale, (seed, err, scale)


def _session_pair(seed, chunk=11, coarse=0):
    from tests._fuzz import run_random_session
    o = OracleContext(48000)
    ref, ref_log = run_random_session(o, seed)
    h = OfflineAudioContext(48000)
    h.SetOption("max_chunk_blocks", chunk)
    h.SetOption("coarse_min_blocks", 1 if coarse else 1 << 30)
    h.SetOption("coarse_tail", 0 if coarse == 2 else 1)
    h.SetOption("coarse_premix", 0 if coarse == 3 else 1)
    h.SetOption("coarse_tail_private", 1 if coarse == 4 else 0)
    got, got_log = run_random_session(h, seed)
    return ref, ref_log, got, got_log


# 2850: a shared-IR convolver taken out of the graph for 17 blocks and plugged back (its delay line has to freeze)
@pytest.mark.parametrize("coarse", [0, 1, 2, 3, 4])   # 2: D without carried tails; 3: D without the time-domain pre-mix; 4: tails for private IRs too
# 2573: a ramp on a biquad's frequency -- the per-block coefficients are evaluated on the device, where cos / sin / pow have to be
#       rounded once from double like the C library's cosf / sinf / powf behind MathF (7.9e-6 -> 2.7e-9)
# 25085 (3.2e-5 in round 3: a biquad fed by a convolver and a source) and 5761 (a notch at 153 Hz, Q 2.5, behind a convolver: 9.4e-6):
#       convolvers in front of resonant biquads take the reference-order route (formulation R) since round 4
@pytest.mark.parametrize("seed", list(range(60)) + [2850, 2573, 20284, 25085, 5761] + list(range(20000, 20012)) + list(range(30000, 30006)) + list(range(40000, 40006)) + list(range(50000, 50010)) + [50178, 60001, 61173, 70427, 64064])   # 50178: a convolver inside a feedback loop (formulation R); 60001, 61173, 70427: an edit moves the entry of a loop away and back; 64064: a loop of convolvers nothing has reached yet carries exact zeros
def test_random_edit_session_matches_oracle(seed, coarse):
    """The graph is edited between render pieces (parameter writes, automation, stop, new voices, dispose, rewiring,
    impulse-response swaps, audio-rate modulation, channel settings): same output and the same exceptions."""
    try:
        ref, ref_log, got, got_log = _session_pair(seed, coarse=coarse)
    except NotSupportedException as e:
        pytest.skip(f"session uses a feature outside the device path: {e}")
    assert ref_log == got_log
    if not np.isfinite(ref).all() or G.rms(ref) > 50.0:
        pytest.skip("a feedback loop with gain above one: the reference's own output overflows or grows without bound, and every last-bit difference with it")
    err = G.rms(ref - got)
    scale = max(G.rms(ref), 1e-3)
    # (seeds >= 50000 have feedback loops; an edit can push a loop's gain above 1 and the signal grows without bound -- the absolute
    # bound is then taken at the signal's level)
    assert err <= 1e-5 * max(1.0, scale if seed >= 50000 else 1.0) and err <= 2e-5 * scale, (seed, err, scale)


def test_session_42867_with_formulation_d_forced():
    """The open defect of round 3 (3e-2 RMS in blocks 11-14 of one chunk).  Cause: a true-stereo convolver whose source has
    not started yet modulates a panner's pan through a depth gain; the reference's convolver puts out exact zeros until its
    first input block, so `pan != _lastPan` (StereoPannerNode.cs:92-99) never fires and the panner keeps the stereo-law gains of
    its first block.  Formulation D left ~1e-8 of circular rounding in front of the onset inside the same 16,384-point window:
    the pan 'changed' at the first sample where that survived the float addition (frame 131 of the chunk) and the gains were
    re-derived with the mono law.  Fixed in the planner: a convolver nothing has reached yet hands out the zero page
    (Context::chunkPlanNodes); deterministic cases in tests/test_gpu_coarse.py."""
    ref, ref_log, got, got_log = _session_pair(42867, coarse=1)
    assert ref_log == got_log
    assert G.rms(ref - got) <= 1e-5
