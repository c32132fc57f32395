"""ga_stats bookkeeping of the per-stage profile: sampled HIP events (options "profile", "profile_every"), launches and bytes of
every chunk."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G


def test_sampled_stage_profile():
    ctx = OfflineAudioContext(48000)
    ctx.SetOption("profile", 1)
    ctx.SetOption("profile_every", 3)
    ctx.SetOption("coarse_min_blocks", 1)
    frames = 128 * 40
    G.config3_convolver(ctx, voices=8, taps=20000, frames=frames * 7)
    out = np.zeros((2, frames), np.float32)
    for _ in range(7):
        ctx.Render(out, frames)
    st = ctx.GetStats()
    ctx.Dispose()
    assert st["chunks"] == 7
    assert st["profiled_chunks"] == 3                      # chunks 0, 3 and 6 recorded their events
    assert st["stage_launches"][5] == 7                    # GA_STAGE_COARSE_FWD: every chunk counts its launches ...
    assert st["stage_bytes"][5] > 0
    assert st["stage_ms"][5] > 0 and st["device_ms_total"] > 0   # ... the times come from the sampled ones
    assert st["coarse_carried_outputs"] == 2 * 6          # two bus channels, every chunk after the first


def test_no_profile_no_times():
    ctx = OfflineAudioContext(48000)
    frames = 128 * 20
    G.config3_convolver(ctx, voices=2, taps=3000, frames=frames)
    ctx.Render(np.zeros((2, frames), np.float32), frames)
    st = ctx.GetStats()
    ctx.Dispose()
    assert st["profiled_chunks"] == 0 and st["device_ms_total"] == 0
    assert st["kernel_launches"] > 0
