#!/usr/bin/env python3
"""Where the cycles of biquad_pipe_kernel go INSIDE the engine (config 4 at full size): a library built with
`VARIANT_KERNELS=1 tools/build_variant.sh bqprobe -DGA_BQ_PROBE` sums s_memtime ticks per phase (ga_kernels.hip, GA_BQ_PROBE)."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphaudio_amd import OfflineAudioContext, _capi
from tests import _graphs as G
lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "variants", "bqprobe.so")
_capi.use_library(lib)
voices = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
frames = 120000 * 4
ctx = OfflineAudioContext(48000)
ch = G.config4_eq(ctx, voices=voices, frames=frames)
out = np.zeros((ch, frames), np.float32)
h = ctypes.CDLL(lib)
pr = (ctypes.c_ulonglong * 8)()
for rep in range(4):
    t0 = time.time()
    ctx.Render(out, 120000, rep * 120000)
    dt = time.time() - t0
    h.ga_bq_probe_read(pr)
    w = max(pr[6], 1)
    print(f"piece {rep}: {dt * 1e3:.1f} ms | waves {pr[6]} cascades {pr[7]} | per wave ticks: total {pr[3] / w:.0f} stage-in {pr[0] / w:.0f} walk {pr[1] / w:.0f} "
          f"stage-out {pr[2] / w:.0f} | batches per wave: steady {pr[4] / w:.0f} masked {pr[5] / w:.0f}")
