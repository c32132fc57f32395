#!/usr/bin/env python3
"""Steady-state parity of the convolver at the full 65,536-tap size (all 512 partitions active).

Renders a few voices of BASELINE.json config 3 on the HIP path and on the CPU oracle (needs a GPU; ~1 min of CPU) and
prints the per-voice relative error plus the bus error extrapolated to V voices (errors of independent voices add
incoherently: err_bus ~ eps_rel * sigma_voice * sqrt(V))."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
from tests._oracle import OracleContext

voices = int(sys.argv[1]) if len(sys.argv) > 1 else 4
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 700
taps = 65536
frames = blocks * 128
outs = []
for mk in (OracleContext, OfflineAudioContext):
    ctx = mk(48000)
    t0 = time.time()
    ch = G.config3_convolver(ctx, voices=voices, taps=taps, frames=frames)
    outs.append(G.render(ctx, ch, frames))
    print(mk.__name__, f"{time.time() - t0:.1f} s")
ref, got = outs
tail = slice(520 * 128, None)  # all partitions populated
err = G.rms(ref[:, tail] - got[:, tail])
sig = G.rms(ref[:, tail])
print(f"voices={voices} bus sigma={sig:.4e} abs rms err={err:.3e} rel={err / sig:.3e}")
sigma_voice = sig / np.sqrt(voices)
eps = err / sig
print(f"extrapolated abs rms error at 1024 voices: {eps * sigma_voice * np.sqrt(1024):.3e} (tolerance 1e-5)")
