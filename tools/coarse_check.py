#!/usr/bin/env python3
"""Quick look at formulation D (coarse partitions) against the CPU oracle: python tools/coarse_check.py [voices taps blocks]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
from tests._oracle import OracleContext

voices = int(sys.argv[1]) if len(sys.argv) > 1 else 3
taps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 300
frames = blocks * 128
o = OracleContext(48000)
G.config3_convolver(o, voices=voices, taps=taps, frames=frames)
ref = G.render(o, 2, frames)
for coarse in (0, 1):
    h = OfflineAudioContext(48000)
    h.SetOption("coarse", coarse)
    h.SetOption("coarse_min_blocks", 1)
    h.SetOption("profile", 1)
    G.config3_convolver(h, voices=voices, taps=taps, frames=frames)
    got = G.render(h, 2, frames)
    st = h.GetStats()
    err = G.rms(ref - got)
    print(f"coarse={coarse}: bus rms {G.rms(ref):.4e} err {err:.3e} rel {err / G.rms(ref):.3e} max {np.abs(ref - got).max():.3e}",
          {k: round(v, 3) for k, v in zip(st['stage_ms'][:9], st['stage_ms'][:9])} and [round(x, 3) for x in st["stage_ms"][:9]])
    if err / G.rms(ref) > 1e-4:
        d = np.abs(ref - got)
        bad = np.argwhere(d > 1e-3 * G.rms(ref))
        print("  first bad", bad[:5].tolist(), "count", len(bad), "of", d.size)
        for t in range(0, frames, 8192):
            print(f"   coarse block {t // 8192}: err {G.rms(ref[:, t:t + 8192] - got[:, t:t + 8192]):.3e} ref {G.rms(ref[:, t:t + 8192]):.3e}")
    h.Dispose()
