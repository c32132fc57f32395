#!/usr/bin/env python3
"""Sweep of the carried-tail / pre-mix edit sessions of tests/test_gpu_tail_fuzz.py over a seed range: tools/tail_session_sweep.py FIRST LAST"""
import sys, time
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
from tests._oracle import OracleContext
from tests.test_gpu_tail_fuzz import _session, SR
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = []
t0 = time.time()
for seed in range(first, last):
    for ragged in (False, True):
        blocks = 500 if ragged else 700
        o = OracleContext(SR)
        ref, log = _session(o, seed, blocks, ragged=ragged)
        o.Dispose()
        for premix in (1, 0):
            h = OfflineAudioContext(SR)
            h.SetOption("coarse_min_blocks", 1)
            h.SetOption("coarse_premix", premix)
            got, log2 = _session(h, seed, blocks, ragged=ragged)
            h.Dispose()
            err, sig = G.rms(ref - got), G.rms(ref)
            if log != log2 or not (err <= 1e-5 and err <= 2e-6 * sig):
                bad.append((seed, ragged, premix, err, sig))
                print("FAIL", bad[-1], flush=True)
    if seed % 20 == 0:
        print(f"seed {seed}  {time.time() - t0:.0f} s  failures {len(bad)}", flush=True)
print("done", last - first, "seeds,", len(bad), "failures", bad[:5])
