"""Sweep random edit sessions (tests/_fuzz.py::run_random_session) over a seed range on the GPU."""
import sys, traceback, numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext, NotSupportedException
from tests import _graphs as G
from tests._fuzz import run_random_session
from tests._oracle import OracleContext
import os
ASYNC = os.environ.get("GA_FUZZ_ASYNC") == "1"
lo, hi = int(sys.argv[1]), int(sys.argv[2])
bad = []; skipped = 0
for seed in range(lo, hi):
    try:
        o = OracleContext(48000); ref, rl = run_random_session(o, seed)
    except Exception as e:
        print("oracle raised", seed, type(e).__name__, e); continue
    try:
        h = OfflineAudioContext(48000); h.SetOption("max_chunk_blocks", 11)
        if ASYNC: h.SetOption("async", 1)   # GA_FUZZ_ASYNC=1: pipelined renders, synchronised once at the end
        got, gl = run_random_session(h, seed)
        if ASYNC: h.Synchronize()
    except NotSupportedException as e:
        skipped += 1; continue
    except Exception as e:
        bad.append((seed, "raised " + type(e).__name__ + " " + str(e)[:100])); continue
    if rl != gl:
        bad.append((seed, "log differs")); continue
    err = G.rms(ref - got); scale = max(G.rms(ref), 1e-3)
    if not (err <= 1e-5 or err <= 5e-5 * scale):   # large-amplitude transients of automated biquads: device sinf/cosf vs glibc
        d = np.abs(ref - got).max(axis=0); bf = np.nonzero(d > 1e-5)[0]
        bad.append((seed, err, scale, (int(bf[0]) // 128, int(bf[0]) % 128, len(bf)) if len(bf) else None))
print("bad", bad); print("skipped", skipped, "of", hi - lo)
