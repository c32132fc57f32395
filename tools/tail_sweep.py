import sys, numpy as np
sys.path.insert(0, '.')
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
from tests._oracle import OracleContext
from tests.test_gpu_tail_fuzz import _session, SR
worst = (0, -1)
carried = 0
for seed in range(24, 140):
    o = OracleContext(SR); ref, log = _session(o, seed, 500); o.Dispose()
    h = OfflineAudioContext(SR); h.SetOption("coarse_min_blocks", 1); got, log2 = _session(h, seed, 500)
    st = h.GetStats(); h.Dispose()
    err, sig = G.rms(ref - got), G.rms(ref)
    carried += st["coarse_carried_outputs"]
    rel = err / max(sig, 1e-9)
    if rel > worst[0]: worst = (rel, seed)
    if not (err <= 1e-5 and err <= 2e-6 * sig) or log != log2:
        print("FAIL", seed, err, sig, log); break
else:
    print("sweep ok; worst relative error", worst, "carried outputs", carried)
