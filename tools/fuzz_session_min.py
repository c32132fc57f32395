import sys, numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
import tests._fuzz as F
from tests._oracle import OracleContext
def err(seed, keep):
    o = OracleContext(48000); ref, rl = F.run_random_session(o, seed, keep=keep)
    h = OfflineAudioContext(48000); got, gl = F.run_random_session(h, seed, keep=keep)
    d = np.abs(ref - got).max(axis=0); bf = np.nonzero(d > 1e-5)[0]
    return G.rms(ref - got), (int(bf[0]) // 128, int(bf[0]) % 128, len(bf)) if len(bf) else None
for seed in [int(x) for x in sys.argv[1:]]:
    keep = set(range(10))
    print("seed", seed, "all", err(seed, keep))
    for v in range(10):
        if err(seed, keep - {v})[1] is not None: keep.discard(v)
    print("  minimal voices", sorted(keep), err(seed, keep))
