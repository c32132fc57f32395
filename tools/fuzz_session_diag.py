import sys, numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
import tests._fuzz as F
from tests._oracle import OracleContext
for seed in [int(x) for x in sys.argv[1:]]:
    o = OracleContext(48000); ref, rl = F.run_random_session(o, seed)
    res = {}
    for chunk in (0, 11):
        h = OfflineAudioContext(48000)
        if chunk: h.SetOption("max_chunk_blocks", chunk)
        got, gl = F.run_random_session(h, seed)
        d = np.abs(ref - got).max(axis=0); bf = np.nonzero(d > 1e-5)[0]
        res[chunk] = (G.rms(ref - got), (int(bf[0]) // 128, int(bf[0]) % 128, len(bf)) if len(bf) else None)
    # piece boundaries (same rng replay)
    rng = np.random.default_rng(seed ^ 0x5EED)
    print("seed", seed, res)
    print("   log", F.details)
    print("   pieces(frames end)", getattr(F, "last_pieces", None))
