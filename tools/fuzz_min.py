"""Minimise a failing fuzz seed to the smallest set of voices that still fails (GPU)."""
import sys, itertools, numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
from tests._fuzz import build_random_graph
from tests._oracle import OracleContext
frames = 128 * 36
def err(seed, keep, chunk=0):
    o = OracleContext(48000); ch = build_random_graph(o, seed, frames, keep)
    ref = np.zeros((ch, frames), np.float32); o.Render(ref, frames)
    h = OfflineAudioContext(48000); (h.SetOption("max_chunk_blocks", chunk) if chunk else None); build_random_graph(h, seed, frames, keep)
    got = np.zeros_like(ref); h.Render(got, frames)
    d = np.abs(ref - got).max(axis=0); badf = np.nonzero(d > float(__import__('os').environ.get('THR','1e-5')))[0]
    return G.rms(ref - got), (int(badf[0]) // 128, int(badf[0]) % 128, len(badf)) if len(badf) else None
for seed in [int(x) for x in sys.argv[1:]]:
    keep = set(range(10))
    print("seed", seed, "all", err(seed, keep))
    for v in range(10):
        if err(seed, keep - {v})[1] is not None: keep.discard(v)
    print("  minimal", sorted(keep), err(seed, keep), "tf0" )
