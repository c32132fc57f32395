import sys, numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext
import graphaudio_amd.core as core
from tests import _graphs as G
from tests._fuzz import build_random_graph
from tests._oracle import OracleContext
frames = 128 * 36
log = []
def wrap(cls, name):
    orig = getattr(cls, name)
    def f(self, *a, **k):
        log.append((cls.__name__, getattr(self, "_id", None), name, [getattr(x, "_id", x) if not isinstance(x, core.PlayableAudioBuffer) else f"buf{x.NumberOfChannels}x{x.Length}@{x.SampleRate}" for x in a]))
        return orig(self, *a, **k)
    setattr(cls, name, f)
for seed in [int(x) for x in sys.argv[1:]]:
    o = OracleContext(48000); ch = build_random_graph(o, seed, frames)
    ref = np.zeros((ch, frames), np.float32); o.Render(ref, frames)
    res = {}
    for mode in ("oneshot", "chunk11", "pieces"):
        h = OfflineAudioContext(48000)
        if mode != "oneshot": h.SetOption("max_chunk_blocks", 11)
        build_random_graph(h, seed, frames)
        got = np.zeros_like(ref)
        if mode == "pieces":
            pos = 0; rng = np.random.default_rng(1000 + seed)
            while pos < frames:
                n = int(min(frames - pos, rng.integers(1, 128 * 9))); h.Render(got, n, pos); pos += n
        else:
            h.Render(got, frames)
        d = np.abs(ref - got).max(axis=0)
        badf = np.nonzero(d > 1e-5)[0]
        res[mode] = (G.rms(ref - got), (int(badf[0]) // 128, int(badf[0]) % 128, len(badf)) if len(badf) else None, h.GetStats()["segments"])
    print("seed", seed, "ch", ch, res)
