"""Which voice of a random edit session carries a deviation: replays the session with one voice connected at a time
(the minimiser hook of tests/_fuzz.py) on the oracle and on the device, and prints each voice's chain."""
import sys, numpy as np
sys.path.insert(0, ".")
import graphaudio_amd as ga
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
import tests._fuzz as F
from tests._oracle import OracleContext

seed = int(sys.argv[1])
edges = []
orig = ga.AudioNode.Connect
def rec(self, target, *a, **k):
    edges.append((type(self).__name__, getattr(self, "_id", None), type(target).__name__, getattr(target, "_id", None), a))
    return orig(self, target, *a, **k)
ga.AudioNode.Connect = rec
o = OracleContext(48000); ref, rl = F.run_random_session(o, seed)
ga.AudioNode.Connect = orig
for e in edges: print("  edge", e)
print("details", F.details)
print("pieces", F.last_pieces)
nv = sum(1 for e in edges if e[0] == "AudioBufferSourceNode")
for keep in [None, set()] + [{v} for v in range(12)]:
    try:
        o = OracleContext(48000); ref, rl = F.run_random_session(o, seed, keep=keep)
        h = OfflineAudioContext(48000)
        import os
        for kv in os.environ.get("GA_OPTS", "").split(","):
            if "=" in kv:
                h.SetOption(kv.split("=")[0], float(kv.split("=")[1]))
        got, gl = F.run_random_session(h, seed, keep=keep)
    except Exception as e:
        print(keep, "exception", type(e).__name__, e); continue
    d = np.abs(ref - got).max(axis=0); bf = np.nonzero(d > 2e-5)[0]
    print("keep", keep, "err %.3e scale %.3f" % (G.rms(ref - got), G.rms(ref)), "first bad frame", int(bf[0]) if len(bf) else None)
