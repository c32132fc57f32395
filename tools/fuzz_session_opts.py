"""tools/fuzz_session_opts.py SEED: one random edit session under the option sets of fuzz mode 3 and neighbours"""
import sys, numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
import tests._fuzz as F
from tests._oracle import OracleContext
seed = int(sys.argv[1])
o = OracleContext(48000); ref, rl = F.run_random_session(o, seed)
base = {"max_chunk_blocks": 11, "coarse_min_blocks": 1, "coarse_premix": 0}
for name, extra in (("mode 3", {}), ("+ premix on (mode 1)", {"coarse_premix": 1}), ("no tails", {"coarse_tail": 0}), ("no ext history", {"coarse_ext_history": 0}),
                    ("no gain fold/pass", {"gain_fold": 0, "gain_pass_through": 0}), ("no wide/mfma", {"coarse_wide": 0}), ("one chunk", {"max_chunk_blocks": 4096}),
                    ("no D", {"coarse_min_blocks": 1 << 30})):
    h = OfflineAudioContext(48000)
    for k, v in {**base, **extra}.items():
        h.SetOption(k, v)
    got, gl = F.run_random_session(h, seed)
    d = np.abs(ref - got).max(axis=0); bad = np.nonzero(d > 1e-4)[0]
    print(f"{name:24s} err {G.rms(ref - got):.3e} scale {G.rms(ref):.3f} first bad", (int(bad[0]), int(bad[0]) // 128, len(bad)) if len(bad) else None, "log same", rl == gl)
print("pieces", F.last_pieces)
print("details", [(d[0], d[1]) for d in F.details])
