import sys, numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
import tests._fuzz as F
from tests._oracle import OracleContext
for seed in (2573, 5761):
    o = OracleContext(48000); ref, rl = F.run_random_session(o, seed)
    for name, opts in (("default coarse forced", {"coarse_min_blocks": 1}), ("no coarse", {"coarse_min_blocks": 1 << 30}), ("coarse, no premix/ext/split", {"coarse_min_blocks": 1, "coarse_premix": 0, "coarse_ext_history": 0, "biquad_time_split": 0, "coarse_wide": 0}), ("coarse no tail", {"coarse_min_blocks": 1, "coarse_tail": 0})):
        h = OfflineAudioContext(48000); h.SetOption("max_chunk_blocks", 11)
        for k, v in opts.items(): h.SetOption(k, v)
        got, gl = F.run_random_session(h, seed)
        err = G.rms(ref - got); sc = G.rms(ref)
        d = np.abs(ref - got).max(axis=0); bf = np.nonzero(d > 2e-5)[0]
        print(seed, name, "err %.3e scale %.3f rel %.2e" % (err, sc, err / sc), "first bad block", (int(bf[0]) // 128, len(bf)) if len(bf) else None, "log same", rl == gl)
    print("  log", rl[:12])
for seed, b0 in ((2573, 30), (5761, 18)):
    o = OracleContext(48000); ref, rl = F.run_random_session(o, seed)
    print(seed, [e for e in rl if b0 - 6 <= e[0] <= b0 + 1])
    h = OfflineAudioContext(48000); got, gl = F.run_random_session(h, seed)
    d = np.abs(ref - got)
    for b in range(b0 - 1, min(b0 + 12, ref.shape[1] // 128)):
        print("   block", b, "max diff %.2e" % d[:, b*128:(b+1)*128].max(), "ref max %.3f" % np.abs(ref[:, b*128:(b+1)*128]).max())
