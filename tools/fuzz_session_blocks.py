"""tools/fuzz_session_blocks.py SEED [VOICE]: per-block deviation of a random edit session with formulation D forced"""
import sys, numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
import tests._fuzz as F
from tests._oracle import OracleContext
seed = int(sys.argv[1]); keep = {int(sys.argv[2])} if len(sys.argv) > 2 else None
if len(sys.argv) > 3:   # a variant build (tools/build_variant.sh), e.g. with GA_DEBUG_D=1 for the D stage's plan on stderr
    from graphaudio_amd import _capi
    _capi.use_library(sys.argv[3])
o = OracleContext(48000); ref, rl = F.run_random_session(o, seed, keep=keep)
print("pieces", F.last_pieces); print("details", [(d[0], d[1], round(d[3] * 48000)) for d in F.details])
for name, opts in (("D forced", {"coarse_min_blocks": 1}),) + ((("C (no D)", {"coarse_min_blocks": 1 << 30}),) if len(sys.argv) <= 3 else ()):
    h = OfflineAudioContext(48000); h.SetOption("max_chunk_blocks", 11)
    for k, v in opts.items(): h.SetOption(k, v)
    got, gl = F.run_random_session(h, seed, keep=keep)
    d = np.abs(ref - got)
    print(name, "err %.3e" % G.rms(ref - got), "stats", {k: h.GetStats()[k] for k in ("chunks", "segments", "coarse_carried_outputs", "coarse_premixed_signals")})
    for b in range(ref.shape[1] // 128):
        m = d[:, b * 128:(b + 1) * 128].max()
        if m > 1e-5:
            ch = int(d[:, b * 128:(b + 1) * 128].max(axis=1).argmax())
            seg = slice(b * 128, (b + 1) * 128)
            print("   block %2d ch %d max diff %.3e  ref rms %.4f got rms %.4f  rms(got-ref) %.4f" % (b, ch, m, G.rms(ref[ch, seg]), G.rms(got[ch, seg]), G.rms(got[ch, seg] - ref[ch, seg])))
