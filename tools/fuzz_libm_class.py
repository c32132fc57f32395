"""tools/fuzz_libm_class.py SEED...: is a session's deviation from the oracle the last bit of cosf / sinf / powf?

The reference's MathF.Cos / Sin / Pow are the C runtime's single-precision functions (the oracle: glibc's); the device evaluates the
coefficients of AUTOMATED biquads and the gains of panners in double and rounds once (ga_kernels.hip, biquad_update_coefficients) --
the same float except where the true value lies within ~1e-9 of a rounding boundary.  A graph that quantises such a value (a delay
time in whole samples, `pan != lastPan`) turns that last bit into a block that sounds different.  This tool renders the session
three times: oracle, device, and a DIAGNOSTIC oracle built with -DGAO_DOUBLE_TRIG (the device's evaluation of those three functions,
everything else the reference's): a deviation that vanishes against the third is that class and nothing else.
    g++ -std=c++17 -O3 -mavx2 -ffp-contract=off -fno-fast-math -fPIC -DGAO_DOUBLE_TRIG -shared -o tools/variants/libga_oracle_dtrig.so oracle/ga_oracle.cpp
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext
from graphaudio_amd._capi import CApi
from tests import _graphs as G
import tests._fuzz as F
from tests._oracle import OracleContext
dtrig = CApi(C.CDLL(os.path.join("tools", "variants", "libga_oracle_dtrig.so")), "gao_")
def run(ctx, seed, graph):
    if not graph:
        return F.run_random_session(ctx, seed)[0]
    frames = 128 * 36
    ch = F.build_random_graph(ctx, seed, frames)
    out = np.zeros((ch, frames), np.float32)
    pos = 0
    rng = np.random.default_rng(1000 + seed)
    while pos < frames:   # (the pieces of tests/test_gpu_fuzz.py::test_random_graph_matches_oracle)
        n = int(min(frames - pos, rng.integers(1, 128 * 9)))
        ctx.Render(out, n, pos)
        pos += n
    return out


for a in sys.argv[1:]:   # SEED: an edit session ; gSEED: a graph of test_random_graph_matches_oracle
    graph = a.startswith("g")
    seed = int(a[1:] if graph else a)
    ref = run(OracleContext(48000), seed, graph)
    ref2 = run(OfflineAudioContext(48000, _api=dtrig), seed, graph)
    h = OfflineAudioContext(48000)
    h.SetOption("max_chunk_blocks", 11)
    h.SetOption("coarse_min_blocks", 1)
    got = run(h, seed, graph)
    scale = max(G.rms(ref), 1e-3)
    d = np.abs(ref - got).max(axis=0).reshape(-1, 128).max(axis=1)
    for b in np.nonzero(d > 1e-4)[0][:6]:   # a block that is exact silence on one side only: a silence FLAG differs (DelayNode)
        rb, gb = ref[:, b * 128:(b + 1) * 128], got[:, b * 128:(b + 1) * 128]
        print(f"      block {b}: oracle rms {G.rms(rb):.4g} ({int((rb != 0).sum())} non-zero samples)  device rms {G.rms(gb):.4g} ({int((gb != 0).sum())} non-zero)")
    print(f"{'graph' if graph else 'session'} {seed}: device vs oracle {G.rms(ref - got):.3e}   device vs oracle with the device's cos/sin/pow {G.rms(ref2 - got):.3e}   "
          f"oracle vs that oracle {G.rms(ref - ref2):.3e}   (signal {scale:.3f}; blocks off by > 1e-4: {np.nonzero(d > 1e-4)[0][:8].tolist()})", flush=True)
