"""tools/fuzz_node_dump.py SEED KEEP(comma list) BLOCK GOT.npy: which node's signal a deviation is made of.
The device's output of the session (GOT.npy, saved on the GPU box) minus the oracle's is projected on every output channel of every
node of the ORACLE in that block (a diagnostic oracle build with -DGAO_DEBUG_DUMP writes them): an error that is +-1 x some node's
output says which signal the device dropped, added or delayed.
    g++ -std=c++17 -O2 -mavx2 -ffp-contract=off -fno-fast-math -fPIC -DGAO_DEBUG_DUMP -shared -o tools/variants/libga_oracle_dump.so oracle/ga_oracle.cpp"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext
from graphaudio_amd._capi import CApi
import tests._fuzz as F
graph = sys.argv[1].startswith("g")   # gSEED: a graph of test_random_graph_matches_oracle
seed = int(sys.argv[1][1:] if graph else sys.argv[1]); keep = {int(x) for x in sys.argv[2].split(",")}; block = int(sys.argv[3]); got = np.load(sys.argv[4])
dump = "/tmp/gao_dump.txt"
if os.path.exists(dump): os.remove(dump)
os.environ["GAO_DUMP_FROM"] = str(block - 1); os.environ["GAO_DUMP_TO"] = str(block); os.environ["GAO_DUMP_FILE"] = dump
api = CApi(C.CDLL(os.path.join("tools", "variants", "libga_oracle_dump.so")), "gao_")
if graph:
    frames = 128 * 36
    octx = OfflineAudioContext(48000, _api=api)
    ch = F.build_random_graph(octx, seed, frames, keep=keep)
    ref = np.zeros((ch, frames), np.float32)
    pos = 0
    rng = np.random.default_rng(1000 + seed)
    while pos < frames:
        n = int(min(frames - pos, rng.integers(1, 128 * 9)))
        octx.Render(ref, n, pos)
        pos += n
else:
    ref, _ = F.run_random_session(OfflineAudioContext(48000, _api=api), seed, keep=keep)
err = (got - ref)[:, block * 128:(block + 1) * 128]
print("error per channel (max abs)", np.abs(err).max(axis=1))
rows = {}
lines = open(dump).read().splitlines()
i = 0
while i < len(lines):
    h = lines[i].split()
    b, node, typ, proc, out, ch, silent = int(h[1]), int(h[3]), int(h[5]), int(h[7]), int(h[9]), int(h[11]), int(h[13])
    for c in range(ch):
        rows[(b, node, typ, out, c, proc, silent)] = np.array(lines[i + 1 + c].split(), dtype=np.float64)
    i += 1 + ch
for ech in range(err.shape[0]):
    e = err[ech].astype(np.float64)
    if np.abs(e).max() < 1e-6: continue
    best = []
    for key, v in rows.items():
        nv = float(v @ v)
        if nv < 1e-12: continue
        a = float(e @ v) / nv                 # least-squares factor
        res = float(np.linalg.norm(e - a * v) / np.linalg.norm(e))
        best.append((res, a, key))
    best.sort(key=lambda t: t[0])
    print(f"error channel {ech}: best single-signal explanations (residual, factor, (block, node, type, output, channel, processed, silent))")
    for t in best[:6]: print("   %.4f  x %+.4f  %s" % t)
