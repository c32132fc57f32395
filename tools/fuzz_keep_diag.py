"""tools/fuzz_keep_diag.py SEED VOICE [LIBRARY]: one voice of a random graph (minimiser hook), oracle vs device, per block"""
import sys, numpy as np
sys.path.insert(0, ".")
if len(sys.argv) > 3:
    from graphaudio_amd import _capi
    _capi.use_library(sys.argv[3])
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
from tests._fuzz import build_random_graph
from tests._oracle import OracleContext
seed, voice = int(sys.argv[1]), int(sys.argv[2])
frames = 128 * 36
outs = []
for mk in (OracleContext, OfflineAudioContext):
    c = mk(48000)
    if mk is OfflineAudioContext:
        import os
        for kv in os.environ.get("GA_OPTS", "").split(","):
            if "=" in kv:
                c.SetOption(kv.split("=")[0], float(kv.split("=")[1]))
    ch = build_random_graph(c, seed, frames, keep={voice})
    out = np.zeros((ch, frames), np.float32)
    c.Render(out, frames)
    outs.append(out)
d = np.abs(outs[0] - outs[1])
print("err %.3e scale %.3f" % (G.rms(outs[0] - outs[1]), G.rms(outs[0])))
for b in range(36):
    m = d[:, b * 128:(b + 1) * 128].max()
    if m > 1e-6:
        print("  block", b, "max diff %.3e" % m, "ref max %.3f" % np.abs(outs[0][:, b * 128:(b + 1) * 128]).max(), "got max %.3f" % np.abs(outs[1][:, b * 128:(b + 1) * 128]).max())
bad = np.nonzero(d.max(axis=0) > 1e-6)[0]
if len(bad):
    f = int(bad[0])
    print("first differing frame", f, "= block", f // 128, "+", f % 128, " n differing", len(bad), " last", int(bad[-1]))
    for g in (f - 2, f - 1, f, f + 1, f + 2, f + 3):
        if 0 <= g < frames:
            print("   frame", g, "ref", outs[0][:, g], "got", outs[1][:, g])
    # runs of differing frames
    runs, start, prev = [], bad[0], bad[0]
    for x in bad[1:]:
        if x != prev + 1:
            runs.append((int(start), int(prev))); start = x
        prev = x
    runs.append((int(start), int(prev)))
    print("   runs:", runs[:12])
