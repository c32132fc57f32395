"""tools/fuzz_session_ddmin.py SEED: the smallest set of voices of a random edit session that still deviates from the oracle
(greedy removal with the minimiser hook of tests/_fuzz.py), then the per-block deviation of that set."""
import sys, numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
import tests._fuzz as F
from tests._oracle import OracleContext
graph = sys.argv[1].startswith("g")   # gSEED: a graph of test_random_graph_matches_oracle instead of an edit session
seed = int(sys.argv[1][1:] if graph else sys.argv[1])


def run(ctx, keep):
    if not graph:
        return F.run_random_session(ctx, seed, keep=keep)[0]
    frames = 128 * 36
    ch = F.build_random_graph(ctx, seed, frames, keep=keep)
    out = np.zeros((ch, frames), np.float32)
    pos = 0
    rng = np.random.default_rng(1000 + seed)
    while pos < frames:
        n = int(min(frames - pos, rng.integers(1, 128 * 9)))
        ctx.Render(out, n, pos)
        pos += n
    return out


def err_of(keep):
    try:
        ref = run(OracleContext(48000), keep)
        h = OfflineAudioContext(48000)
        h.SetOption("max_chunk_blocks", 11)
        h.SetOption("coarse_min_blocks", 1)
        got = run(h, keep)
    except Exception as e:   # noqa: BLE001
        return None, None, None
    return G.rms(ref - got), ref, got


keep = set(range(16))
e0, _, _ = err_of(keep)
print("all voices:", e0)
for v in range(16):
    trial = keep - {v}
    e, _, _ = err_of(trial)
    if e is not None and e > 1e-4:
        keep = trial
print("minimal set", sorted(keep))
e, ref, got = err_of(keep)
d = np.abs(ref - got)
for b in range(ref.shape[1] // 128):
    m = d[:, b * 128:(b + 1) * 128].max()
    if m > 1e-5:
        print("   block %2d max diff %.3e per channel" % (b, m), [float("%.3g" % x) for x in d[:, b * 128:(b + 1) * 128].max(axis=1)])
print("pieces", getattr(F, "last_pieces", None))
np.save(f"gpurun_out/got_{seed}.npy", got)   # (for tools/fuzz_node_dump.py SEED KEEP BLOCK gpurun_out/got_SEED.npy)
print("keep", ",".join(str(v) for v in sorted(keep)))
