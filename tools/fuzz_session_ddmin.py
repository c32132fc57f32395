"""tools/fuzz_session_ddmin.py SEED: the smallest set of voices of a random edit session that still deviates from the oracle
(greedy removal with the minimiser hook of tests/_fuzz.py), then the per-block deviation of that set."""
import sys, numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
import tests._fuzz as F
from tests._oracle import OracleContext
seed = int(sys.argv[1])


def err_of(keep):
    try:
        ref, _ = F.run_random_session(OracleContext(48000), seed, keep=keep)
        h = OfflineAudioContext(48000)
        h.SetOption("max_chunk_blocks", 11)
        h.SetOption("coarse_min_blocks", 1)
        got, _ = F.run_random_session(h, seed, keep=keep)
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
print("pieces", F.last_pieces)
np.save(f"gpurun_out/got_{seed}.npy", got)   # (for tools/fuzz_node_dump.py SEED KEEP BLOCK gpurun_out/got_SEED.npy)
print("keep", ",".join(str(v) for v in sorted(keep)))
