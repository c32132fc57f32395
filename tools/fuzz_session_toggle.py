"""tools/fuzz_session_toggle.py SEED [coarse]: one random edit session as tests/test_gpu_fuzz.py runs it, then again with one
planner shortcut switched off at a time -- which of them (if any) a deviation from the oracle belongs to."""
import sys, numpy as np
sys.path.insert(0, ".")
import os
from graphaudio_amd import OfflineAudioContext, _capi
if os.environ.get("GA_TOOL_LIBRARY"):   # a tools/build_variant.sh build with experiment switches (e.g. GA_BQ_NOPIPE=1: no pipelined cascade kernel)
    _capi.use_library(os.environ["GA_TOOL_LIBRARY"])
from tests import _graphs as G
import tests._fuzz as F
from tests._oracle import OracleContext
seed = int(sys.argv[1])
coarse = int(sys.argv[2]) if len(sys.argv) > 2 else 1
o = OracleContext(48000)
ref, rl = F.run_random_session(o, seed)
base = {"max_chunk_blocks": 11, "coarse_min_blocks": 1 if coarse else 1 << 30}
for name, extra in (("as the test runs it", {}), ("twin_channels 0", {"twin_channels": 0}), ("sim_replay 0", {"sim_replay": 0}),
                    ("gain_fold / pass 0", {"gain_fold": 0, "gain_pass_through": 0}), ("biquad_time_split 0", {"biquad_time_split": 0}),
                    ("resample_fast 0", {"resample_fast": 0}), ("cycle_delay_split 0", {"cycle_delay_split": 0}),
                    ("every sensitive sink counts", {"conv_ref_min_deviation": 0.0}), ("EVERY convolver in reference order", {"conv_reference_order": 2}),
                    ("no convolver in reference order", {"conv_reference_order": 0}), ("one block per chunk", {"max_chunk_blocks": 1})):
    h = OfflineAudioContext(48000)
    for k, v in {**base, **extra}.items():
        h.SetOption(k, v)
    got, gl = F.run_random_session(h, seed)
    st = h.GetStats()
    d = np.abs(ref - got).max(axis=0)
    bad = np.nonzero(d > 1e-4 * max(1.0, G.rms(ref)))[0]
    print(f"{name:36s} err {G.rms(ref - got):.3e} scale {G.rms(ref):.3f} first bad", (int(bad[0]), int(bad[0]) // 128, len(bad)) if len(bad) else None,
          "log same", rl == gl, "twin", st["twin_rows"], "ref-order rows", st["ref_order_rows"], flush=True)
print("pieces", F.last_pieces)
