#!/usr/bin/env python3
"""One-off wider sweeps of the GPU fuzzers (not part of the test suite): tools/fuzz_sweep.py FIRST LAST [coarse values]"""
import sys, time
sys.path.insert(0, ".")
import pytest
from tests.test_gpu_fuzz import test_random_graph_matches_oracle, test_random_edit_session_matches_oracle
first, last = int(sys.argv[1]), int(sys.argv[2])
coarse_values = [int(x) for x in sys.argv[3:]] or [1]
bad = []
t0 = time.time()
for seed in range(first, last):
    for coarse in coarse_values:
        for fn in (test_random_graph_matches_oracle, test_random_edit_session_matches_oracle):
            try:
                fn(seed, coarse)
            except pytest.skip.Exception:
                pass
            except Exception as e:   # noqa: BLE001
                bad.append((fn.__name__, seed, coarse, repr(e)[:200]))
                print("FAIL", bad[-1], flush=True)
    if seed % 50 == 0:
        print(f"seed {seed}  {time.time() - t0:.0f} s  failures {len(bad)}", flush=True)
print("done", last - first, "seeds,", len(bad), "failures", bad[:5])
