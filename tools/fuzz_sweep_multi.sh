#!/bin/bash
# several seed ranges of tools/fuzz_sweep.py in one GPU call: tools/fuzz_sweep_multi.sh "FIRST LAST MODES..." ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
for spec in "$@"; do
  echo "=== $spec"
  python tools/fuzz_sweep.py $spec 2>&1 | grep -E "FAIL|done|seed .*00 "
done
