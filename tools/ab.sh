#!/bin/bash
# A/B of bench.py argument sets on ONE box, interleaved (boxes differ by several per cent): tools/ab.sh ROUNDS "argsA" "argsB" ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
rounds=$1; shift
for r in $(seq $rounds); do
  for a in "$@"; do
    python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-variants --no-check $a 2>/dev/null | python -c '
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=r["stages"]
print("%-40s ms/step %.4f device %.4f | "%(sys.argv[1][:40],r["ms_per_step"],r["device_ms_per_step"])+" ".join("%s %.4f"%(k.replace("coarse_",""),v["ms_per_step"]) for k,v in s.items()))' "$a"
  done
done
