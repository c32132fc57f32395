#!/bin/bash
# A/B of one bench.py variant (private_ir, config5_1gpu, per_voice_spectra) over argument sets, interleaved on ONE box:
#   tools/ab_variant.sh private_ir 3 "" "--library tools/variants/pb4.so"
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
v=$1; rounds=$2; shift; shift
for r in $(seq $rounds); do
  for a in "$@"; do
    python bench.py --only-variant $v --no-check --variant-steps 8 $a 2>/dev/null | python -c '
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=r.get("variants",{}).get(sys.argv[2],r); s=r["stages"]
print("%-44s ms/step %.4f device %.4f | "%(sys.argv[1][:44],r["ms_per_step"],r["device_ms_per_step"])+" ".join("%s %.4f"%(k.replace("coarse_",""),x["ms_per_step"]) for k,x in s.items()))' "$a" $v
  done
done
