#!/bin/bash
# A/B of coarse_premix_kernel variants on the GPU box (run from the repo root): builds each variant of ga_coarse.hip and times
# the headline step with it.  tools/premix_sweep.sh "name:-Dflags" ...
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
run() {
  python bench.py --steps 16 --warmup 4 --no-cpu-baseline --no-variants --no-check "$@" 2>/dev/null | python -c '
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
s=r["stages"]
print("  ms/step %.4f device %.4f | "%(r["ms_per_step"],r["device_ms_per_step"])+" ".join("%s %.4f"%(k.replace("coarse_",""),v["ms_per_step"]) for k,v in s.items()), "| premix GB/s %.0f"%(s["coarse_premix"]["gb_per_s"] if "coarse_premix" in s else 0))'
}
echo "== product"; run; run
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  tools/build_variant.sh $name $flags > /dev/null 2>gpurun_out/variant_$name.err || { echo "== $name: build failed"; continue; }
  echo "== $name ($flags)"; run --library tools/variants/$name.so; run --library tools/variants/$name.so
done
