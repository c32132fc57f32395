#!/bin/bash
# legs of the multiply-accumulate kernel (GA_COARSE_EXP >> 4): 0 = all, 16 = no sweep, 32 = no loads of later terms, 48 = neither, 64 = no stores
for e in 0 32 160 128; do
  GA_COARSE_EXP=$e python bench.py --only-variant private_ir --no-check --variant-steps 6 --library tools/variants/exp.so > gpurun_out/exp_$e.json 2> gpurun_out/exp_$e.err
  python - <<P
import json
d=json.loads(open("gpurun_out/exp_$e.json").read().strip().splitlines()[-1])
v=d["variants"]["private_ir"] if "variants" in d else d
print("exp $e", "step %.3f ms" % v["ms_per_step"], {k:round(s["ms_per_step"],3) for k,s in v.get("stages",{}).items()} if isinstance(v.get("stages"),dict) else v.get("roofline",{}).get("avg_launch_ms"))
P
done
