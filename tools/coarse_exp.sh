#!/bin/bash
# timing experiments on the formulation D kernels (GA_COARSE_EXP bit flags; results are wrong by construction, only stage times count)
# forward: 1 no stores, 2 no combine, 4 no FFT, 8 no input loads ; multiply-accumulate (x16): 1 no MAC, 2 no X loads, 4 no Y stores
for e in "$@"; do
  echo "== GA_COARSE_EXP=$e"
  GA_COARSE_EXP=$e python bench.py --no-cpu-baseline --steps 5 --warmup 2 2>gpurun_out/exp_err.log | python tools/bench_stages.py
  grep coarse_mac gpurun_out/exp_err.log | sort | uniq -c | head -3
done
