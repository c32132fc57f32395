#!/bin/bash
# config 4 at 4096 voices: cascades per wave of biquad_pipe_kernel<5> (GA_BQ_JPW; VARIANT_KERNELS=1 tools/build_variant.sh kexp)
cd "$GRAFT_REPO_ROOT"
for j in 4 6 8 12; do
  echo "== GA_BQ_JPW=$j"
  GA_TOOL_LIBRARY=tools/variants/kexp.so GA_BQ_JPW=$j timeout -k 10 200 python3 tools/run_configs.py 4 10 4096 2>&1 | grep -E "render piece|device_ms|rms" || exit 1
done
