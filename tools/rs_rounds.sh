#!/bin/bash
# resample_fast_kernel: rounds per workgroup (GA_RS_ROUNDS; VARIANT_KERNELS=1 tools/build_variant.sh kexp), config 4 at 4096 voices
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export GA_TOOL_LIBRARY=tools/variants/kexp.so
for r in "$@"; do
  export GA_RS_ROUNDS=$r
  rm -rf /tmp/prof_rs
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_rs -o st -- python3 tools/run_configs.py 4 10 4096 > /tmp/prof_rs.log 2>&1 || { tail -3 /tmp/prof_rs.log; exit 1; }
  echo "rounds $r: $(grep resample_fast $(find /tmp/prof_rs -name '*kernel_stats.csv' | head -1) | cut -d, -f2-4,7)"
done
