#!/bin/bash
# config 4: what the time of biquad_pipe_kernel<5> is made of (VARIANT_KERNELS=1 tools/build_variant.sh kexp / kexp1 -DGA_BQ_EXP=1
# (no recurrence) / kexp2 -DGA_BQ_EXP=2 (no global loads / stores)); kernel durations from rocprofv3 --kernel-trace --stats.
#   tools/bq_waves.sh "<library> <voices> <GA_BQ_JPW or 0>" ...
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for spec in "$@"; do
  set -- $spec
  echo "== library $1 voices $2 GA_BQ_JPW=$3"
  if [ "$1" = "product" ]; then unset GA_TOOL_LIBRARY; else export GA_TOOL_LIBRARY=tools/variants/$1.so; fi
  if [ "$3" != "0" ]; then export GA_BQ_JPW=$3; else unset GA_BQ_JPW; fi
  rm -rf /tmp/prof_bq
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bq -o st -- python3 tools/run_configs.py 4 10 $2 > /tmp/prof_bq.log 2>&1 || { tail -5 /tmp/prof_bq.log; exit 1; }
  grep -E "render piece 3" /tmp/prof_bq.log
  grep -E "biquad_pipe|resample_fast" $(find /tmp/prof_bq -name "*kernel_stats.csv" | head -1) | cut -c1-160
done
