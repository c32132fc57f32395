#!/bin/bash
# builds a variant of the product library with extra compiler flags into tools/variants/<name>.so (measurements only):
#   tools/build_variant.sh mac6 -DGA_MAC_WAVES=6 -DGA_MAC_TW=11 -DGA_MAC_PB2=4
set -e
name=$1; shift
cd "$(dirname "$0")/../graphaudio_amd/csrc"
mkdir -p ../../tools/variants /tmp/variant_$name
F="-DGA_EXPERIMENTS --offload-arch=gfx950 -O3 -g1 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c ga_coarse.hip -o /tmp/variant_$name/ga_coarse.o &
KO=ga_kernels.o
if [ -n "$VARIANT_KERNELS" ]; then   # (ga_kernels.hip too: its experiment switches, e.g. GA_BQ_JPW; minutes of compile time)
  KO=/tmp/variant_$name/ga_kernels.o
  /opt/rocm/bin/hipcc $F "$@" -c ga_kernels.hip -o $KO &
fi
for f in ga_chunk ga_sources ga_plan_nodes ga_plan_conv; do   # (the planner shares the job-size constants and the experiment switches)
  /opt/rocm/bin/hipcc $F "$@" -x hip -c $f.cpp -o /tmp/variant_$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/variants/$name.so $KO /tmp/variant_$name/ga_coarse.o ga_engine.o /tmp/variant_$name/ga_chunk.o \
  /tmp/variant_$name/ga_sources.o /tmp/variant_$name/ga_plan_nodes.o /tmp/variant_$name/ga_plan_conv.o ga_comm.o ga_api.o -ldl
echo built tools/variants/$name.so
