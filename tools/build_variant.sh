#!/bin/bash
# builds a variant of the product library with extra compiler flags into tools/variants/<name>.so (measurements only):
#   tools/build_variant.sh mac6 -DGA_MAC_WAVES=6 -DGA_MAC_TW=11 -DGA_MAC_PB2=4
set -e
name=$1; shift
cd "$(dirname "$0")/../graphaudio_amd/csrc"
mkdir -p ../../tools/variants /tmp/variant_$name
F="-DGA_EXPERIMENTS --offload-arch=gfx950 -O3 -g1 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden -Wall -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c ga_coarse.hip -o /tmp/variant_$name/ga_coarse.o &
/opt/rocm/bin/hipcc $F "$@" -x hip -c ga_chunk.cpp -o /tmp/variant_$name/ga_chunk.o &   # (the planner shares the job-size constants)
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/variants/$name.so ga_kernels.o /tmp/variant_$name/ga_coarse.o ga_engine.o /tmp/variant_$name/ga_chunk.o ga_comm.o ga_api.o -ldl
echo built tools/variants/$name.so
