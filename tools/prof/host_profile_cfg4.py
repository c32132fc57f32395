#!/usr/bin/env python3
"""Host-side sampling profile of the chunk engine on config 4 (4096 voices x 7 nodes): many short renders so that the control
plane dominates.  LD_PRELOAD=tools/prof/libsigprof.so GA_SIGPROF_OUT=gpurun_out/prof.txt python tools/prof/host_profile_cfg4.py"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from graphaudio_amd import OfflineAudioContext, _capi
if os.environ.get("GA_TOOL_LIBRARY"):   # a tools/build_variant.sh build (e.g. with -g for line numbers; set GA_SIGPROF_LIB to its file name)
    _capi.use_library(os.environ["GA_TOOL_LIBRARY"])
from tests import _graphs as G
voices = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
pieces = int(sys.argv[2]) if len(sys.argv) > 2 else 60
frames = 128 * (int(sys.argv[3]) if len(sys.argv) > 3 else 64)
ctx = OfflineAudioContext(48000)
ch = G.config4_eq(ctx, voices=voices, frames=frames * pieces)
out = np.zeros((ch, frames), np.float32)
ctx.Render(out, frames)
ctx.Render(out, frames)
if os.environ.get("GA_SIGPROF_DEFER"):
    ctypes.CDLL(None).sigprof_start()
t0 = time.time()
for _ in range(pieces - 1):
    ctx.Render(out, frames)
print(f"{(time.time() - t0) / (pieces - 1) * 1e3:.2f} ms per render of {frames} frames, {voices} voices", flush=True)
if os.environ.get("GA_SIGPROF_OUT"):
    ctypes.CDLL(None).sigprof_dump()
