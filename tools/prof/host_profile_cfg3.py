#!/usr/bin/env python3
"""Host-side sampling profile of the chunk engine on the headline graph (1024 voices -> ConvolverNode -> destination, 10 s steps,
pipelined): with the time-domain pre-mix the device needs ~0.5 ms per step, so the host's simulation + planning is what bounds
the step.  LD_PRELOAD=tools/prof/libsigprof.so GA_SIGPROF_OUT=gpurun_out/prof3.txt python tools/prof/host_profile_cfg3.py [steps]"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
voices = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
frames = 480000
ctx = OfflineAudioContext(48000)
ctx.SetOption("async", 1)
bench.build_graph(ctx, voices, 0, 65536, frames, G)
host = torch.zeros((2, frames), dtype=torch.float32).pin_memory()
out = host.numpy()
for _ in range(3):
    ctx.Render(out, frames)
ctx.Synchronize()
t0 = time.time()
for _ in range(steps):
    ctx.Render(out, frames)
t1 = time.time()
ctx.Synchronize()
t2 = time.time()
print(f"{(t1 - t0) / steps * 1e3:.3f} ms host issue per step, {(t2 - t0) / steps * 1e3:.3f} ms per step, {voices} voices", flush=True)
if os.environ.get("GA_SIGPROF_OUT"):
    ctypes.CDLL(None).sigprof_dump()
