#!/usr/bin/env python3
"""Render the other BASELINE.json configurations at full size on the HIP path and report frames/s (not bench lines:
bench.py measures configs[2]; these are parity-test cases, timed here to find performance bugs)."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphaudio_amd import OfflineAudioContext, _capi
from tests import _graphs as G

if os.environ.get("GA_TOOL_LIBRARY"):   # a tools/build_variant.sh build (this tool only; the product reads no such variable)
    _capi.use_library(os.environ["GA_TOOL_LIBRARY"])

SR = 48000
which = sys.argv[1] if len(sys.argv) > 1 else "2"
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
frames = int(seconds * SR) // 128 * 128
ctx = OfflineAudioContext(SR)
ctx.SetOption("profile", 1)
for kv in os.environ.get("GA_OPTS", "").split(","):   # e.g. GA_OPTS=coarse_tail_private=0,async=1
    if "=" in kv:
        ctx.SetOption(kv.split("=")[0], float(kv.split("=")[1]))
t0 = time.time()
if which == "2":
    ch = G.config2_biquad(ctx, voices=256, frames=frames + 256)
elif which == "4":
    ch = G.config4_eq(ctx, voices=int(sys.argv[3]) if len(sys.argv) > 3 else 4096, frames=frames)
elif which == "3u":
    ch = G.config3_convolver(ctx, voices=int(sys.argv[3]) if len(sys.argv) > 3 else 64, taps=65536, frames=frames, shared=False)
elif which == "5":
    from graphaudio_amd import AudioBufferSourceNode, ConvolverNode, PlayableAudioBuffer
    nsrc = int(sys.argv[3]) if len(sys.argv) > 3 else 512
    taps, C = 32768, 16
    ctx.Destination.SetChannelCount(C)
    n = np.arange(taps)
    env = np.exp(-6.9 * n / taps).astype(np.float32)
    for v in range(nsrc):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), SR)
        rng = np.random.default_rng(7 + 100 * v)
        irs = [(rng.standard_normal(taps, dtype=np.float32) * env) for c in range(C)]
        cv = ConvolverNode(ctx)
        cv.Buffer = PlayableAudioBuffer.FromChannelArrays(irs, SR)
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()
    ch = C
elif which == "kit":   # buses + panners + post-mix reverb
    ch = G.kit_scene(ctx, voices=int(sys.argv[3]) if len(sys.argv) > 3 else 256, frames=frames + 256, taps=65536)
elif which == "osc":   # oscillators + panners (8(f) rank 1 nodes at scale)
    from graphaudio_amd import OscillatorNode, OscillatorType, StereoPannerNode, GainNode
    n_osc = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
    for v in range(n_osc):
        o = OscillatorNode(ctx)
        o.Type = OscillatorType(v % 4)
        o.Frequency.Value = 55.0 * 2.0 ** (v / 128.0)
        p = StereoPannerNode(ctx)
        p.Inputs[0].SetChannelCount(1)
        p.Pan.Value = -1.0 + 2.0 * v / max(n_osc - 1, 1)
        g = GainNode(ctx)
        g.Gain.Value = 1.0 / 64.0
        o.Connect(p).Connect(g).Connect(ctx.Destination)
        o.Start()
    ch = 2
else:
    raise SystemExit("unknown config")
print(f"build {time.time() - t0:.1f} s")
out = np.zeros((ch, frames), np.float32)
piece = frames // 4 // 128 * 128
for rep in range(4):   # the first pieces carry one-time costs (formulation assignment, first touch of scratch memory): quote the last
    t0 = time.time()
    ctx.Render(out, piece, rep * piece)
    dt = time.time() - t0
    print(f"render piece {rep}: {dt * 1e3:.1f} ms -> {piece / dt / 1e6:.2f} M frames/s")
st = ctx.GetStats()
print(json.dumps({k: st[k] for k in ("chunks", "segments", "kernel_launches", "device_ms_total", "mac_ms_total", "fft_ms_total", "other_ms_total", "device_bytes_in_use")}))
print("rms", G.rms(out))

if os.environ.get("GA_SIGPROF_OUT"):   # tools/prof/sigprof.c preloaded: write its samples before the process leaves through _exit
    import ctypes
    ctypes.CDLL(None).sigprof_dump()
