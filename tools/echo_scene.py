#!/usr/bin/env python3
"""The headline graph with ONE echo on its master bus (tools/echo_scene.py [voices] [delay_s] [feedback]): what a feedback loop costs.
1024 voices -> ConvolverNode (shared 65,536-tap stereo IR) -> bus -> destination ; bus -> DelayNode -> destination ; DelayNode -> Gain -> DelayNode"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphaudio_amd import AudioBufferSourceNode, ConvolverNode, DelayNode, GainNode, OfflineAudioContext, PlayableAudioBuffer
from tests import _graphs as G
SR = 48000
voices = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
delay = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
fbg = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
frames = 10 * SR // 128 * 128
for echo in (False, True):
    ctx = OfflineAudioContext(SR)
    for kv in os.environ.get("GA_OPTS", "").split(","):
        if "=" in kv: ctx.SetOption(kv.split("=")[0], float(kv.split("=")[1]))
    ir = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, 65536) for c in range(2)], SR)
    bus = GainNode(ctx); bus.Gain.Value = 0.5
    for v in range(voices):
        s = AudioBufferSourceNode(ctx); s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v % 64, frames), SR); s.Loop = True
        cv = ConvolverNode(ctx); cv.Buffer = ir
        s.Connect(cv).Connect(bus); s.Start()
    bus.Connect(ctx.Destination)
    if echo:
        d = DelayNode(ctx, 1.0); d.DelayTime.Value = delay
        fb = GainNode(ctx); fb.Gain.Value = fbg
        bus.Connect(d); d.Connect(fb).Connect(d); d.Connect(ctx.Destination)
    out = np.zeros((2, frames), np.float32)
    for rep in range(3):
        t0 = time.time(); ctx.Render(out, frames); dt = time.time() - t0
        st = ctx.GetStats()
        print(f"echo={echo} rep {rep}: {dt*1e3:8.2f} ms per 10 s -> {frames/dt/1e6:8.2f} M frames/s  chunks {st['chunks']} launches {st['kernel_launches']} ref rows {st['ref_order_rows']}", flush=True)
    ctx.Dispose()
