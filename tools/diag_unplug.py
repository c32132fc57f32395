import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from graphaudio_amd import *
from tests import _graphs as G
from tests._oracle import OracleContext
SR=48000
def run(ctx, kind, taps, chunk, nv):
    if isinstance(ctx, OfflineAudioContext):
        ctx.SetOption("max_chunk_blocks", chunk)
    ctx.Destination.SetChannelCount(2)
    convs = []
    for v in range(nv):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(70 + v, 128 * 40), SR)
        if kind == "gain":
            c = GainNode(ctx); c.Gain.Value = 0.5
        else:
            c = ConvolverNode(ctx)
            c.Buffer = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(ch, taps, seed0=50 + 10 * v) for ch in range(2)], SR)
        s.Connect(c); c.Connect(ctx.Destination); s.Start(); convs.append(c)
    out = np.zeros((2, 128 * 40), np.float32)
    pos = [0]
    def piece(nblk):
        ctx.Render(out, 128 * nblk, pos[0]); pos[0] += 128 * nblk
    piece(5); convs[0].Disconnect(); piece(3); convs[0].Connect(ctx.Destination); piece(32)
    return out
for kind, taps, chunk, nv in (("gain", 0, 4096, 1), ("gain", 0, 4096, 2), ("conv", 128*9, 4096, 1), ("conv", 128*9, 4096, 2), ("conv", 128*9, 2, 2), ("conv", 128*70, 4096, 2), ("conv", 128*70, 2, 2)):
    ro = run(OracleContext(SR), kind, taps, chunk, nv); go = run(OfflineAudioContext(SR), kind, taps, chunk, nv)
    e = [float(np.sqrt(np.mean((ro[:, b*128:(b+1)*128]-go[:, b*128:(b+1)*128])**2))) for b in range(40)]
    print(kind, taps, chunk, nv, "first bad block:", next((b for b, x in enumerate(e) if x > 1e-6), None), [f"{x:.0e}" for x in e][:14], flush=True)
