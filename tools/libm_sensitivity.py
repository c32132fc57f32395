"""Is a quantised parameter behind an AUTOMATED biquad sensitive to the last bit of cosf / sinf?  (no convolver in the graph)"""
import sys; sys.path.insert(0, ".")
import numpy as np
from graphaudio_amd import *
from tests import _graphs as G
from tests._oracle import OracleContext
SR = 48000
def scene(ctx, seed):
    rng = np.random.default_rng(seed)
    frames = 128 * 60
    v = AudioBufferSourceNode(ctx); v.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(seed, frames), SR)
    d = DelayNode(ctx, 0.05); d.DelayTime.Value = float(rng.uniform(0.001, 0.01))
    v.Connect(d).Connect(ctx.Destination); v.Start()
    x = AudioBufferSourceNode(ctx); x.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(seed + 100, frames), SR); x.Start()
    bq = BiQuadFilterNode(ctx); bq.Type = FilterType(int(rng.integers(0, 8))); bq.Frequency.Value = float(rng.uniform(300, 3000))
    lfo = AudioBufferSourceNode(ctx); lfo.Buffer = PlayableAudioBuffer.FromMonoArray((rng.standard_normal(700) * 300).astype(np.float32), SR); lfo.Loop = True; lfo.Start()
    lfo.Connect(bq.Frequency)
    depth = GainNode(ctx); depth.Gain.Value = 0.004
    x.Connect(bq).Connect(depth); depth.Connect(d.DelayTime)
bad = 0
for seed in range(40):
    o = OracleContext(SR); scene(o, seed); ref = G.render(o, 2, 128 * 60)
    h = OfflineAudioContext(SR); scene(h, seed); got = G.render(h, 2, 128 * 60)
    d = np.abs(ref - got); nb = int((d.reshape(2, -1, 128).max(axis=(0, 2)) > 1e-4).sum())
    if nb: bad += 1; print("seed", seed, "blocks off", nb, "rms err %.2e" % G.rms(ref - got))
print("graphs with a block off:", bad, "of 40")
