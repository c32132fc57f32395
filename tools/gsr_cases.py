"""Small targeted cases for the general source replay path (GPU vs oracle)."""
import sys, numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext, AudioBufferSourceNode, PlayableAudioBuffer
from tests._oracle import OracleContext
from tests import _graphs as G
rng = np.random.default_rng(3)
data = (rng.standard_normal(5089) * 0.25).astype(np.float32)
def run(mk, loop0, toggle_at, t_ev, offset, duration, rate_after=0.99):
    ctx = mk(48000); ctx.Destination.SetChannelCount(1)
    s = AudioBufferSourceNode(ctx); s.Buffer = PlayableAudioBuffer.FromMonoArray(data, 48000)
    s.Loop = loop0
    if t_ev is not None:
        s.PlaybackRate.SetValueAtTime(rate_after, t_ev)
        s.PlaybackRate.LinearRampToValueAtTime(1.0004, t_ev + 0.08)
    s.Connect(ctx.Destination); s.Start(0.0, offset, duration)
    out = np.zeros((1, 128 * 40), np.float32)
    if toggle_at:
        ctx.Render(out, toggle_at, 0); s.Loop = not s.Loop; ctx.Render(out, 128 * 40 - toggle_at, toggle_at)
    else:
        ctx.Render(out, 128 * 40)
    return out
cases = {
 "timeline no loop": (False, 0, 0.0367, 0.0, float("inf")),
 "timeline loop from start": (True, 0, 0.0367, 0.0, float("inf")),
 "timeline loop toggled on": (False, 1579, 0.0367, 0.0, float("inf")),
 "timeline loop toggled on, offset+duration": (False, 1579, 0.0367, 0.00636, 0.088),
 "timeline, offset+duration no loop": (False, 0, 0.0367, 0.00636, 0.088),
 "timeline loop from start, offset+duration": (True, 0, 0.0367, 0.00636, 0.088),
 "const rate loop toggled on": (False, 1579, None, 0.0, float("inf")),
}
for name, a in cases.items():
    ref = run(OracleContext, *a); got = run(OfflineAudioContext, *a)
    d = np.abs(ref - got).max(axis=0); bf = np.nonzero(d > 1e-6)[0]
    print(f"{name:45s} err {G.rms(ref-got):.3e} first bad {(int(bf[0])//128, int(bf[0])%128, len(bf)) if len(bf) else None}")
