import sys, numpy as np
sys.path.insert(0, ".")
import graphaudio_amd.core as core
from tests._fuzz import build_random_graph
from tests._oracle import OracleContext
seed = int(sys.argv[1]); frames = 128 * 36
o = OracleContext(48000)
orig = o._call
def logged(fn, *a):
    def fmt(x):
        if isinstance(x, (int, float, bytes)): return x
        return type(x).__name__
    if fn not in ("param_get_value",): print(fn, [fmt(x) for x in a])
    return orig(fn, *a)
o._call = logged
orig_bid = core.PlayableAudioBuffer._native_id
def bid(self, ctx):
    r = orig_bid(self, ctx); print("   buffer", r, "ch", self.NumberOfChannels, "len", self.Length, "sr", self.SampleRate); return r
core.PlayableAudioBuffer._native_id = bid
build_random_graph(o, seed, frames)
