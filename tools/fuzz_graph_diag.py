"""Which option a deviating random GRAPH (tests/test_gpu_fuzz.py::test_random_graph_matches_oracle) depends on:
tools/fuzz_graph_diag.py SEED..."""
import sys, numpy as np
sys.path.insert(0, ".")
from graphaudio_amd import OfflineAudioContext
from tests import _graphs as G
from tests._fuzz import build_random_graph
from tests._oracle import OracleContext
frames = 128 * 36
for seed in [int(x) for x in sys.argv[1:]]:
    o = OracleContext(48000)
    ch = build_random_graph(o, seed, frames)
    ref = np.zeros((ch, frames), np.float32)
    o.Render(ref, frames)
    for name, opts in (("default (D forced)", {"coarse_min_blocks": 1}), ("no D", {"coarse_min_blocks": 1 << 30}),
                       ("no D, fft64", {"coarse_min_blocks": 1 << 30, "fft64": 1}),
                       ("D, no pass-through", {"coarse_min_blocks": 1, "gain_pass_through": 0}),
                       ("no D, no gain folding", {"coarse_min_blocks": 1 << 30, "gain_fold": 0}),
                       ("no D, one chunk", {"coarse_min_blocks": 1 << 30, "max_chunk_blocks": 4096}),
                       ("D, no biquad split", {"coarse_min_blocks": 1, "biquad_time_split": 0}),
                       ("D, no premix / ext hist / private tails", {"coarse_min_blocks": 1, "coarse_premix": 0, "coarse_ext_history": 0, "coarse_tail_private": 0}),
                       ("one chunk", {"coarse_min_blocks": 1, "max_chunk_blocks": 4096})):
        h = OfflineAudioContext(48000)
        h.SetOption("max_chunk_blocks", 11)
        for k, v in opts.items():
            h.SetOption(k, v)
        build_random_graph(h, seed, frames)
        got = np.zeros_like(ref)
        pos = 0
        rng = np.random.default_rng(1000 + seed)
        while pos < frames:
            n = int(min(frames - pos, rng.integers(1, 128 * 9)))
            h.Render(got, n, pos)
            pos += n
        err, sc = G.rms(ref - got), G.rms(ref)
        d = np.abs(ref - got).max(axis=0)
        bad = np.nonzero(d > 20 * err)[0]
        print(f"{seed} {name:44s} err {err:.3e} scale {sc:.4f} rel {err / sc:.2e}  max diff {d.max():.2e} at frame {int(d.argmax())}")
        h.Dispose()
    # one voice at a time (the minimiser hook of tests/_fuzz.py)
    import graphaudio_amd as ga
    for keep in [set()] + [{v} for v in range(10)]:
        outs = []
        for mk in (OracleContext, OfflineAudioContext):
            c = mk(48000)
            edges = []
            orig = ga.AudioNode.Connect
            def rec(self, target, *a, _o=orig, _e=edges, **k):
                _e.append(f"{type(self).__name__}{getattr(self, '_id', '')}->{type(target).__name__}{getattr(target, '_id', '')}")
                return _o(self, target, *a, **k)
            ga.AudioNode.Connect = rec
            try:
                build_random_graph(c, seed, frames, keep=keep)
            finally:
                ga.AudioNode.Connect = orig
            out = np.zeros((ch, frames), np.float32)
            c.Render(out, frames)
            outs.append(out)
        err = G.rms(outs[0] - outs[1])
        print(f"   keep {sorted(keep)}: err {err:.3e} scale {G.rms(outs[0]):.4f}")
    print("   edges:", " ".join(edges))
