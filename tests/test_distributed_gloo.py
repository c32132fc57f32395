"""N>1 path on CPU: two gloo ranks run the host side of the sharded render -- ga_shard_range from the product library (pure host
code, loads without a GPU), the communicator-id exchange of graphaudio_amd.distributed -- render their voice shard with the
oracle standing in for the device, and sum the destination buses with a gloo reduce standing in for the RCCL reduce that
ga_render_reduce issues on the GPU box (tests/test_gpu_device_out.py covers that call on the device)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, os.environ["GA_ROOT"])
from graphaudio_amd.distributed import shard_range, exchange_comm_id
from graphaudio_amd import AudioBufferSourceNode, ConvolverNode, PlayableAudioBuffer
from tests import _graphs as G
from tests._oracle import OracleContext
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
V, frames, taps = 6, 128 * 12, 900
b, e = shard_range(V, world, rank)
class FakeRank0:                      # the id exchange only needs CommUniqueId() on rank 0
    def CommUniqueId(self): return bytes(range(128))
assert exchange_comm_id(FakeRank0(), rank, world) == bytes(range(128))
ctx = OracleContext(48000)
irbuf = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, taps) for c in range(2)], 48000)
for v in range(b, e):
    s = AudioBufferSourceNode(ctx); s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, frames + 256), 48000)
    cv = ConvolverNode(ctx); cv.Buffer = irbuf
    s.Connect(cv).Connect(ctx.Destination); s.Start()
out = np.zeros((2, frames), np.float32)
ctx.Render(out, frames)
bus = torch.from_numpy(out)
dist.reduce(bus, dst=0, op=dist.ReduceOp.SUM)
if rank == 0:
    np.save(os.environ["GA_OUT"], bus.numpy())
dist.destroy_process_group()
'''


def test_two_rank_shard_and_bus_sum(tmp_path):
    outp = str(tmp_path / "bus.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, GA_ROOT=ROOT, GA_OUT=outp, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", str(script)]
    subprocess.check_call(cmd, env=env, cwd=ROOT, timeout=300)
    got = np.load(outp)
    # single-process reference: all voices in one context
    sys.path.insert(0, ROOT)
    from tests import _graphs as G
    from tests._oracle import OracleContext
    ctx = OracleContext(48000)
    ch = G.config3_convolver(ctx, voices=6, taps=900, frames=128 * 12)
    ref = G.render(ctx, ch, 128 * 12)
    # the cross-rank sum order differs from the sequential connection order: float32 rounding only
    assert np.abs(got - ref).max() <= 4e-7 * np.abs(ref).max() + 1e-9


def test_shard_range_partitions():
    from graphaudio_amd.distributed import shard_range
    for total in (1, 7, 1024, 4097):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in spans]
            assert max(sizes) - min(sizes) <= 1
