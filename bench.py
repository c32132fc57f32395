#!/usr/bin/env python3
"""bench.py -- rendered frames/sec @48 kHz on the BASELINE.json headline workload.

Workload (config.workload, BASELINE.json configs[2], SURVEY.md section 8d "Config 3"):
    1024 mono voices -> per-voice ConvolverNode sharing one 2-channel 65,536-tap IR -> destination (2 ch), 48 kHz,
    synthetic noise voices + synthetic room-like IR.
A "step" renders `--seconds` (default 10 s = 480,000 frames = 3,750 blocks) of that graph through the C ABI; steps continue
the same render (DSP state persists), voices loop over a 10 s buffer so inputs stay resident in HBM for any number of steps.

Multi-GPU (--gpus N): one rank per GPU.  Launched by `torch.distributed.run --nproc-per-node N` the ranks come from the
environment; launched plainly with --gpus N > 1 this script starts the N rank processes itself (before anything touches a
GPU) and relays rank 0's line.  The voices are sharded V/N per rank (strong scaling: the job is the same 1024-voice mix);
every rank calls ga_render_reduce: render its share, ONE RCCL sum of the destination bus per step inside the product
library (include/graphaudio_hip.h "sharded render"), the result lands in rank 0's page-locked host buffer.  The process
group (gloo) only carries the communicator id, the barrier and the max-over-ranks time.

One JSON line on stdout (rank 0).  Besides the contract keys:
  roofline       -- the dominant kernel of the EXECUTED formulation (default: formulation D, coarse partitions; its forward
                    transform coarse_fwd_kernel).  `achieved` = the HBM bytes the launch HAS to move in that formulation
                    (inputs read once + outputs written once, computed by the planner, ga_stats.stage_bytes) / its average
                    launch duration measured live with HIP events on the context's stream; `peak` = 8 TB/s; `traffic` = null
                    here (the PMC measurement of the same command, with the guide's gfx950 FETCH_SIZE correction, is in
                    profiles/: it needs its own rocprofv3 passes).
  stages         -- the same three numbers for every stage of the step, and `whole_step` for their sum.
  streaming_formulation -- SURVEY.md 8(d)'s per-block STREAMING bytes of the reference's algorithm (1.086 GB/block here) over the
                    step time: how much of the reference's traffic the formulation removes (not a roofline fraction).
  cpu_baseline   -- the CPU oracle (C++ restatement of the reference's single-threaded render; .NET cannot run here) timed on
                    this host: 1 thread on the full 1024-voice graph (the reference renders on one thread), and all cores
                    with the voices partitioned across processes.
  parity         -- RMS error of this GPU path against the oracle on the 1 s short form of the SAME 1024-voice graph.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 48000
PEAK_F32_TFLOPS = 157.3        # MI355X_MICROARCH.md: peak FP32 (vector = matrix)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak (spec); ~6.3 TB/s is what a streaming copy reaches
STAGES = ("other", "mix", "rfft_fwd", "mac", "rfft_inv", "coarse_fwd", "coarse_mac", "coarse_inv", "coarse_hist", "coarse_section")
STAGE_KERNELS = {
    "mix": "mix_kernel (AudioNodeInput.MixBuffer)",
    "rfft_fwd": "hist_copy_b_kernel + rfft_fwd_b_kernel (256-point forward transforms)",
    "mac": "tconv16_kernel (partition sum as an FFT convolution along the block axis) / spectral_mac_* (matrix cores)",
    "rfft_inv": "irfft_ola_b_kernel (256-point inverse transforms + overlap-add)",
    "coarse_fwd": "coarse_fwd_kernel (16,384-point real transforms of the input windows: two 4096-point complex radix-16 "
                  "transforms per window in LDS + combine pass)",
    "coarse_mac": "coarse_mac_kernel (sliding partition sum over LDS-staged spectra, accumulators of 32 voices in registers)",
    "coarse_inv": "coarse_inv_kernel (frequency-domain mix + inverse transforms)",
    "coarse_hist": "coarse_hist_kernel (input history of the next chunk)",
}


def build_graph(ctx, voices, v0, taps, loop_frames, G, loop=True, private_ir=False):
    from graphaudio_amd import AudioBufferSourceNode, ConvolverNode, PlayableAudioBuffer
    irbuf = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, taps) for c in range(2)], SR)
    ctx.Destination.SetChannelCount(2)
    for v in range(v0, v0 + voices):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, loop_frames), SR)
        s.Loop = loop
        cv = ConvolverNode(ctx)
        if private_ir:   # --private-ir: every voice its own impulse response (the general multiply-accumulate path; not the headline)
            irbuf = PlayableAudioBuffer.FromChannelArrays([np.roll(G.synth_ir(c, taps), 37 * v) for c in range(2)], SR)
        cv.Buffer = irbuf
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()


def _cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_worker(args):
    """one process of the all-cores baseline: voices [v0, v0 + n) of the graph, `blocks` blocks; returns seconds"""
    v0, n, taps, blocks = args
    from tests import _graphs as G
    from tests._oracle import OracleContext
    ctx = OracleContext(SR)
    frames = blocks * 128
    build_graph(ctx, n, v0, taps, frames + 256, G, loop=False)
    out = np.zeros((2, frames), np.float32)
    ctx.Render(out, 128)
    t0 = time.perf_counter()
    ctx.Render(out, frames - 128, 128)
    dt = time.perf_counter() - t0
    ctx.Dispose()
    return dt


def cpu_all_cores(voices, taps, args):
    """All cores: the voices partitioned across worker processes (the oracle is single-threaded like the reference).  Runs BEFORE
    this process initialises the GPU: worker processes are started with fork + exec, which a GPU-initialised process must not do."""
    blocks = args.baseline_blocks
    frames = blocks * 128
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, args.baseline_cores or 16, voices))   # (a 1-GPU box gives this job a 16-core share)
    per = [(voices * i // cores, voices * (i + 1) // cores - voices * i // cores, taps, blocks) for i in range(cores)]
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(cores) as pool:
        pool.map(_oracle_worker, [(0, 1, 1024, 2)] * cores)   # start the workers (imports, oracle build check) outside the timed region
        t0 = time.perf_counter()
        pool.map(_oracle_worker, per, chunksize=1)
        dtn = time.perf_counter() - t0
    return {"value": (frames - 128) / dtn, "unit": "frames/s", "cores": cores,
            "sample": f"all {voices} voices x {blocks - 1} blocks of the bench graph, voices partitioned over {cores} processes, {dtn:.1f} s wall "
                      f"incl. graph construction (an upper bound: the reference renders on one thread)"}


def cpu_baseline_and_parity(voices, taps, G, args, all_cores):
    """(i) the oracle, ONE thread, on the full graph for the 1 s short form (375 blocks): the faithful CPU number, and the
    reference output the GPU render of the same graph is compared with; (ii) all cores, voices partitioned across processes
    (an upper bound the single-threaded reference cannot reach without modification)."""
    from graphaudio_amd import OfflineAudioContext
    from tests._oracle import OracleContext
    blocks = args.baseline_blocks
    frames = blocks * 128
    ctx = OracleContext(SR)
    build_graph(ctx, voices, 0, taps, frames + 256, G, loop=False)
    ref = np.zeros((2, frames), np.float32)
    ctx.Render(ref, 128)   # first block outside the timed region (queued commands, lazy allocations)
    t0 = time.perf_counter()
    ctx.Render(ref, frames - 128, 128)
    dt1 = time.perf_counter() - t0
    ctx.Dispose()
    fps1 = (frames - 128) / dt1
    # the GPU path on the same graph (default options): as one render call, and as two (the second one renders from the tails
    # the first one left -- the steady state of the timed steps)
    sig = G.rms(ref)
    errs, carried = [], 0
    for pieces in ([frames], [min(frames, 256 * 128), frames - min(frames, 256 * 128)]):
        h = OfflineAudioContext(SR)
        build_graph(h, voices, 0, taps, frames + 256, G, loop=False)
        got = np.zeros((2, frames), np.float32)
        pos = 0
        for n in pieces:
            if n > 0:
                h.Render(got, n, pos)
                pos += n
        carried = h.GetStats()["coarse_carried_outputs"]
        h.Dispose()
        errs.append(G.rms(ref - got))
    err = max(errs)
    parity = {"rms_abs": err, "rms_relative_to_bus": err / sig, "bus_rms": sig, "tolerance_rms_abs": 1e-5,
              "rms_abs_one_call": errs[0], "rms_abs_two_calls": errs[1], "outputs_from_carried_tails_in_the_second_call": carried,
              "sample": f"{voices} voices x {blocks} blocks (1 s short form) of the bench graph, all voices, vs the CPU oracle; "
                        "rendered as one call and as 256 + the remaining blocks"}
    base = {
        "value": fps1, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"all {voices} voices x {blocks - 1} blocks of the bench graph ({taps}-tap stereo IR, 1 s short form), "
                  f"{dt1:.1f} s on one thread, unscaled",
        "all_cores": all_cores,
        "host_cpu": _cpu_name(), "host_cores_available": os.cpu_count(),
    }
    return base, parity


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (this process never touches a GPU)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GA_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out = procs[0].communicate()[0].decode()
    rc = 0
    for p in procs:
        rc = rc or p.wait()
    sys.stdout.write(out)
    sys.exit(rc)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=10.0, help="audio seconds rendered per step")
    ap.add_argument("--voices", type=int, default=1024)
    ap.add_argument("--taps", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--baseline-blocks", type=int, default=375)
    ap.add_argument("--baseline-cores", type=int, default=0)
    ap.add_argument("--direct", action="store_true", help="direct (matrix-core) partition sum, formulation A")
    ap.add_argument("--no-coarse", action="store_true", help="formulation C (block-axis FFT) instead of D (coarse partitions)")
    ap.add_argument("--overlap", action="store_true", help="formulation D: forward and multiply-accumulate stages concurrently on two streams (measured slower)")
    ap.add_argument("--profile-every", type=int, default=4, help="record the per-stage HIP events on every k-th chunk of the timed region")
    ap.add_argument("--no-profile", action="store_true", help="no per-stage HIP events (measurement of their cost; the roofline object is then empty)")
    ap.add_argument("--private-ir", action="store_true", help="every voice convolves with its own impulse response (general path; measurement, not the headline config)")
    ap.add_argument("--no-host-direct", action="store_true", help="copy the bus to the host with copy kernels instead of writing it from the last kernel (measurement)")
    ap.add_argument("--no-tail", action="store_true", help="formulation D: no carried output tails, every chunk re-transforms the input history (measurement)")
    ap.add_argument("--no-carry", action="store_true", help="formulation D: copy the input history with its own kernel instead of from the forward transforms (measurement)")
    ap.add_argument("--copy-stream", action="store_true", help="hand the bus to the host on a copy stream of its own (measurement)")
    ap.add_argument("--sync-steps", action="store_true", help="one blocking render per step (no host/device pipelining)")
    ap.add_argument("--force-dist", action="store_true", help="exercise the sharded-render path (ga_render_reduce) even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        spawn_ranks(args.gpus)   # does not return
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")

    all_cores = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        all_cores = cpu_all_cores(args.voices, args.taps, args)   # before anything initialises the GPU in this process

    import torch   # (first: its bundled HIP runtime has to be the one in the process, tests/conftest.py)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only; the bus sum is RCCL inside the library

    from graphaudio_amd import OfflineAudioContext
    from graphaudio_amd.distributed import init_sharded, shard_range
    from tests import _graphs as G

    frames = int(round(args.seconds * SR)) // 128 * 128
    voices_total = args.voices
    v0, v1 = shard_range(voices_total, world, rank)
    use_reduce = world > 1 or args.force_dist

    ctx = OfflineAudioContext(SR, device=local_rank)
    ctx.SetOption("profile", 0 if args.no_profile else 1)
    # the events cost device time: only every k-th chunk records them (every chunk when the run is too short to sample)
    ctx.SetOption("profile_every", args.profile_every if args.steps >= 2 * args.profile_every else 1)
    ctx.SetOption("max_chunk_blocks", 4096)
    if args.direct:
        ctx.SetOption("time_fft", 0)
    if args.no_coarse:
        ctx.SetOption("coarse", 0)
    if args.overlap:
        ctx.SetOption("coarse_overlap", 1)
    if args.no_carry:
        ctx.SetOption("coarse_carry", 0)
    if args.no_tail:
        ctx.SetOption("coarse_tail", 0)
    if args.no_host_direct:
        ctx.SetOption("host_direct", 0)
    if args.copy_stream:
        ctx.SetOption("host_copy_stream", 1)
    build_graph(ctx, v1 - v0, v0, args.taps, frames, G, private_ir=args.private_ir)
    if use_reduce:
        init_sharded(ctx, rank, world)

    # the caller's output buffer is page-locked host memory (the D2H copy of the 3.8 MB bus is inside the timed region)
    host_pin = torch.zeros((2, frames), dtype=torch.float32).pin_memory()
    host_out = host_pin.numpy()
    pipelined = not args.sync_steps
    if pipelined:
        # ga_synchronize (include/graphaudio_hip.h): a render call returns once its work is enqueued, so the host-side
        # simulation + planning of step k + 1 overlaps the device execution of step k; every step still renders its 10 s, sums
        # the buses and lands in the page-locked host buffer -- the timed region ends with a full synchronisation
        ctx.SetOption("async", 1)

    def step():
        if use_reduce:
            ctx.RenderReduce(host_out, frames)
        else:
            ctx.Render(host_out, frames)

    def sync():
        ctx.Synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    st0 = ctx.GetStats()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0   # host time to issue all steps (== dt when every step blocks)
    sync()
    dt = time.perf_counter() - t0
    st1 = ctx.GetStats()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        value = frames * args.steps / dt
        blocks = frames // 128
        stages = {}
        tot_ms = tot_b = 0.0
        # stage times come from the chunks that recorded events (every --profile-every-th), scaled to a step
        nprof = max(st1["profiled_chunks"] - st0["profiled_chunks"], 1)
        per_step = (st1["chunks"] - st0["chunks"]) / args.steps / nprof
        for i, name in enumerate(STAGES):
            ms = (st1["stage_ms"][i] - st0["stage_ms"][i]) * per_step
            nl = (st1["stage_launches"][i] - st0["stage_launches"][i]) / args.steps
            by = (st1["stage_bytes"][i] - st0["stage_bytes"][i]) / args.steps
            if nl <= 0:
                continue
            gbs = by / (ms * 1e-3) / 1e9 if ms > 0 and by > 0 else None
            stages[name] = {"ms_per_step": ms, "launches_per_step": nl, "necessary_gb_per_step": by / 1e9 if by > 0 else None,
                            "gb_per_s": gbs, "frac_of_hbm_peak": gbs / PEAK_HBM_GBS if gbs else None,
                            "kernel": STAGE_KERNELS.get(name, name)}
            if by > 0:
                tot_ms += ms
                tot_b += by
        if args.no_profile or not stages:   # --no-profile (measurement of the events' cost): nothing to price
            print(json.dumps({"ms_per_step": dt / args.steps * 1e3, "value": value, "host_issue_ms_per_step": t_enq / args.steps * 1e3,
                              "device_ms_per_step": 0.0, "stages": {}}))
            return
        dom = max((n for n in stages if stages[n]["necessary_gb_per_step"]), key=lambda n: stages[n]["ms_per_step"])
        d = stages[dom]
        per_launch_ms = d["ms_per_step"] / d["launches_per_step"]
        per_launch_bytes = d["necessary_gb_per_step"] * 1e9 / d["launches_per_step"]
        form = ("formulation D (DESIGN.md): overlap-save with coarse partitions of 8192 samples (8 per 65,536-tap IR instead of 512), "
                "16,384-point real transforms in LDS per voice, the destination sum fused in the frequency domain"
                + (" (voices that share the impulse response: their spectra are summed before the spectral multiply)" if not args.private_ir else
                   " (a private impulse response per voice: every product evaluated)")
                + ", output tails carried from step to step instead of re-transforming the input history" * (not args.no_tail)
                if "coarse_fwd" in stages else
                "formulation C: partition sum as an FFT convolution along the block axis" if not args.direct else
                "formulation A: direct partition sum on the f32 matrix cores")
        dev_ms = (st1["device_ms_total"] - st0["device_ms_total"]) * per_step
        stream_bytes = (st1["mac_bytes_total"] - st0["mac_bytes_total"]) / args.steps
        rec = {
            "metric": "rendered frames/sec @48kHz, 1024-voice convolver graph",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{voices_total} voices -> PartitionedConvolver, {args.taps}-tap stereo IR {'of its own per voice' if args.private_ir else 'shared by all voices'} "
                                   f"(P={(args.taps + 127) // 128}), 128-sample blocks, 48 kHz, {blocks} blocks per step",
                       "voices": voices_total, "taps": args.taps, "frames_per_step": frames,
                       "parallelism": f"voice-shard x{world}, one RCCL reduce of the bus per step (ga_render_reduce)" if world > 1 else "single GPU",
                       "steps_pipelined": pipelined, "formulation": form},
            "realtime_factor": value / SR,
            "host_issue_ms_per_step": t_enq / args.steps * 1e3,
            "device_ms_per_step": dev_ms,
            "profiled_chunks": nprof,
            "roofline": {"bound": "hbm", "achieved": per_launch_bytes / (per_launch_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": per_launch_bytes / (per_launch_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None,
                         "kernel": d["kernel"], "stage": dom, "avg_launch_ms": per_launch_ms, "launches_per_step": d["launches_per_step"],
                         "necessary_bytes_per_launch": per_launch_bytes,
                         "bytes_definition": "HBM bytes the launch has to move in the executed formulation: inputs read once + outputs "
                                             "written once (planner, ga_stats.stage_bytes); HIP-event time on the context's stream",
                         "traffic_note": "PMC traffic of this command (FETCH_SIZE x2 on gfx950, WRITE_SIZE): profiles/"},
            "stages": stages,
            "whole_step": {"necessary_gb": tot_b / 1e9, "kernel_ms": tot_ms, "gb_per_s": tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms else None,
                           "frac_of_hbm_peak": tot_b / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if tot_ms else None},
            "streaming_formulation": {"bytes_per_step": stream_bytes, "tb_per_s_if_streamed": stream_bytes / (dt / args.steps) / 1e12,
                                      "note": "SURVEY.md 8(d) per-block streaming bytes of the reference's algorithm over the step time: "
                                              "the factor of traffic the executed formulation removes, not a roofline fraction"},
            "device_bytes_in_use": st1["device_bytes_in_use"],
        }
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"], rec["parity"] = cpu_baseline_and_parity(voices_total, args.taps, G, args, all_cores)
            rec["parity_rms"] = rec["parity"]["rms_abs"]
            rec["speedup_vs_cpu_1thread"] = value / rec["cpu_baseline"]["value"]
            rec["speedup_vs_cpu_all_cores"] = value / rec["cpu_baseline"]["all_cores"]["value"]
        else:
            rec["cpu_baseline"] = None
        print(json.dumps(rec))
        sys.stdout.flush()
    if use_reduce:
        ctx.CommDestroy()
    ctx.Dispose()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
