#!/usr/bin/env python3
"""bench.py -- rendered frames/sec @48 kHz on the BASELINE.json headline workload.

Workload (config.workload, BASELINE.json configs[2], SURVEY.md section 8d "Config 3"):
    1024 mono voices -> per-voice ConvolverNode sharing one 2-channel 65,536-tap IR (P = 512 partitions of
    128 samples) -> destination (2 ch), 48 kHz, synthetic noise voices + synthetic room-like IR.
A "step" renders `--seconds` (default 10 s = 480,000 frames = 3,750 blocks) of that graph through the C ABI
(ga_render); steps continue the same render (DSP state persists), voices loop over a 10 s buffer so inputs stay
resident in HBM for any number of steps.

Multi-GPU (--gpus N, launched by torch.distributed.run, one rank per GPU): the voices are sharded V/N per rank
(strong scaling: the job is the same 1024-voice mix), every rank renders its shard with ga_render_device and the
destination bus is summed with one RCCL reduce per step (torch.distributed backend "nccl" = RCCL over xGMI).

One JSON line on stdout (rank 0).  Besides the contract keys:
  roofline      -- dominant kernel.  Default path ("formulation C"): the partition sum runs as an overlap-save
                   FFT convolution along the block axis (tconv16_kernel), an HBM-bound stage.  `achieved` follows the
                   contract (SURVEY.md 8(d) per-block STREAMING bytes, 1.086 GB/block here, x blocks per launch / HIP-event
                   launch time) and therefore exceeds the 8 TB/s peak by construction; `traffic` is the HBM bytes per
                   launch measured with rocprofv3 PMC passes (profiles/), and
  roofline_measured_traffic prices the same launch with those measured bytes (the honest distance to the HBM roofline).
  roofline_flops -- the MAC stage in algorithmic TFLOP/s (8*P*129 flop per channel-instance per block) against the f32
                   matrix/vector peak (157.3 TFLOP/s); with --direct (time-batched MFMA GEMM) this is the binding roofline.
  cpu_baseline  -- the CPU oracle (C++ restatement of the reference's single-threaded render path; .NET cannot run
                   here) timed on this host on a bounded sample, scaled to the full workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SR = 48000
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: Peak FP32 (matrix)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E peak


def build_graph(ctx, voices, v0, taps, loop_frames, G):
    from graphaudio_amd import AudioBufferSourceNode, ConvolverNode, PlayableAudioBuffer
    irbuf = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, taps) for c in range(2)], SR)
    ctx.Destination.SetChannelCount(2)
    for v in range(v0, v0 + voices):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(G.voice(v, loop_frames), SR)
        s.Loop = True
        cv = ConvolverNode(ctx)
        cv.Buffer = irbuf
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()


def cpu_baseline(voices_full, taps, G):
    """Time the CPU oracle (1 thread, like the reference) on a bounded sample of the same graph."""
    from tests._oracle import OracleContext
    sample_voices, sample_blocks = 128, 751  # 2 s of 128 voices: ~10-20 s of single-thread CPU work at 65,536 taps
    ctx = OracleContext(SR)
    frames = sample_blocks * 128
    build_graph(ctx, sample_voices, 0, taps, frames + 256, G)
    out = np.zeros((2, frames), np.float32)
    ctx.Render(out, 128)  # first block outside the timed region (queued commands, lazy allocations)
    t0 = time.perf_counter()
    ctx.Render(out, frames - 128, 128)
    dt = time.perf_counter() - t0
    ctx.Dispose()
    fps_sample = (frames - 128) / dt
    fps_full = fps_sample * sample_voices / voices_full  # cost is linear in the voice count
    return {
        "value": fps_full, "unit": "frames/s", "cores": 1, "kind": "port",
        "sample": f"{sample_voices} voices x {sample_blocks - 1} blocks of the same graph ({taps}-tap stereo IR), "
                  f"{dt:.1f} s single thread, {fps_sample:.1f} frames/s measured, scaled by {sample_voices}/{voices_full}",
        "host_cpu": _cpu_name(), "host_cores_available": os.cpu_count(),
    }


def _cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=10.0, help="audio seconds rendered per step")
    ap.add_argument("--voices", type=int, default=1024)
    ap.add_argument("--taps", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--direct", action="store_true", help="use the direct (matrix-core) partition sum instead of the block-axis FFT")
    ap.add_argument("--sync-steps", action="store_true", help="one blocking render per step (no host/device pipelining)")
    ap.add_argument("--no-coarse", action="store_true", help="formulation C (block-axis FFT) instead of D (coarse partitions)")
    ap.add_argument("--force-dist", action="store_true", help="exercise the process-group / RCCL reduce path even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    n_gpus = args.gpus
    dist = None
    import torch
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    assert world == n_gpus or world == 1, "launch with torch.distributed.run --nproc-per-node N for --gpus N"

    from graphaudio_amd import OfflineAudioContext
    from tests import _graphs as G

    frames = int(round(args.seconds * SR)) // 128 * 128
    voices_total = args.voices
    shard = voices_total // world
    v0 = rank * shard
    if rank == world - 1:
        shard = voices_total - v0

    ctx = OfflineAudioContext(SR, device=local_rank)
    ctx.SetOption("profile", 1)
    ctx.SetOption("max_chunk_blocks", 4096)
    if args.direct:
        ctx.SetOption("time_fft", 0)
    if args.no_coarse:
        ctx.SetOption("coarse", 0)
    build_graph(ctx, shard, v0, args.taps, frames, G)

    # the caller's output buffer is page-locked host memory (the D2H copy of the 3.8 MB bus is inside the timed region: a
    # pageable destination costs an extra staging pass per step, INTEGRATION.md "Output buffers")
    host_pin = torch.zeros((2, frames), dtype=torch.float32).pin_memory()
    host_out = host_pin.numpy()
    if use_dist:
        dev_out = torch.zeros((2, frames), dtype=torch.float32, device=f"cuda:{local_rank}")

    # Pipelined steps (ga_synchronize in include/graphaudio_hip.h): a render call returns once its work is enqueued, so the
    # host-side simulation + planning of step k + 1 (0.55 ms at 1024 voices) overlaps the device execution of step k; every
    # step still renders its 10 s, sums the buses and lands in the page-locked host buffer -- the timed region ends with a
    # full synchronisation.  `--sync-steps` restores one blocking render per step.
    pipelined = not args.sync_steps
    pipe_stream = None
    if pipelined:
        ctx.SetOption("async", 1)
        if use_dist:   # render, RCCL reduce and D2H copy of a step are ordered on one (non-default) torch stream
            pipe_stream = torch.cuda.Stream(device=local_rank)
            torch.cuda.synchronize()
            ctx.SetStream(pipe_stream.cuda_stream)

    def dist_step():
        ctx.RenderDevice([dev_out[0].data_ptr(), dev_out[1].data_ptr()], frames)
        dist.reduce(dev_out, dst=0, op=dist.ReduceOp.SUM)   # the destination-bus sum, RCCL over xGMI
        if rank == 0:
            host_pin.copy_(dev_out, non_blocking=pipelined)   # D2H into the page-locked output buffer

    def step():
        if not use_dist:
            ctx.Render(host_out, frames)
        elif pipe_stream is not None:
            with torch.cuda.stream(pipe_stream):
                dist_step()
        else:
            dist_step()

    def sync():
        if pipelined:
            ctx.Synchronize()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    st0 = ctx.GetStats()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enq = time.perf_counter() - t0   # host time to issue all steps (== dt when every step blocks)
    sync()
    dt = time.perf_counter() - t0
    st1 = ctx.GetStats()
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        total_frames = frames * args.steps
        value = total_frames / dt
        d = {k: st1[k] - st0[k] for k in ("mac_ms_total", "mac_flops_total", "mac_bytes_total", "mac_launches",
                                           "fft_ms_total", "other_ms_total", "device_ms_total", "kernel_launches")}
        mac_s = d["mac_ms_total"] * 1e-3
        ach_tflops = d["mac_flops_total"] / mac_s / 1e12 if mac_s > 0 else 0.0
        ach_gbs = d["mac_bytes_total"] / mac_s / 1e9 if mac_s > 0 else 0.0
        blocks = frames // 128
        # HBM bytes per step of the dominant stage from rocprofv3 PMC passes of this exact workload
        # (profiles/r01_pmc_hbm_traffic_v4_mixed_plan.json: separate FETCH_SIZE / WRITE_SIZE passes as MI355X_MICROARCH.md prescribes;
        # the stage is two launches of one kernel template, 4096- and 1024-point segments); null for other shapes
        traffic = None
        try:
            if world == 1 and voices_total == 1024 and args.taps == 65536 and blocks == 3750 and not args.direct:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic_v4_mixed_plan.json")))
                traffic = pm["tconv16_stage_hbm_bytes_per_step"]["total"]
            elif world == 1 and voices_total == 1024 and args.taps == 65536 and blocks == 3750:
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")))
                traffic = pm["spectral_mac_shared_kernel_hbm_bytes_per_launch"]["total"]
        except (OSError, KeyError, ValueError):
            traffic = None
        launches = max(d["mac_launches"], 1)
        avg_ms = d["mac_ms_total"] / launches
        alg_bytes = d["mac_bytes_total"] / launches
        if args.direct:
            kernel = "spectral_mac_shared_kernel (v_mfma_f32_16x16x4_f32, banded-Toeplitz GEMM per bin)"
            form = ("direct partition sum, time-batched on the f32 matrix cores; dense f32 MFMA peak 157.3 TFLOP/s is the "
                    "binding roofline: see roofline_flops")
        else:
            kernel = ("tconv16_kernel<N2> (overlap-save FFT convolution along the block axis: radix 16-16-R Stockham, packed f32, two LDS "
                      "round trips); one step = one 4096-point segment launch + one 1024-point launch, timed together")
            form = ("partition sum evaluated as an FFT convolution over the block index (formulation C, DESIGN.md): ~20x "
                    "fewer flops than the direct sum, so the stage is bound by HBM traffic of the spectra planes")
        rec = {
            "metric": "rendered frames/sec @48kHz, 1024-voice convolver graph",
            "value": value, "unit": "frames/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{voices_total} voices -> PartitionedConvolver, {args.taps}-tap stereo IR shared by all "
                                   f"voices (P={(args.taps + 127) // 128}), 128-sample blocks, 48 kHz, {blocks} blocks per step",
                       "voices": voices_total, "taps": args.taps, "frames_per_step": frames,
                       "parallelism": f"voice-shard x{world} + RCCL bus reduce" if world > 1 else "single GPU",
                       "steps_pipelined": pipelined},
            "realtime_factor": value / SR,
            "host_issue_ms_per_step": t_enq / args.steps * 1e3,
            # `achieved` follows the contract: ALGORITHMIC bytes of the reference's per-block streaming formulation
            # (1.086 GB/block for this workload) x blocks per launch / average launch duration (HIP events on
            # the context's stream).  It exceeds the HBM peak by construction: every spectrum loaded once serves P outputs.
            "roofline": {"bound": "hbm", "achieved": alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
                         "peak": PEAK_HBM_GBS, "unit": "GB/s",
                         "frac": (alg_bytes / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if avg_ms > 0 else 0.0,
                         "traffic": traffic, "kernel": kernel, "avg_launch_ms": avg_ms, "launches": d["mac_launches"],
                         "algorithmic_bytes_per_launch": alg_bytes, "formulation": form},
            # the same launch priced with the HBM bytes it actually moved (PMC): the honest distance to the 8 TB/s roofline
            "roofline_measured_traffic": ({"bound": "hbm", "achieved": traffic / (avg_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS,
                                           "unit": "GB/s", "frac": traffic / (avg_ms * 1e-3) / 1e9 / PEAK_HBM_GBS}
                                          if traffic and avg_ms > 0 else None),
            "roofline_flops": {"bound": "mfma" if args.direct else "valu", "achieved": ach_tflops, "peak": PEAK_F32_MFMA_TFLOPS,
                               "unit": "TFLOP/s", "frac": ach_tflops / PEAK_F32_MFMA_TFLOPS,
                               "note": "algorithmic flops (8*P*129 per channel-instance per block) / MAC-stage kernel time"},
            "kernel_ms_per_step": {"mac_stage": d["mac_ms_total"] / args.steps, "rfft256_fwd_inv": d["fft_ms_total"] / args.steps,
                                   "other": d["other_ms_total"] / args.steps, "device_total": d["device_ms_total"] / args.steps,
                                   "launches": d["kernel_launches"] / args.steps},
            "device_bytes_in_use": st1["device_bytes_in_use"],
            "stage_ms_per_step": {n: (st1["stage_ms"][i] - st0["stage_ms"][i]) / args.steps for i, n in enumerate(
                ("other", "mix", "rfft_fwd", "mac", "rfft_inv", "coarse_fwd", "coarse_mac", "coarse_inv", "coarse_hist"))},
            "stage_gb_per_step": {n: (st1["stage_bytes"][i] - st0["stage_bytes"][i]) / args.steps / 1e9 for i, n in enumerate(
                ("other", "mix", "rfft_fwd", "mac", "rfft_inv", "coarse_fwd", "coarse_mac", "coarse_inv", "coarse_hist"))},
        }
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"] = cpu_baseline(voices_total, args.taps, G)
            rec["speedup_vs_cpu_1thread"] = value / rec["cpu_baseline"]["value"]
        else:
            rec["cpu_baseline"] = None
        print(json.dumps(rec))
    ctx.Dispose()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
