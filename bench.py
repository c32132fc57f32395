#!/usr/bin/env python3
"""bench.py -- rendered frames/sec @48 kHz on the BASELINE.json headline workload.

Workload (config.workload, BASELINE.json configs[2], SURVEY.md section 8d "Config 3"):
    1024 mono voices -> per-voice ConvolverNode sharing one 2-channel 65,536-tap IR -> destination (2 ch), 48 kHz,
    synthetic noise voices + synthetic room-like IR.
A "step" renders `--seconds` (default 10 s = 480,000 frames = 3,750 blocks) of that graph through the C ABI; steps continue
the same render (DSP state persists), voices loop over a 10 s buffer so inputs stay resident in HBM for any number of steps.

Multi-GPU (--gpus N): one rank per GPU.  Launched by `torch.distributed.run --nproc-per-node N` the ranks come from the
environment; launched plainly with --gpus N > 1 this script starts the N rank processes itself (before anything touches a
GPU), watches all of them and relays rank 0's line.  `--scaling strong` (default): the 1024 voices are sharded V/N per rank;
`--scaling weak`: every rank renders `--voices` voices of its own (the job grows with N).  Every rank calls ga_render_reduce:
render its share, ONE RCCL sum of the destination bus per step inside the product library (include/graphaudio_hip.h "sharded
render"), the result lands in rank 0's page-locked host buffer.  The process group (gloo) only carries the communicator id,
the barrier, the max-over-ranks time and the float64 check's input sum.

One JSON line on stdout (rank 0).  Besides the contract keys:
  roofline       -- the dominant kernel of the EXECUTED formulation; its name is what the library reports it ran
                    (ga_stats.stage_kernel).  `achieved` = the HBM bytes the launch HAS to move in that formulation (inputs read
                    once + outputs written once, computed by the planner, ga_stats.stage_bytes) / its average launch duration
                    measured live with HIP events on the context's stream; `peak` = 8 TB/s; `flops_frac` = the planner's count
                    of the floating-point operations the launch executes / that time / 157.3 TFLOP/s; `traffic` = null here (the
                    PMC measurement of the same command, with the guide's gfx950 FETCH_SIZE correction, is in profiles/: it needs
                    its own rocprofv3 passes).
  stages         -- the same numbers for every stage of the step, and `whole_step` for their sum.
  variants       -- the SAME measurement on the graphs / formulations the headline's algebra does not apply to, each with its own
                    dominant kernel and both roofline fractions (a4, the per-voice spectral multiply-accumulate, is measured here):
                      per_voice_spectra : headline graph, every voice transformed, spectra summed (option coarse_premix = 0)
                      private_ir        : every voice its own 65,536-tap stereo impulse response (general multiply-accumulate)
                      config5_1gpu      : BASELINE.json configs[4] whole on one GPU: 512 sources x 16-channel 32,768-tap private IRs
  parity         -- (a) timed_step_rms_vs_f64: the output of the LAST TIMED STEP against float64 mathematics -- the voices loop over
                    exactly one step, so the steady-state bus is the circular convolution of sum_v x_v with the scaled impulse
                    response (one numpy rfft of a step's length); (b) the GPU path against the CPU oracle on the 1 s short form
                    of the same 1024-voice graph.
  streaming_formulation -- SURVEY.md 8(d)'s per-block STREAMING bytes of the reference's algorithm (1.086 GB/block here) over the
                    step time: how much of the reference's traffic the formulation removes (not a roofline fraction).
  cpu_baseline   -- the CPU oracle (C++ restatement of the reference's single-threaded render; .NET cannot run here) timed on
                    this host: 1 thread on the full 1024-voice graph (the reference renders on one thread), and all cores
                    with the voices partitioned across processes.
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
STAGES = ("other", "mix", "rfft_fwd", "mac", "rfft_inv", "coarse_fwd", "coarse_mac", "coarse_inv", "coarse_hist", "coarse_section",
          "coarse_premix")
STAGE_NOTES = {   # what the stage computes (the kernel NAME comes from the library: ga_stats.stage_kernel)
    "mix": "AudioNodeInput.MixBuffer sums",
    "rfft_fwd": "256-point forward transforms (formulations A/B/C)",
    "mac": "partition sum of formulations A/B/C",
    "rfft_inv": "256-point inverse transforms + overlap-add",
    "coarse_premix": "time-domain sum of the voices that share one impulse response (+ their input histories of the next chunk)",
    "coarse_fwd": "16,384-point real transforms of the input windows (two 4096-point complex radix-16 transforms in LDS + combine pass)",
    "coarse_mac": "partition sum over coarse partitions + the consumer's sum in the frequency domain",
    "coarse_inv": "frequency-domain mix + inverse transforms (+ the bus written to the caller's page-locked rows)",
    "coarse_hist": "input history of the next chunk",
}


_voices = {}


def voice(G, v, n):
    """the synthetic voice v (tests/_graphs.py::voice), generated once per process: the main graph and the variants play the same voices"""
    key = (v, n)
    if key not in _voices:
        _voices[key] = G.voice(v, n)
    return _voices[key]


def shared_ir(G, taps):
    return [G.synth_ir(c, taps) for c in range(2)]


def private_ir(G, taps, v):
    return [np.roll(G.synth_ir(c, taps), 37 * v) for c in range(2)]


def build_graph(ctx, voices, v0, taps, loop_frames, G, loop=True, private=False, xsum=None):
    """voices [v0, v0 + voices) of the headline graph; `xsum` (float64, loop_frames) accumulates the voices' samples"""
    from graphaudio_amd import AudioBufferSourceNode, ConvolverNode, PlayableAudioBuffer
    irbuf = PlayableAudioBuffer.FromChannelArrays(shared_ir(G, taps), SR)
    ctx.Destination.SetChannelCount(2)
    for v in range(v0, v0 + voices):
        s = AudioBufferSourceNode(ctx)
        x = voice(G, v, loop_frames) if loop else G.voice(v, loop_frames)
        if xsum is not None:
            xsum += x
        s.Buffer = PlayableAudioBuffer.FromMonoArray(x, SR)
        s.Loop = loop
        cv = ConvolverNode(ctx)
        if private:   # every voice its own impulse response (the general multiply-accumulate path; not the headline)
            irbuf = PlayableAudioBuffer.FromChannelArrays(private_ir(G, taps, v), SR)
        cv.Buffer = irbuf
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()
    return 2


def build_config5(ctx, sources, v0, taps, loop_frames, G, channels=16):
    """BASELINE.json configs[4]: sources -> ConvolverNode with its own 16-channel impulse response -> 16-channel destination"""
    from graphaudio_amd import AudioBufferSourceNode, ConvolverNode, PlayableAudioBuffer
    ctx.Destination.SetChannelCount(channels)
    for v in range(v0, v0 + sources):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(voice(G, v, loop_frames), SR)
        s.Loop = True
        cv = ConvolverNode(ctx)
        cv.Buffer = PlayableAudioBuffer.FromChannelArrays([G.synth_ir(c, taps, seed0=7 + 100 * v) for c in range(channels)], SR)
        s.Connect(cv).Connect(ctx.Destination)
        s.Start()
    return channels


def build_config2(ctx, voices, loop_frames, G):
    """BASELINE.json configs[1]: voices -> BiQuadFilterNode (lowpass, f = 200 * 2^(v/32) Hz capped at 20 kHz, Q 0.707) -> Gain(1/16) ->
    mono mix (SURVEY.md 8d "Config 2"); the voices loop over one step like the headline's"""
    from graphaudio_amd import AudioBufferSourceNode, BiQuadFilterNode, FilterType, GainNode, PlayableAudioBuffer
    ctx.Destination.SetChannelCount(1)
    for v in range(voices):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = PlayableAudioBuffer.FromMonoArray(voice(G, v, loop_frames), SR)
        s.Loop = True
        bq = BiQuadFilterNode(ctx)
        bq.Type = FilterType.Lowpass
        bq.Frequency.Value = min(20000.0, 200.0 * 2.0 ** (v / 32.0))
        bq.Q.Value = 0.707
        g = GainNode(ctx)
        g.Gain.Value = 1.0 / 16.0
        bq.Inputs[0].SetChannelCount(1)
        g.Inputs[0].SetChannelCount(1)
        s.Connect(bq).Connect(g).Connect(ctx.Destination)
        s.Start()
    return 1


def build_config4(ctx, voices, total_frames, G, distinct=64, src_sr=44100):
    """BASELINE.json configs[3] whole on one GPU: voices at 44.1 kHz -> CubicResampler (rate 0.91875) -> 5-band biquad EQ -> gain
    automation -> mix (SURVEY.md 8d "Config 4", tests/_graphs.py::config4_eq).  The voices play through (a looping resampled source
    takes the general-replay path, not the one the configuration names), so the buffers cover every step of the run; to keep 4096 x
    17 s of noise out of the host's memory the voices share `distinct` buffers (voice v plays buffer v mod distinct) -- every voice
    still is a source node with its own resampler, equaliser and gain curve: the device work is that of 4096 distinct voices."""
    from graphaudio_amd import AudioBufferSourceNode, BiQuadFilterNode, FilterType, GainNode, PlayableAudioBuffer
    n_in = int(total_frames * src_sr / SR) + 2048
    bufs = [PlayableAudioBuffer.FromMonoArray(G.voice(v, n_in), src_sr) for v in range(min(distinct, voices))]
    bands = [(FilterType.Lowshelf, 100.0, 1.0, 6.0), (FilterType.Peaking, 400.0, 1.0, -6.0), (FilterType.Peaking, 1000.0, 1.0, 6.0),
             (FilterType.Peaking, 4000.0, 1.0, -6.0), (FilterType.Highshelf, 10000.0, 1.0, 6.0)]
    for v in range(voices):
        s = AudioBufferSourceNode(ctx)
        s.Buffer = bufs[v % len(bufs)]
        node = s
        for (ft, f, q, gdb) in bands:
            bq = BiQuadFilterNode(ctx)
            bq.Type = ft
            bq.Frequency.Value = f
            bq.Q.Value = q
            bq.Gain.Value = gdb
            node = node.Connect(bq)
        g = GainNode(ctx)
        g.Gain.SetValueAtTime(0.0, 0.0)
        g.Gain.LinearRampToValueAtTime(1.0 / 64.0, 0.5)
        g.Gain.SetTargetAtTime(0.0, 8.0, 0.3)
        node.Connect(g).Connect(ctx.Destination)
        s.Start()
    return 2


# the serial recurrences are latency bound, not bandwidth bound: the bound that applies is the dependent-operation
# chain of the reference's direct-form-II section,  w = (x - a1 * w1) - a2 * w2  (BiQuadFilterNode.cs:136-141): from w[n-1] to w[n]
# one multiply and two subtractions that cannot overlap, at the measured latency of a dependent float32 operation
DEP_OP_NS = 3.46   # profiles/r01_micro_dependent_valu_latency.txt: "dependent mul -> add", 8.25 cycles at 2.39 GHz
DEP_OPS_PER_SAMPLE = 3


def serial_bound(frames, kernel_ms, note):
    bound_ms = frames * DEP_OPS_PER_SAMPLE * DEP_OP_NS * 1e-6
    return {"bound": "dependent-operation latency of one cascade walked sample by sample (every cascade in parallel; the sections of a "
                     "cascade pipelined across lanes): frames x 3 dependent float32 operations (mul, sub, sub of the direct-form-II "
                     "recursion, BiQuadFilterNode.cs:136-141) x 3.46 ns (profiles/r01_micro_dependent_valu_latency.txt)",
            "what_the_walk_costs": "measured (profiles/r04_biquad_pipe_probe.txt): a wave that has a SIMD to itself pays ~5.1 cycles per instruction, "
                                   "whatever the instruction; biquad_pipe_kernel's steady state is 8 vector instructions per step + 1.2 for LDS and the loop "
                                   "(47 cycles; the 3-operation chain is 25)",
            "chain_ops_per_sample": DEP_OPS_PER_SAMPLE, "dependent_op_ns": DEP_OP_NS, "frames_per_step": frames,
            "bound_ms_per_step": bound_ms, "kernel_ms_per_step": kernel_ms, "frac": bound_ms / kernel_ms if kernel_ms else None, "note": note}


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
        "build": "oracle/Makefile: g++ -std=c++17 -O3 -mavx2 -ffp-contract=off -fno-fast-math (SURVEY.md 8d names -O2: -O3 is the faster CPU number)",
        "sample": f"all {voices} voices x {blocks - 1} blocks of the bench graph ({taps}-tap stereo IR, 1 s short form), "
                  f"{dt1:.1f} s on one thread, unscaled",
        "all_cores": all_cores,
        "host_cpu": _cpu_name(), "host_cores_available": os.cpu_count(),
    }
    return base, parity


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes (this process never touches a GPU), watch ALL of
    them, and take the others down when one fails -- a rank that dies early would otherwise leave its peers waiting in the
    rendezvous or in the collective for ever."""
    import socket
    if any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD", "HSA_TOOLS_LIB")):
        sys.exit("bench.py: refusing to start rank processes under a profiler preload (the preloaded library has initialised the GPU in "
                 "this process; profile one rank: --gpus 1)")
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
    import threading
    out = []
    reader = threading.Thread(target=lambda: out.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    live = set(range(n))
    while live and rc == 0:
        time.sleep(0.2)
        for r in list(live):
            code = procs[r].poll()
            if code is not None:
                live.discard(r)
                if code != 0:
                    rc = code
                    sys.stderr.write(f"bench.py: rank {r} exited with code {code}; stopping the other ranks\n")
    if rc != 0:
        for r in live:
            procs[r].terminate()
        deadline = time.time() + 10
        for r in live:
            try:
                procs[r].wait(max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                procs[r].kill()
    reader.join(5)
    sys.stdout.write(b"".join(out).decode())
    sys.exit(rc)


# ------------------------------------------------------------------------------------------------------------------------
def stage_table(st0, st1, steps):
    """per-stage numbers of the timed region.  Times come from the chunks that recorded events (every --profile-every-th), scaled
    to a step; launches, bytes and flops count every chunk."""
    nprof = max(st1["profiled_chunks"] - st0["profiled_chunks"], 1)
    per_step = (st1["chunks"] - st0["chunks"]) / steps / nprof
    stages = {}
    tot_ms = tot_b = 0.0
    for i, name in enumerate(STAGES):
        ms = (st1["stage_ms"][i] - st0["stage_ms"][i]) * per_step
        nl = (st1["stage_launches"][i] - st0["stage_launches"][i]) / steps
        by = (st1["stage_bytes"][i] - st0["stage_bytes"][i]) / steps
        fl = (st1["stage_flops"][i] - st0["stage_flops"][i]) / steps
        if nl <= 0:
            continue
        gbs = by / (ms * 1e-3) / 1e9 if ms > 0 and by > 0 else None
        tfs = fl / (ms * 1e-3) / 1e12 if ms > 0 and fl > 0 else None
        stages[name] = {"ms_per_step": ms, "launches_per_step": nl, "necessary_gb_per_step": by / 1e9 if by > 0 else None,
                        "gb_per_s": gbs, "frac_of_hbm_peak": gbs / PEAK_HBM_GBS if gbs else None,
                        "gflop_per_step": fl / 1e9 if fl > 0 else None, "tflop_per_s": tfs,
                        "frac_of_f32_peak": tfs / PEAK_F32_TFLOPS if tfs else None,
                        "kernel": st1["stage_kernel"][i] or name, "computes": STAGE_NOTES.get(name, name)}
        if by > 0 and name != "coarse_section":
            tot_ms += ms
            tot_b += by
    whole = {"necessary_gb": tot_b / 1e9, "kernel_ms": tot_ms, "gb_per_s": tot_b / (tot_ms * 1e-3) / 1e9 if tot_ms else None,
             "frac_of_hbm_peak": tot_b / (tot_ms * 1e-3) / 1e9 / PEAK_HBM_GBS if tot_ms else None}
    return stages, whole, nprof, per_step


def roofline_of(stages):
    cands = [n for n in stages if stages[n]["necessary_gb_per_step"] and n != "coarse_section"]
    if not cands:
        return None
    dom = max(cands, key=lambda n: stages[n]["ms_per_step"])
    d = stages[dom]
    per_launch_ms = d["ms_per_step"] / d["launches_per_step"]
    per_launch_bytes = d["necessary_gb_per_step"] * 1e9 / d["launches_per_step"]
    ach = per_launch_bytes / (per_launch_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": ach, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": ach / PEAK_HBM_GBS, "traffic": None}
    if "mfma" in d["kernel"] and (d["frac_of_f32_peak"] or 0.0) > r["frac"]:
        # a kernel on the matrix cores whose flop fraction is the larger one is priced against the dense f32 MFMA peak
        r = {"bound": "mfma", "achieved": d["tflop_per_s"], "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s", "frac": d["frac_of_f32_peak"], "traffic": None,
             "hbm_gb_per_s": ach, "hbm_frac": ach / PEAK_HBM_GBS}
    r.update({"kernel": d["kernel"], "computes": d["computes"], "stage": dom, "avg_launch_ms": per_launch_ms,
            "launches_per_step": d["launches_per_step"], "necessary_bytes_per_launch": per_launch_bytes,
            "flops_per_launch": (d["gflop_per_step"] or 0.0) * 1e9 / d["launches_per_step"],
            "tflop_per_s": d["tflop_per_s"], "flops_frac": d["frac_of_f32_peak"], "f32_peak_tflops": PEAK_F32_TFLOPS,
            "bytes_definition": "HBM bytes the launch has to move in the executed formulation: inputs read once + outputs "
                                "written once (planner, ga_stats.stage_bytes); HIP-event time on the context's stream",
            "traffic_note": "PMC traffic of this command (FETCH_SIZE x2 on gfx950, WRITE_SIZE): profiles/"})
    return r


def attach_pmc_traffic(roof, profile_json):
    """`traffic` of the roofline object: HBM bytes per launch of the dominant kernel from the PMC passes committed under profiles/
    (rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes of this command, FETCH_SIZE x 2 on gfx950: tools/collect_profiles.sh).
    bench.py cannot collect counters itself; the committed figure is attached only when it is for the same kernel moving the same
    necessary bytes (within 2 %), and the line says where it comes from.  Nothing under profiles/ is needed to run the bench."""
    try:
        if not roof or not os.path.exists(profile_json):
            return roof
        prof = json.load(open(profile_json))
        for name, k in prof.get("kernels", {}).items():
            nec = k.get("necessary_gb_per_launch (stage, planner)")
            if name.split("<")[0] in roof["kernel"] and nec and abs(nec * 1e9 / roof["necessary_bytes_per_launch"] - 1.0) < 0.02 and k.get("pmc_total_x2_gb"):
                roof["traffic"] = k["pmc_total_x2_gb"] * 1e9
                roof["traffic_measured_in_this_run"] = False   # (a committed measurement of the same command; counters need their own passes)
                roof["traffic_source"] = (os.path.relpath(profile_json, ROOT) + ": rocprofv3 --pmc FETCH_SIZE (x 2: gfx950 correction) + WRITE_SIZE, "
                                          "separate passes of `" + str(prof.get("command", "bench.py")).strip() + "`, bytes per launch")
                break
    except Exception:   # a missing or malformed profile never fails the bench
        pass
    return roof


def timed_steps(ctx, step, sync, steps, warmup):
    for _ in range(warmup):
        step()
    sync()
    st0 = ctx.GetStats()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    t_enq = time.perf_counter() - t0   # host time to issue all steps (== dt when every step blocks)
    sync()
    dt = time.perf_counter() - t0
    return dt, t_enq, st0, ctx.GetStats()


def circular_truth(xsum, irs):
    """steady-state bus of looping voices: circular convolution (period = the loop) of the summed input with the scaled taps"""
    from tests import _f64model as M
    return np.stack([M.circular_conv(xsum, M.scaled_ir64(h)) for h in irs])


def _truth_worker(job):
    """float64 truth of a variant whose voices have impulse responses of their own, for voices [v0, v1): the spectrum of
    sum_v circ(x_v, h_{v,c}) (period = one step), accumulated in the frequency domain.  Runs in worker processes before the parent
    touches the GPU."""
    kind, v0, v1, frames, taps = job
    from tests import _f64model as M
    from tests import _graphs as G
    channels = 16 if kind == "config5_1gpu" else 2
    acc = np.zeros((channels, frames // 2 + 1), np.complex128)
    h = np.zeros(frames, np.float64)
    for v in range(v0, v1):
        X = np.fft.rfft(G.voice(v, frames).astype(np.float64))
        irs = private_ir(G, taps, v) if kind == "private_ir" else [G.synth_ir(c, taps, seed0=7 + 100 * v) for c in range(channels)]
        for c in range(channels):
            hc = M.scaled_ir64(irs[c])
            h[:len(hc)] = hc
            acc[c] += X * np.fft.rfft(h)
    return acc


def precompute_truths(args, frames, voices_total):
    """the variants' float64 truths (circular convolutions, see circular_truth), on all cores, before the GPU is initialised"""
    import multiprocessing as mp
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, 16))
    cases = {"private_ir": (min(voices_total, 1024), args.taps), "config5_1gpu": (512, 32768)}
    jobs = []
    for kind, (nv, taps) in cases.items():
        per = max(1, (nv + 4 * cores - 1) // (4 * cores))
        jobs += [(kind, a, min(nv, a + per), frames, taps) for a in range(0, nv, per)]
    with mp.get_context("spawn").Pool(cores) as pool:
        parts = pool.map(_truth_worker, jobs, chunksize=1)
    out = {}
    for kind in cases:
        acc = sum(p for j, p in zip(jobs, parts) if j[0] == kind)
        out[kind] = np.fft.irfft(acc, n=frames, axis=1)
    return out


def oracle_short_form(build, channels, frames, options=None):
    """the CPU oracle (one thread) and the device on the short form of a variant's graph: (frames/s of the oracle, rms error, bus rms)"""
    from graphaudio_amd import OfflineAudioContext
    from tests import _graphs as G
    from tests._oracle import OracleContext
    o = OracleContext(SR)
    build(o)
    ref = np.zeros((channels, frames), np.float32)
    o.Render(ref, 128)
    t0 = time.perf_counter()
    o.Render(ref, frames - 128, 128)
    dt = time.perf_counter() - t0
    o.Dispose()
    h = OfflineAudioContext(SR)
    for k, v in (options or {}).items():
        h.SetOption(k, v)
    build(h)
    got = np.zeros_like(ref)
    h.Render(got, frames)
    st = h.GetStats()
    h.Dispose()
    return (frames - 128) / dt, dt, G.rms(ref - got), G.rms(ref), st


def run_variant(name, torch, G, frames, steps, warmup, build, channels, options, describe, truth=None, extra=None):
    """one more measurement on a context of its own: same step loop, same stage table, its own dominant kernel"""
    from graphaudio_amd import OfflineAudioContext
    from tests import _f64model as M
    t_build = time.perf_counter()
    ctx = OfflineAudioContext(SR)
    ctx.SetOption("profile", 1)
    ctx.SetOption("profile_every", 1)
    ctx.SetOption("max_chunk_blocks", 4096)
    ctx.SetOption("async", 1)
    for k, v in options.items():
        ctx.SetOption(k, v)
    build(ctx)
    host = torch.zeros((channels, frames), dtype=torch.float32).pin_memory()
    out = host.numpy()
    t_build = time.perf_counter() - t_build

    def sync():
        ctx.Synchronize()
        torch.cuda.synchronize()
    dt, t_enq, st0, st1 = timed_steps(ctx, lambda: ctx.Render(out, frames), sync, steps, warmup)
    stages, whole, _, per_step = stage_table(st0, st1, steps)
    rec = {"description": describe, "ms_per_step": dt / steps * 1e3, "frames_per_s": frames * steps / dt, "steps": steps, "warmup": warmup,
           "frames_per_step": frames, "host_issue_ms_per_step": t_enq / steps * 1e3,
           "device_ms_per_step": (st1["device_ms_total"] - st0["device_ms_total"]) * per_step,
           "roofline": attach_pmc_traffic(roofline_of(stages), os.path.join(ROOT, "profiles", {"per_voice_spectra": "r03_per_voice", "private_ir": "r03_private_ir",
                                                                                                   "config5_1gpu": "r03_config5"}.get(name, "none") + "_pmc_hbm_traffic.json")),
           "stages": stages, "whole_step": whole, "setup_s": t_build}
    if truth is not None:
        ref = truth()
        err, sig = M.rms(out - ref), M.rms(ref)
        rec["timed_step_rms_vs_f64"] = {"rms_abs": err, "rms_relative_to_bus": err / sig, "bus_rms": sig, "tolerance_rms_abs": 1e-5}
    if extra is not None:
        rec.update(extra(rec, st0, st1))
    ctx.Dispose()
    del host
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--seconds", type=float, default=10.0, help="audio seconds rendered per step")
    ap.add_argument("--voices", type=int, default=1024)
    ap.add_argument("--taps", type=int, default=65536)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="strong: --voices in total, sharded over the ranks; weak: --voices per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-variants", action="store_true", help="skip the per-voice / private-IR / config 5 variants of the default run")
    ap.add_argument("--no-check", action="store_true", help="skip the float64 check of the last timed step")
    ap.add_argument("--variant-steps", type=int, default=12)
    ap.add_argument("--only-variant", choices=["per_voice_spectra", "private_ir", "config5_1gpu"], default="",
                    help="measure just this variant (--steps / --warmup apply) and print ITS record: profiling runs (tools/collect_profiles.sh)")
    ap.add_argument("--baseline-blocks", type=int, default=375)
    ap.add_argument("--baseline-cores", type=int, default=0)
    ap.add_argument("--direct", action="store_true", help="direct (matrix-core) partition sum, formulation A")
    ap.add_argument("--no-coarse", action="store_true", help="formulation C (block-axis FFT) instead of D (coarse partitions)")
    ap.add_argument("--no-premix", action="store_true", help="formulation D without the time-domain pre-mix: every voice is transformed, the spectra are summed")
    ap.add_argument("--overlap", action="store_true", help="formulation D: forward and multiply-accumulate stages concurrently on two streams (measured slower)")
    ap.add_argument("--profile-every", type=int, default=8, help="record the per-stage HIP events on every k-th chunk of the timed region "
                    "(a chunk that records them runs ~0.08 ms longer: every event is a barrier between two launches)")
    ap.add_argument("--no-profile", action="store_true", help="no per-stage HIP events (measurement of their cost; the roofline object is then empty)")
    ap.add_argument("--private-ir", action="store_true", help="every voice convolves with its own impulse response (general path; measurement, not the headline config)")
    ap.add_argument("--no-host-direct", action="store_true", help="copy the bus to the host with copy kernels instead of writing it from the last kernel (measurement)")
    ap.add_argument("--no-tail", action="store_true", help="formulation D: no carried output tails, every chunk re-transforms the input history (measurement)")
    ap.add_argument("--no-carry", action="store_true", help="formulation D: copy the input history with its own kernel instead of from the forward transforms (measurement)")
    ap.add_argument("--copy-stream", action="store_true", help="hand the bus to the host on a copy stream of its own (measurement)")
    ap.add_argument("--sync-steps", action="store_true", help="one blocking render per step (no host/device pipelining)")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE", help="ga_set_option on the main context (measurements)")
    ap.add_argument("--library", default="", help="load this build of the library instead of the product (tools/build_variant.sh; measurements)")
    ap.add_argument("--force-dist", action="store_true", help="exercise the sharded-render path (ga_render_reduce) even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        spawn_ranks(args.gpus)   # does not return
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")

    frames = int(round(args.seconds * SR)) // 128 * 128
    if args.only_variant:
        import torch
        if args.library:
            from graphaudio_amd import _capi
            _capi.use_library(os.path.abspath(args.library))
        from tests import _graphs as G
        builds = {"per_voice_spectra": (lambda c: build_graph(c, args.voices, 0, args.taps, frames, G), 2, {"coarse_premix": 0}),
                  "private_ir": (lambda c: build_graph(c, args.voices, 0, args.taps, frames, G, private=True), 2, {}),
                  "config5_1gpu": (lambda c: build_config5(c, 512, 0, 32768, frames, G), 16, {})}
        build, ch, opts = builds[args.only_variant]
        for kv in args.opt:
            k, v = kv.split("=", 1)
            opts[k] = float(v)
        print(json.dumps(run_variant(args.only_variant, torch, G, frames, args.steps, args.warmup, build, ch, opts, args.only_variant)))
        return
    all_cores = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        all_cores = cpu_all_cores(args.voices, args.taps, args)   # before anything initialises the GPU in this process
    with_variants = (rank == 0 and world == 1 and not args.no_variants
                     and not (args.private_ir or args.no_premix or args.no_coarse or args.direct or args.no_profile))
    truths = precompute_truths(args, frames, args.voices) if with_variants and not args.no_check else None

    import torch   # (first: its bundled HIP runtime has to be the one in the process, tests/conftest.py)
    dist = None
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        # control plane only; the bus sum is RCCL inside the library.  A finite timeout: a peer that died must not hang this rank.
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))

    if args.library:
        from graphaudio_amd import _capi
        _capi.use_library(os.path.abspath(args.library))
    from graphaudio_amd import OfflineAudioContext
    from graphaudio_amd.distributed import init_sharded, shard_range
    from tests import _graphs as G

    if args.scaling == "weak":
        voices_total = args.voices * world
        v0, v1 = rank * args.voices, (rank + 1) * args.voices
    else:
        voices_total = args.voices
        v0, v1 = shard_range(voices_total, world, rank)
    use_reduce = world > 1 or args.force_dist

    ctx = OfflineAudioContext(SR, device=local_rank)
    ctx.SetOption("profile", 0 if args.no_profile else 1)
    # the events cost device time: only every k-th chunk records them (every chunk when the run is too short to sample)
    pe = args.profile_every
    while pe > 1 and args.steps < 2 * pe:
        pe //= 2
    ctx.SetOption("profile_every", pe)
    ctx.SetOption("max_chunk_blocks", 4096)
    if args.direct:
        ctx.SetOption("time_fft", 0)
    if args.no_coarse:
        ctx.SetOption("coarse", 0)
    if args.no_premix:
        ctx.SetOption("coarse_premix", 0)
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
    for kv in args.opt:
        k, v = kv.split("=", 1)
        ctx.SetOption(k, float(v))
    check = not args.no_check and not args.no_profile
    xsum = np.zeros(frames, np.float64) if check and not args.private_ir else None
    build_graph(ctx, v1 - v0, v0, args.taps, frames, G, private=args.private_ir, xsum=xsum)
    comm_info = None
    if use_reduce:
        init_sharded(ctx, rank, world)
        # what RCCL itself says (ncclCommCount / ncclCommUserRank): a run whose ranks did not join ONE communicator of `world` ranks
        # is not an N-GPU measurement, whatever WORLD_SIZE says
        try:
            comm_info = ctx.CommInfo()
        except Exception as e:   # (an RCCL without ncclCommCount: the line then says so instead of a rank count)
            comm_info = {"error": str(e)}
        if "error" not in comm_info and (comm_info["ranks"] != world or comm_info["rank"] != rank or (world > 1 and not comm_info["uses_rccl"])):
            sys.exit(f"bench.py: rank {rank}: the communicator reports {comm_info}, expected {world} ranks")

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

    dt, t_enq, st0, st1 = timed_steps(ctx, step, sync, args.steps, args.warmup)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if xsum is not None:   # the float64 check needs the sum over ALL ranks' voices
            xs = torch.from_numpy(xsum)
            dist.reduce(xs, dst=0, op=dist.ReduceOp.SUM)

    if rank == 0:
        from tests import _f64model as M
        value = frames * args.steps / dt
        blocks = frames // 128
        stages, whole, nprof, per_step = stage_table(st0, st1, args.steps)
        if args.no_profile or not stages:   # --no-profile (measurement of the events' cost): nothing to price
            print(json.dumps({"ms_per_step": dt / args.steps * 1e3, "value": value, "host_issue_ms_per_step": t_enq / args.steps * 1e3,
                              "device_ms_per_step": 0.0, "stages": {}}))
            return
        premixed = "coarse_premix" in stages
        form = ("formulation D (DESIGN.md): overlap-save with coarse partitions of 8192 samples (8 per 65,536-tap IR instead of 512), "
                "16,384-point real transforms in LDS"
                + (", the voices that share the impulse response summed in the TIME domain in front of ONE set of transforms "
                   "(sum_v (x_v * h) = (sum_v x_v) * h; planner decision, option coarse_premix; the per-voice route is variants.per_voice_spectra)"
                   if premixed else
                   " per voice, the destination sum fused in the frequency domain"
                   + (" (voices that share the impulse response: their spectra are summed before the spectral multiply)" if not args.private_ir else
                      " (a private impulse response per voice: every product evaluated)"))
                + ", output tails carried from step to step instead of re-transforming the input history" * (not args.no_tail)
                if "coarse_fwd" in stages else
                "formulation C: partition sum as an FFT convolution along the block axis" if not args.direct else
                "formulation A: direct partition sum on the f32 matrix cores")
        dev_ms = (st1["device_ms_total"] - st0["device_ms_total"]) * per_step
        stream_bytes = (st1["mac_bytes_total"] - st0["mac_bytes_total"]) / args.steps
        rec = {
            "metric": "rendered frames/sec @48kHz, 1024-voice convolver graph",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{voices_total} voices -> PartitionedConvolver, {args.taps}-tap stereo IR {'of its own per voice' if args.private_ir else 'shared by all voices'} "
                                   f"(P={(args.taps + 127) // 128}), 128-sample blocks, 48 kHz, {blocks} blocks per step",
                       "voices": voices_total, "voices_per_gpu": v1 - v0, "taps": args.taps, "frames_per_step": frames,
                       "parallelism": f"voice-shard x{world}, one RCCL reduce of the bus per step (ga_render_reduce)" if world > 1 else "single GPU",
                       "steps_pipelined": pipelined, "formulation": form},
            "realtime_factor": value / SR,
            "host_issue_ms_per_step": t_enq / args.steps * 1e3,
            "device_ms_per_step": dev_ms,
            "profiled_chunks": nprof,
            "roofline": attach_pmc_traffic(roofline_of(stages), os.path.join(ROOT, "profiles", "r04_cfg3_pmc_hbm_traffic.json")),
            "stages": stages,
            "whole_step": whole,
            "streaming_formulation": {"bytes_per_step": stream_bytes, "tb_per_s_if_streamed": stream_bytes / (dt / args.steps) / 1e12,
                                      "note": "SURVEY.md 8(d) per-block streaming bytes of the reference's algorithm over the step time: "
                                              "the factor of traffic the executed formulation removes, not a roofline fraction"},
            "device_bytes_in_use": st1["device_bytes_in_use"],
            # the communicator's own rank count (ga_comm_info -> ncclCommCount), checked against WORLD_SIZE on every rank above; null = no
            # communicator in this run (one GPU).  No multi-GPU hardware was available to the builder: N > 1 is unmeasured until the driver runs it.
            "comm": comm_info,
        }
        parity = {}
        if check and args.warmup + args.steps >= 2 and frames >= args.taps:
            # the output of the LAST TIMED STEP, as it sits in the caller's page-locked rows, against float64 mathematics
            if args.private_ir:
                truth = np.fft.irfft(sum(_truth_worker(("private_ir", a, min(voices_total, a + 64), frames, args.taps))
                                         for a in range(0, voices_total, 64)), n=frames, axis=1)
            else:
                truth = circular_truth(xsum, shared_ir(G, args.taps))
            err, sig = M.rms(host_out - truth), M.rms(truth)
            parity["timed_step_rms_vs_f64"] = {
                "rms_abs": err, "rms_relative_to_bus": err / sig, "bus_rms": sig, "tolerance_rms_abs": 1e-5,
                "what": f"the {frames}-frame output of the last timed step (all {voices_total} voices, all ranks) vs the circular convolution of "
                        "the summed looping voices with the scaled impulse response in float64 (tests/_f64model.py)"}
            rec["timed_step_rms_vs_f64"] = err
        if not args.no_cpu_baseline and world == 1:
            rec["cpu_baseline"], vs_oracle = cpu_baseline_and_parity(voices_total, args.taps, G, args, all_cores)
            parity.update(vs_oracle)
            rec["parity_rms"] = vs_oracle["rms_abs"]
            rec["speedup_vs_cpu_1thread"] = value / rec["cpu_baseline"]["value"]
            rec["speedup_vs_cpu_all_cores"] = value / rec["cpu_baseline"]["all_cores"]["value"]
        else:
            rec["cpu_baseline"] = None
        rec["parity"] = parity or None
    if use_reduce:
        ctx.CommDestroy()
    ctx.Dispose()
    del host_pin
    if rank == 0:
        if with_variants:
            # the graphs / formulations the headline's algebra does not apply to, measured the same way
            vs, vw = args.variant_steps, 3
            variants = {}
            sf = int(2.5 * SR) // 128 * 128   # 2.5 s steps: these graphs play through (no looping), the buffers cover every step
            ss, sw = 6, 2

            def serial_extra(build_short, ch, short_frames, voices_n, what, bound_frames, options=None):
                def extra(r, st0, st1):
                    fps, dt, err, sig, _ = oracle_short_form(build_short, ch, short_frames, options)
                    other = r["stages"].get("other", {}).get("ms_per_step", 0.0) + r["stages"].get("mix", {}).get("ms_per_step", 0.0)
                    if r.get("roofline"):
                        r["roofline"]["note"] = ("the HBM-bound stage of this graph; its dominant stage by time is 'other' (serial recurrences / launch-bound "
                                                 "elementwise kernels), which no bandwidth roof prices: see serial_bound")
                    return {"parity_vs_oracle": {"rms_abs": err, "rms_relative_to_bus": err / max(sig, 1e-30), "bus_rms": sig, "tolerance_rms_abs": 1e-5,
                                                 "sample": f"{voices_n} voices x {short_frames // 128} blocks (short form of this graph) vs the CPU oracle"},
                            "cpu_baseline": {"value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
                                             "sample": f"{voices_n} voices x {short_frames // 128 - 1} blocks of this graph, {dt:.1f} s on one thread, unscaled"},
                            "speedup_vs_cpu_1thread": r["frames_per_s"] / fps,
                            "host_ms_vs_device_ms": {"host_issue_ms_per_step": r["host_issue_ms_per_step"], "device_ms_per_step": r["device_ms_per_step"]},
                            "serial_bound": serial_bound(bound_frames, other, what)}
                return extra
            # (config 4 first: 28,672 node records planned per step -- the host side of that variant is bound by cache misses, and it
            # measured 7.2 instead of 3.9 ms per step when the records were allocated from the heap the earlier variants' contexts left behind)
            variants["config4_eq_1gpu"] = run_variant(
                "config4_eq_1gpu", torch, G, sf, 14, sw, lambda c: build_config4(c, 4096, sf * (14 + sw), G), 2, {},   # (14 steps: with 6 the one chunk that drains the pipeline at the end was a seventh of the time)
                "BASELINE.json configs[3] whole on ONE GPU: 4096 voices, 44.1 kHz -> CubicResampler -> 5-band biquad EQ -> gain automation -> mix "
                "(its 8-GPU form shards 512 voices per GPU); 2.5 s steps of one continuing render; 64 distinct noise buffers shared by the voices",
                extra=serial_extra(lambda c: G.config4_eq(c, voices=4096, frames=128 * 94), 2, 128 * 94, 4096,
                                   "the planner refuses to split this equaliser (its 100 Hz low shelf carries ~1e-4 of round-off noise in the reference's "
                                   "own arithmetic: any re-association leaves the 1e-5 contract, Context::biquadDeviation): one walk, five sections pipelined "
                                   "across lanes (biquad_pipe_kernel<5>)", sf))
            variants["per_voice_spectra"] = run_variant(
                "per_voice_spectra", torch, G, frames, vs, vw, lambda c: build_graph(c, voices_total, 0, args.taps, frames, G), 2,
                {"coarse_premix": 0},
                "the headline graph with option coarse_premix = 0: every voice's input transformed, the spectra of the voices summed "
                "before ONE spectral multiply (round 2's default path)",
                truth=(lambda: circular_truth(xsum, shared_ir(G, args.taps))) if xsum is not None else None)
            pv = min(voices_total, 1024)
            variants["private_ir"] = run_variant(
                "private_ir", torch, G, frames, vs, vw, lambda c: build_graph(c, pv, 0, args.taps, frames, G, private=True), 2, {},
                f"{pv} voices, each through its OWN {args.taps}-tap stereo impulse response: every voice's spectral multiply-accumulate is "
                "evaluated (SURVEY.md 8(d) 'unique-IR-per-voice variant'; row a4)",
                truth=(lambda: truths["private_ir"]) if truths else None)
            variants["config5_1gpu"] = run_variant(
                "config5_1gpu", torch, G, frames, vs, vw, lambda c: build_config5(c, 512, 0, 32768, frames, G), 16, {},
                "BASELINE.json configs[4] whole on ONE GPU: 512 sources x 16-channel 32,768-tap private impulse responses -> 16-channel bus "
                "(its 8-GPU form shards 64 sources per GPU)",
                truth=(lambda: truths["config5_1gpu"]) if truths else None)
            # ---- the latency-bound configurations (BASELINE.json configs[1], configs[3]) and the Kit scene, VERDICT r3 item 3
            variants["config2_biquad"] = run_variant(
                "config2_biquad", torch, G, frames, vs, vw, lambda c: build_config2(c, 256, frames, G), 1, {},
                "BASELINE.json configs[1]: 256 mono voices -> BiQuadFilterNode (lowpass) -> Gain -> mono mix; latency bound, not bandwidth bound",
                extra=serial_extra(lambda c: G.config2_biquad(c, voices=256, frames=SR + 256), 1, SR // 128 * 128, 256,
                                   "the planner splits these well-conditioned low-passes along time (option biquad_time_split: pass A / scan / pass B "
                                   "over G pieces, DESIGN.md section 4), which cuts the chain the bound is written for: frac > 1 says by how much",
                                   frames))
            variants["kit_scene"] = run_variant(
                "kit_scene", torch, G, sf, ss, sw, lambda c: G.kit_scene(c, voices=256, frames=sf * (ss + sw) + 256, taps=65536), 2, {},
                "SURVEY.md 8(f) rank 4: 256 panned voices -> two buses (one fading) -> master -> ReverbEffect shape (dry / down-mix -> "
                "65,536-tap convolver -> wet) -> destination; one post-mix convolver, launch bound; 2.5 s steps of one continuing render",
                extra=serial_extra(lambda c: G.kit_scene(c, voices=256, frames=SR + 256, taps=65536), 2, SR // 128 * 128, 256,
                                   "no serial recurrence in this graph: the entry only relates the elementwise stages' time to the same yardstick", sf))
            # ---- formulation R: the headline's partition sum in the reference's own order and arithmetic, at a size it finishes
            #      in milliseconds: what the route costs where the planner needs it, priced against the non-fused float32 VALU rate
            rv = 64

            def ref_extra(r, st0, st1):
                fps, dt, err, sig, st = oracle_short_form(lambda c: G.config3_convolver(c, voices=rv, taps=args.taps, frames=128 * 94), 2, 128 * 94,
                                                          {"conv_reference_order": 2})
                from tests import _graphs as Gm
                mac = r["stages"].get("mac", {})
                ops_peak = PEAK_F32_TFLOPS / 2.0   # (no fma: one operation per lane and issue slot)
                return {"parity_vs_oracle": {"rms_abs": err, "rms_relative_to_bus": err / max(sig, 1e-30), "bus_rms": sig, "tolerance_rms_abs": 1e-5,
                                             "ref_order_rows": st["ref_order_rows"],
                                             "sample": f"{rv} voices x 94 blocks of this graph vs the CPU oracle (not all 512 partitions live yet)"},
                        "cpu_baseline": {"value": fps, "unit": "frames/s", "cores": 1, "kind": "port",
                                         "sample": f"{rv} voices x 93 blocks, {dt:.1f} s on one thread, unscaled"},
                        "speedup_vs_cpu_1thread": r["frames_per_s"] / fps,
                        "valu_roofline": {"bound": "f32 VALU without fma", "peak_top_per_s": ops_peak, "achieved_top_per_s": mac.get("tflop_per_s"),
                                          "frac": (mac.get("tflop_per_s") or 0.0) / ops_peak, "kernel": mac.get("kernel")}}
            variants["reference_order_64_voices"] = run_variant(
                "reference_order_64_voices", torch, G, frames, 6, 2, lambda c: build_graph(c, rv, 0, args.taps, frames, G), 2, {"conv_reference_order": 2},
                f"formulation R forced on {rv} voices of the headline graph: double-precision 256-point transforms around the reference's own "
                "sequential float32 partition sum (refmac_kernel); the route the planner takes where last-bit differences would be amplified downstream",
                truth=None, extra=ref_extra)
            rec["variants"] = variants
        print(json.dumps(rec))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
