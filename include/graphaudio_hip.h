/*
 * graphaudio_hip.h -- C ABI of libgraphaudio_hip.so, the MI355X-native offline render path
 * for GraphAudio's per-block DSP hot path.
 *
 * This is the drop-in boundary.  Every entry point is a flat C mirror
 * of one public member of the reference's AudioContextBase / AudioNode / AudioParam surface;
 * the reference member each one replaces is cited as (file:line), relative to the reference
 * repository root.  A managed host (C# [LibraryImport], see INTEGRATION.md and
 * bindings/csharp/) or the Python ctypes host in graphaudio_amd/ replays graph construction
 * through these calls and then asks for whole renders: O(1) native calls per Render().
 *
 * Conventions (same as the reference's own P/Invoke layers, GraphAudio.IO/Libsndfile.cs:36-68,
 * GraphAudio.Realtime/Miniaudio.cs:303-349): cdecl, opaque handle, int result codes with
 * 0 = success and negatives = errors, error-string getter, every pointer argument borrowed
 * for the duration of the call only.  One context is driven by one thread at a time
 * (AudioContextBase.cs:59-62); different contexts may be used concurrently.
 *
 * The same header is compiled by the CPU oracle (oracle/ga_oracle.cpp) with
 * -DGA_FN(n)=gao_##n so the test-only oracle exposes the identical surface under a gao_ prefix.
 */
#ifndef GRAPHAUDIO_HIP_H
#define GRAPHAUDIO_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef GA_FN
#define GA_FN(name) ga_##name
#endif

#if defined(__GNUC__)
#define GA_EXPORT __attribute__((visibility("default")))
#else
#define GA_EXPORT
#endif

/* ---- result codes: what the managed wrapper turns back into the reference's exceptions ---- */
#define GA_OK 0
#define GA_ERR_INVALID_ARGUMENT (-1)  /* ArgumentException           (OfflineAudioContext.cs:32-51) */
#define GA_ERR_OUT_OF_RANGE (-2)      /* ArgumentOutOfRangeException (AudioBuffer.cs:18, AudioNodeInput.cs:43) */
#define GA_ERR_INVALID_OPERATION (-3) /* InvalidOperationException   (ConvolverNode.cs:45-49, AudioBufferSourceNode.cs:83-90) */
#define GA_ERR_DISPOSED (-4)          /* ObjectDisposedException     (AudioContextBase.cs:54,268) */
#define GA_ERR_CYCLE (-5)             /* InvalidOperationException "Audio graph cycle detected" (Nodes/AudioNode.cs:157-160): unreachable in the
                                         reference (its memo check returns first) and no longer returned here -- loops render with the
                                         reference's one-block stale buffer */
#define GA_ERR_UNSUPPORTED (-6)       /* graph uses a feature outside the accelerated path: host must fall back to the CPU context */
#define GA_ERR_DEVICE (-7)            /* HIP runtime error */
#define GA_ERR_OUT_OF_MEMORY (-8)
#define GA_ERR_NO_DEVICE (-9)         /* no gfx950 device / HIP code object missing: the product never falls back to a CPU path */
/* Errors.  No entry point throws, aborts or exits the host process; a failure is a negative code plus ga_last_error().
 * A render call (ga_render*, ga_process_blocks*) that fails while it validates its arguments or the graph (disposed context,
 * GA_ERR_CYCLE, GA_ERR_UNSUPPORTED, argument checks) leaves the context exactly as it was.  A render that fails later --
 * device out of memory, a rejected launch, a HIP error -- has already advanced control state it cannot take back
 * (AudioContextBase.ProcessBlock has no rollback either): the context is then FAULTED and every further render on it returns
 * GA_ERR_INVALID_OPERATION with the original message.  Destroy it and create a new one. */

/* ---- enums (numeric values follow declaration order in the reference) ---- */
enum { /* node types creatable through ga_node_create; the destination is always node id 0 */
  GA_NODE_DESTINATION = 0,   /* Nodes/AudioDestinationNode.cs:9 */
  GA_NODE_BUFFER_SOURCE = 1, /* Nodes/AudioBufferSourceNode.cs:13 ; params: 0 = playbackRate (k-rate) */
  GA_NODE_GAIN = 2,          /* Nodes/GainNode.cs:9            ; params: 0 = gain (a-rate) */
  GA_NODE_BIQUAD = 3,        /* Nodes/BiQuadFilterNode.cs:10   ; params: 0 = frequency (a), 1 = Q (a), 2 = gain dB (k) */
  GA_NODE_CONVOLVER = 4,     /* Nodes/ConvolverNode.cs:10      ; no params */
  /* the remaining pure-Core nodes */
  GA_NODE_CHANNEL_SPLITTER = 5, /* Nodes/ChannelSplitterNode.cs:9  ; N mono outputs (create_ex arg = numberOfOutputs) */
  GA_NODE_CHANNEL_MERGER = 6,   /* Nodes/ChannelMergerNode.cs:9    ; N inputs (create_ex arg = numberOfInputs) */
  GA_NODE_CONSTANT_SOURCE = 7,  /* Nodes/ConstantSourceNode.cs:15  ; params: 0 = offset (a-rate) ; scheduled source */
  GA_NODE_STEREO_PANNER = 8,    /* Nodes/StereoPannerNode.cs:9     ; params: 0 = pan (a-rate) */
  GA_NODE_OSCILLATOR = 9,       /* Nodes/OscillatorNode.cs:12      ; params: 0 = frequency (a-rate) ; scheduled source */
  GA_NODE_DELAY = 10,           /* Nodes/DelayNode.cs:9            ; params: 0 = delayTime (a-rate) ; create_ex arg = maxDelayTime */
  /* the second call site of the resampler */
  GA_NODE_STREAM_SOURCE = 11    /* GraphAudio.IO/AudioStreamSourceNodeBase.cs:19 (AudioStreamNodeBase) ; params: 0 = playbackRate (k-rate) */
};
enum { GA_STREAM_PLAYING = 0, GA_STREAM_PAUSED = 1, GA_STREAM_STOPPED = 2 };   /* StreamState, AudioStreamSourceNodeBase.cs:12-17 */
enum { GA_FILTER_LOWPASS = 0, GA_FILTER_HIGHPASS, GA_FILTER_BANDPASS, GA_FILTER_NOTCH, GA_FILTER_ALLPASS,
       GA_FILTER_PEAKING, GA_FILTER_LOWSHELF, GA_FILTER_HIGHSHELF }; /* BiQuadFilterNode.cs:288-298 */
enum { GA_OSC_SINE = 0, GA_OSC_SQUARE, GA_OSC_SAWTOOTH, GA_OSC_TRIANGLE };   /* OscillatorNode.cs:207-213 */
enum { GA_COUNT_MODE_MAX = 0, GA_COUNT_MODE_CLAMPED_MAX = 1, GA_COUNT_MODE_EXPLICIT = 2 }; /* AudioNodeInput.cs:258-272 */
enum { GA_INTERP_SPEAKERS = 0, GA_INTERP_DISCRETE = 1 };                                    /* AudioNodeInput.cs:246-256 */

typedef struct ga_context ga_context;

/* Counters filled by ga_get_stats (reference has only BufferPool.GetStatistics, BufferPool.cs:133-149). */
typedef struct ga_stats {
  int64_t blocks_rendered;      /* AudioContextBase.CurrentBlock (AudioContextBase.cs:223) */
  int64_t chunks;               /* device render chunks executed */
  int64_t segments;             /* control-state segments executed */
  int64_t kernel_launches;
  double  device_ms_total;      /* sum of per-chunk device time, HIP events on the context's stream */
  /* dominant-kernel timing (the spectral multiply-accumulate of PartitionedConvolver.cs:154-223) */
  int64_t mac_launches;
  double  mac_ms_total;
  double  mac_flops_total;      /* algorithmic flops: 8 * P * 129 per channel-instance per block */
  double  mac_bytes_total;      /* algorithmic bytes per SURVEY.md section 8(d) streaming formulation */
  double  fft_ms_total;         /* forward rfft256 + inverse/overlap-add kernels */
  double  other_ms_total;       /* source / biquad / gain / mix / param-curve kernels */
  int64_t device_bytes_in_use;
  int32_t n_nodes;
  int32_t n_conv_rows;          /* convolver channel-instances resident on the device */
  /* per-stage device time (HIP events on the context's stream, option "profile"), launches and the HBM bytes each stage HAS TO
     move in the formulation that was executed (inputs read once + outputs written once, from the plan) -- what bench.py prices
     against the HBM roofline.  Index = GA_STAGE_*. */
  double  stage_ms[16];
  int64_t stage_launches[16];
  double  stage_bytes[16];
  int64_t profiled_chunks;      /* chunks whose HIP events device_ms_total and stage_ms[] were read from (options "profile",
                                   "profile_every"); stage_launches[] and stage_bytes[] count every chunk */
  int64_t coarse_carried_outputs; /* formulation D: (output channel, chunk) pairs rendered from a carried tail instead of the
                                     members' input histories */
  double  stage_flops[16];      /* floating-point operations the stage's launches execute in the formulation that ran (counted by the
                                   planner from the job tables: complex multiply-adds of the partition sums, butterflies of the
                                   transforms); 0 where not accounted */
  char    stage_kernel[16][64]; /* name of the kernel (template instance) the stage's most recent launch ran, "" if none */
  int64_t coarse_premixed_signals; /* formulation D: (member input channel, chunk) pairs that were summed in the time domain in
                                      front of their group's transforms instead of being transformed one by one */
  int64_t deferred_handovers;   /* asynchronous renders whose bus crossed PCIe inside the next chunk's pre-mix launch (option
                                   "host_defer") instead of at the end of their own last kernel */
  int64_t biquad_split_cascades; /* (cascade x channel, segment) pairs evaluated in pieces along time (option "biquad_time_split") */
  int64_t ref_order_rows;       /* (convolver channel-instance, chunk) pairs whose partition sum was evaluated in the reference's own
                                   order and float32 arithmetic between double-precision transforms (formulation R, option
                                   "conv_reference_order"): PartitionedConvolver.cs:104-223 bit for bit */
  int64_t sim_replays;          /* chunks whose first block was not traversed again: the control-plane records of the previous chunk's
                                   last segment were taken over (steady renders, option "sim_replay") */
  int64_t twin_rows;            /* channel rows that were not computed a second time: a mono signal in a stereo node or input is the
                                   same numbers on every channel (AudioNodeInput.cs:182-244), evaluated once (option "twin_channels") */
} ga_stats;
enum {
  GA_STAGE_OTHER = 0,        /* sources, biquads, gains, parameter curves, ... */
  GA_STAGE_MIX = 1,          /* AudioNodeInput.MixBuffer sums (incl. the destination bus) */
  GA_STAGE_RFFT_FWD = 2,     /* formulations A/B/C: 256-point forward transforms (+ history copy) */
  GA_STAGE_MAC = 3,          /* formulations A/B/C: partition sum */
  GA_STAGE_RFFT_INV = 4,     /* formulations A/B/C: 256-point inverse transforms + overlap-add */
  GA_STAGE_COARSE_FWD = 5,   /* formulation D: 16,384-point forward transforms */
  GA_STAGE_COARSE_MAC = 6,   /* formulation D: partition sum + frequency-domain mix */
  GA_STAGE_COARSE_INV = 7,   /* formulation D: inverse transforms */
  GA_STAGE_COARSE_HIST = 8,  /* formulation D: input history of the next chunk */
  GA_STAGE_COARSE_SECTION = 9, /* formulation D: wall time of the forward || multiply-accumulate section (the two stages overlap
                                  on two streams, so their own times add up to more than this) */
  GA_STAGE_COARSE_PREMIX = 10, /* formulation D: time-domain sum of a fused group that shares one impulse response (+ the members'
                                  input histories of the next chunk) */
  GA_STAGE_COUNT = 11
};

/* ---- library ---- */
GA_EXPORT const char* GA_FN(strerror)(int code);              /* cf. sf_strerror, GraphAudio.IO/Libsndfile.cs:48-56 */
GA_EXPORT const char* GA_FN(version)(void);
GA_EXPORT int GA_FN(device_count)(void);                       /* number of visible gfx950 devices; 0 if none */

/* ---- context: OfflineAudioContext(int sampleRate = 48000)  (OfflineAudioContext.cs:18, AudioContextBase.cs:35-47) ---- */
GA_EXPORT int GA_FN(context_create)(int sample_rate, int device_ordinal, ga_context** out);
GA_EXPORT int GA_FN(context_destroy)(ga_context* ctx);         /* AudioContextBase.Dispose, AudioContextBase.cs:243-260 */
GA_EXPORT const char* GA_FN(last_error)(ga_context* ctx);      /* message of the last failing call on this context */
GA_EXPORT double GA_FN(current_time)(ga_context* ctx);         /* AudioContextBase.CurrentTime, AudioContextBase.cs:28 */
GA_EXPORT int64_t GA_FN(current_block)(ga_context* ctx);       /* AudioContextBase.CurrentBlock, AudioContextBase.cs:223 */
GA_EXPORT int GA_FN(set_option)(ga_context* ctx, const char* key, double value); /* tuning knobs, see DESIGN.md */
GA_EXPORT int GA_FN(get_stats)(ga_context* ctx, ga_stats* out);

/* ---- PlayableAudioBuffer.FromChannelArrays (PlayableAudioBuffer.cs:122-145): immutable sample storage ---- */
GA_EXPORT int GA_FN(buffer_create)(ga_context* ctx, const float* const* planar, int channels, int64_t frames,
                                   int sample_rate, int* out_buffer_id);
GA_EXPORT int GA_FN(buffer_release)(ga_context* ctx, int buffer_id);

/* ---- nodes: constructors of GainNode / BiQuadFilterNode / ConvolverNode / AudioBufferSourceNode ---- */
GA_EXPORT int GA_FN(node_create)(ga_context* ctx, int node_type, int* out_node_id);
/* constructors with an argument: ChannelSplitterNode(numberOfOutputs) ChannelSplitterNode.cs:14, ChannelMergerNode(numberOfInputs)
   ChannelMergerNode.cs:14, DelayNode(maxDelayTime seconds) DelayNode.cs:22 ; other node types ignore `arg` */
GA_EXPORT int GA_FN(node_create_ex)(ga_context* ctx, int node_type, double arg, int* out_node_id);
GA_EXPORT int GA_FN(node_dispose)(ga_context* ctx, int node);  /* AudioNode.Dispose, Nodes/AudioNode.cs:207-238 */
/* AudioNode.Connect(destination, outputIndex, inputIndex), Nodes/AudioNode.cs:68-73,109-123 */
GA_EXPORT int GA_FN(node_connect)(ga_context* ctx, int src, int dst, int output_index, int input_index);
/* AudioNode.Disconnect(destination?, outputIndex, inputIndex), Nodes/AudioNode.cs:78-81,129-150 ; dst < 0 = all */
GA_EXPORT int GA_FN(node_disconnect)(ga_context* ctx, int src, int dst, int output_index, int input_index);
/* AudioNode.Connect(AudioParam, outputIndex) / Disconnect(AudioParam, ...), Nodes/AudioNode.cs:86-103 */
GA_EXPORT int GA_FN(node_connect_param)(ga_context* ctx, int src, int dst_node, int dst_param, int output_index);
GA_EXPORT int GA_FN(node_disconnect_param)(ga_context* ctx, int src, int dst_node, int dst_param, int output_index);
/* 1 once a scheduled source has raised Ended (AudioBufferSourceNode.cs:378-389); the host raises the event */
GA_EXPORT int GA_FN(node_has_ended)(ga_context* ctx, int node);
/* Bulk form for hosts with thousands of sources: writes the ids of sources whose Ended was raised since the previous
 * call (at most `capacity`) and returns how many were written; call again while it returns `capacity`. */
GA_EXPORT int GA_FN(poll_ended)(ga_context* ctx, int* out_node_ids, int capacity);

/* AudioNodeInput.SetChannelCount / SetChannelCountMode / SetChannelInterpretation (AudioNodeInput.cs:41-58) */
GA_EXPORT int GA_FN(input_set_channel_count)(ga_context* ctx, int node, int input_index, int count);
GA_EXPORT int GA_FN(input_set_channel_count_mode)(ga_context* ctx, int node, int input_index, int mode);
GA_EXPORT int GA_FN(input_set_channel_interpretation)(ga_context* ctx, int node, int input_index, int interpretation);
/* AudioDestinationNode.SetChannelCount (Nodes/AudioDestinationNode.cs:23-32) */
GA_EXPORT int GA_FN(destination_set_channel_count)(ga_context* ctx, int channels);
/* channel count of the destination's last output buffer, or 2 before the first block: what Render(int) sizes by
 * (OfflineAudioContext.cs:108-124) */
GA_EXPORT int GA_FN(destination_output_channels)(ga_context* ctx);

/* ---- AudioParam (AudioParam.cs:34-49, 252-331) ---- */
GA_EXPORT int GA_FN(param_set_value)(ga_context* ctx, int node, int param, float value);
GA_EXPORT int GA_FN(param_get_value)(ga_context* ctx, int node, int param, float* out);
GA_EXPORT int GA_FN(param_set_value_at_time)(ga_context* ctx, int node, int param, float value, double start_time);
GA_EXPORT int GA_FN(param_linear_ramp_to_value_at_time)(ga_context* ctx, int node, int param, float value, double end_time);
GA_EXPORT int GA_FN(param_exponential_ramp_to_value_at_time)(ga_context* ctx, int node, int param, float value, double end_time);
GA_EXPORT int GA_FN(param_set_target_at_time)(ga_context* ctx, int node, int param, float target, double start_time,
                                              double time_constant);
GA_EXPORT int GA_FN(param_cancel_scheduled_values)(ga_context* ctx, int node, int param, double cancel_time);

/* ---- AudioBufferSourceNode (Nodes/AudioBufferSourceNode.cs:39-129) ---- */
GA_EXPORT int GA_FN(source_set_buffer)(ga_context* ctx, int node, int buffer_id); /* buffer_id < 0 = null */
GA_EXPORT int GA_FN(source_set_loop)(ga_context* ctx, int node, int loop, double loop_start, double loop_end);
/* Start / Stop of every IAudioScheduledSourceNode: AudioBufferSourceNode (duration default +inf, :79-129), ConstantSourceNode
   (ConstantSourceNode.cs:44-74) and OscillatorNode (OscillatorNode.cs:54-88) -- for those two pass NaN for "no duration" */
GA_EXPORT int GA_FN(source_start)(ga_context* ctx, int node, double when, double offset, double duration);
GA_EXPORT int GA_FN(source_stop)(ga_context* ctx, int node, double when);

/* ---- OscillatorNode.Type (Nodes/OscillatorNode.cs:33-42) ---- */
GA_EXPORT int GA_FN(oscillator_set_type)(ga_context* ctx, int node, int oscillator_type);

/* ---- BiQuadFilterNode.Type (Nodes/BiQuadFilterNode.cs:21-37) ---- */
GA_EXPORT int GA_FN(biquad_set_type)(ga_context* ctx, int node, int filter_type);

/* ---- ConvolverNode.Normalize / EnableTrueStereo / Buffer (Nodes/ConvolverNode.cs:25-95) ---- */
GA_EXPORT int GA_FN(convolver_set_normalize)(ga_context* ctx, int node, int normalize);
GA_EXPORT int GA_FN(convolver_set_enable_true_stereo)(ga_context* ctx, int node, int enable);
GA_EXPORT int GA_FN(convolver_set_buffer)(ga_context* ctx, int node, int buffer_id); /* buffer_id < 0 = null */

/* ---- AudioStreamNodeBase (GraphAudio.IO/AudioStreamSourceNodeBase.cs:19-329): a queue of PlayableAudioBuffers played back to back
 * through the CubicResampler.  The reference's concrete subclass (AudioDecoderStreamNode) fills the queue from a decoder
 * thread, which makes its output depend on thread timing; here the HOST queues buffers explicitly (between renders), so the
 * render is deterministic: same queue, same output as the reference's Process (:132-301).
 *   ga_stream_queue_buffer       QueueBuffer (:118-124)
 *   ga_stream_set_state          Play / Pause / Stop (:70-93); takes effect at once like the reference's Interlocked state; Stop
 *                                flushes the current and the queued buffers to the processed list and clears the resamplers (:95-116)
 *   ga_stream_dequeue_processed  TryDequeueProcessedBuffer (:126-129): 1 and the buffer id, or 0
 *   ga_stream_queued_count / ga_stream_processed_count   QueuedBufferCount / ProcessedBufferCount (:52-57) */
GA_EXPORT int GA_FN(stream_queue_buffer)(ga_context* ctx, int node, int buffer_id);
GA_EXPORT int GA_FN(stream_set_state)(ga_context* ctx, int node, int state);
GA_EXPORT int GA_FN(stream_dequeue_processed)(ga_context* ctx, int node, int* buffer_id_out);
GA_EXPORT int GA_FN(stream_queued_count)(ga_context* ctx, int node);
GA_EXPORT int GA_FN(stream_processed_count)(ga_context* ctx, int node);

/* ---- OfflineAudioContext.Render(float[][] output, int frameCount, int startIndex = 0) (OfflineAudioContext.cs:30-102)
 * out_planar[ch] points at a host array of at least start_index + frame_count floats.  DSP state, the block clock
 * and the leftover-frame cache persist across calls exactly as in the reference. */
GA_EXPORT int GA_FN(render)(ga_context* ctx, float* const* out_planar, int out_channels, int64_t frame_count,
                            int64_t start_index);
/* Same, but out_planar[ch] are DEVICE pointers on the context's GPU (used so the destination bus of a sharded render
 * can be summed across GPUs with RCCL without a host round trip).  The oracle build returns GA_ERR_UNSUPPORTED. */
GA_EXPORT int GA_FN(render_device)(ga_context* ctx, float* const* out_planar_dev, int out_channels, int64_t frame_count,
                                   int64_t start_index);
/* AudioContextBase.ProcessBlocks(float[][] outputBuffers, int blockCount), AudioContextBase.cs:163-186: block_count whole
   blocks; for every block the first min(out_channels, channels of the destination buffer) channels are written at
   block * 128, further channels are left untouched, null channel pointers are skipped.  (Frames cached by a previous
   partial Render are not consumed, as in the reference.)  out_on_device != 0: the pointers are device memory. */
GA_EXPORT int GA_FN(process_blocks)(ga_context* ctx, float* const* out_planar, int out_channels, int64_t block_count,
                                    int out_on_device);
/* AudioContextBase.ProcessBlockInterleaved(float[] interleavedBuffer, int channels), AudioContextBase.cs:88-157, for
   block_count consecutive blocks (the reference call is block_count = 1): frame-major interleaved samples,
   interleaved[(block * 128 + frame) * channels + ch]; channels beyond the destination buffer's are zero-filled.  The
   interleave runs on the device; this is the format RealtimeAudioContext.RenderLoop hands to the audio device
   (GraphAudio.Realtime/RealtimeAudioContext.cs:143-165). */
GA_EXPORT int GA_FN(process_blocks_interleaved)(ga_context* ctx, float* interleaved, int channels, int64_t block_count,
                                                int out_on_device);

/* Run the render on this HIP stream (a hipStream_t passed as void*) instead of the context's own stream.  On a caller-supplied
   stream every render call enqueues ALL of its work, the copy into the caller's rows included, before it returns: the caller may
   order later work behind it on the same stream, or wait on the stream / an event of their own, and finds the rows complete. */
GA_EXPORT int GA_FN(context_set_stream)(ga_context* ctx, void* hip_stream);
/* Pipelined rendering (no counterpart in the reference, whose Render is synchronous, OfflineAudioContext.cs:30-102): with
   ga_set_option(ctx, "async", 1) ga_render / ga_render_device return as soon as the work of the call is enqueued on the
   context's stream, so the host-side graph simulation of the next call overlaps the device execution of this one (the host
   runs at most one chunk ahead).  When are the output arrays (page-locked host memory or device memory) complete?
     * on a caller-supplied stream (ga_context_set_stream): after any later work on that stream, a wait on it, or ga_synchronize;
     * on the context's own stream: after ga_synchronize ONLY.  With option "host_defer" (default 1) the bus of a render into
       page-locked rows stays in device staging rows when the call returns and crosses PCIe inside the NEXT render call's first
       long launch (or from ga_synchronize) -- the rows of call k are complete once call k + 1's stream work has finished, or
       after ga_synchronize, not before.  Device-memory outputs (ga_render_device) are written by the call's own work.
   The default is synchronous, as in the reference. */
GA_EXPORT int GA_FN(synchronize)(ga_context* ctx);


/* ---- sharded render: voices split over the GPUs of one node, ONE sum of the destination bus per render call ----------------
 * The reference mixes every voice at the destination on one thread (AudioNodeInput.MixBuffer, AudioNodeInput.cs:118-132,
 * 195-198).  Voice chains are independent up to that sum, so a host may build the same graph on N
 * contexts -- one per GPU, each with a contiguous share of the voices (ga_shard_range) -- and let the library add the buses:
 * ga_render_reduce renders this rank's share into device memory and sums the destination bus of all ranks with ONE RCCL
 * reduce over xGMI, ordered on the context's stream (render -> reduce -> copy to the caller's arrays on the root rank).
 * The ranks may be threads of one process or separate processes (one per GPU); the host only has to hand the 128-byte id
 * that rank 0 obtains from ga_comm_unique_id to every rank (any channel: shared memory, a file, a socket).
 * RCCL (librccl.so.1) is loaded on the first ga_comm_* call; single-GPU hosts never need it.  With n_ranks = 1 the calls work
 * without RCCL and ga_render_reduce equals ga_render.  The float32 sum across ranks is associated differently from the
 * reference's sequential order: rounding only. */
#define GA_COMM_ID_BYTES 128
GA_EXPORT int GA_FN(comm_unique_id)(void* id_out);   /* GA_COMM_ID_BYTES bytes; rank 0 calls it, every rank gets a copy */
GA_EXPORT int GA_FN(comm_init)(ga_context* ctx, const void* id, int n_ranks, int rank);   /* collective over the ranks */
GA_EXPORT int GA_FN(comm_destroy)(ga_context* ctx);
/* What the communicator itself says about this context: the number of ranks RCCL sees (ncclCommCount) and this context's rank in it
   (ncclCommUserRank) -- not what ga_comm_init was told.  A host (or a scaling benchmark) compares them with its own world size, so a
   run in which the ranks did not join ONE communicator cannot pass as an N-GPU measurement.  Without a communicator (ga_comm_init was
   never called, or n_ranks == 1: no RCCL involved) both come back as 1 / 0 and *uses_rccl = 0. */
GA_EXPORT int GA_FN(comm_info)(ga_context* ctx, int* n_ranks, int* rank, int* uses_rccl);
/* voices [first, first + count) of n_voices for `rank` of n_ranks (contiguous, sizes differ by at most one) */
GA_EXPORT int GA_FN(shard_range)(int64_t n_voices, int n_ranks, int rank, int64_t* first, int64_t* count);
/* Render(output, frameCount, startIndex) of the sharded graph: collective; out_planar (host arrays, page-locked for
   "async" contexts) is written on rank `root` only and may be null elsewhere.  out_channels must be the same on every rank. */
GA_EXPORT int GA_FN(render_reduce)(ga_context* ctx, float* const* out_planar, int out_channels, int64_t frame_count,
                                   int64_t start_index, int root);

#ifdef __cplusplus
}
#endif
#endif /* GRAPHAUDIO_HIP_H */
