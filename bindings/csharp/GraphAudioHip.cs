// GraphAudioHip.cs -- P/Invoke layer over libgraphaudio_hip.so (include/graphaudio_hip.h).
// Same conventions as the reference's own native bindings (GraphAudio.IO/Libsndfile.cs:6,36-68,
// GraphAudio.Realtime/Miniaudio.cs:303-349): [LibraryImport] + CallConvCdecl on static partial methods of an
// internal static unsafe partial class, opaque IntPtr handle, int result codes (0 = ok, negative = error).
// NOTE: there is no .NET toolchain in the build image of this repository; this file is compiled by the GraphAudio
// maintainer as part of GraphAudio.Core (see INTEGRATION.md).  Keep it mechanical: one method per GA_FN() declaration.
using System;
using System.Runtime.CompilerServices;
using System.Runtime.InteropServices;

namespace GraphAudio.Core.Hip;

internal static unsafe partial class GraphAudioHip
{
    private const string Lib = "graphaudio_hip";   // runtimes/linux-x64/native/libgraphaudio_hip.so

    public const int GA_OK = 0, GA_ERR_INVALID_ARGUMENT = -1, GA_ERR_OUT_OF_RANGE = -2, GA_ERR_INVALID_OPERATION = -3,
                     GA_ERR_DISPOSED = -4, GA_ERR_CYCLE = -5, GA_ERR_UNSUPPORTED = -6, GA_ERR_DEVICE = -7,
                     GA_ERR_OUT_OF_MEMORY = -8, GA_ERR_NO_DEVICE = -9;
    public const int NodeBufferSource = 1, NodeGain = 2, NodeBiquad = 3, NodeConvolver = 4;
    public const int NodeChannelSplitter = 5, NodeChannelMerger = 6, NodeConstantSource = 7, NodeStereoPanner = 8,
                     NodeOscillator = 9, NodeDelay = 10, NodeStreamSource = 11;

    // ga_stats (include/graphaudio_hip.h): the layout has to match field for field -- ga_get_stats writes the whole struct
    [StructLayout(LayoutKind.Sequential)]
    public unsafe struct Stats
    {
        public long blocks_rendered, chunks, segments, kernel_launches;
        public double device_ms_total;
        public long mac_launches;
        public double mac_ms_total, mac_flops_total, mac_bytes_total, fft_ms_total, other_ms_total;
        public long device_bytes_in_use;
        public int n_nodes, n_conv_rows;
        public fixed double stage_ms[16];        // index = GA_STAGE_*
        public fixed long stage_launches[16];
        public fixed double stage_bytes[16];
        public long profiled_chunks;
        public long coarse_carried_outputs;
        public fixed double stage_flops[16];
        public fixed byte stage_kernel[1024];    // [16][64] zero-terminated names
        public long coarse_premixed_signals;
        public long deferred_handovers;
        public long biquad_split_cascades;
        public long ref_order_rows;
        public long sim_replays;
        public long twin_rows;
    }

    [LibraryImport(Lib, EntryPoint = "ga_strerror")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    private static partial IntPtr ga_strerror(int code);
    [LibraryImport(Lib, EntryPoint = "ga_last_error")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    private static partial IntPtr ga_last_error(IntPtr ctx);
    [LibraryImport(Lib, EntryPoint = "ga_device_count")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_device_count();

    [LibraryImport(Lib, EntryPoint = "ga_context_create")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_context_create(int sampleRate, int deviceOrdinal, out IntPtr ctx);
    [LibraryImport(Lib, EntryPoint = "ga_context_destroy")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_context_destroy(IntPtr ctx);
    [LibraryImport(Lib, EntryPoint = "ga_current_time")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial double ga_current_time(IntPtr ctx);
    [LibraryImport(Lib, EntryPoint = "ga_current_block")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial long ga_current_block(IntPtr ctx);
    [LibraryImport(Lib, EntryPoint = "ga_get_stats")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_get_stats(IntPtr ctx, out Stats stats);

    [LibraryImport(Lib, EntryPoint = "ga_buffer_create")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_buffer_create(IntPtr ctx, float** planar, int channels, long frames, int sampleRate, out int bufferId);
    [LibraryImport(Lib, EntryPoint = "ga_buffer_release")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_buffer_release(IntPtr ctx, int bufferId);

    [LibraryImport(Lib, EntryPoint = "ga_node_create")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_node_create(IntPtr ctx, int nodeType, out int nodeId);
    [LibraryImport(Lib, EntryPoint = "ga_node_create_ex")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_node_create_ex(IntPtr ctx, int nodeType, double ctorArg, out int nodeId);
    [LibraryImport(Lib, EntryPoint = "ga_oscillator_set_type")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_oscillator_set_type(IntPtr ctx, int node, int oscillatorType);
    [LibraryImport(Lib, EntryPoint = "ga_node_dispose")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_node_dispose(IntPtr ctx, int node);
    [LibraryImport(Lib, EntryPoint = "ga_node_connect")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_node_connect(IntPtr ctx, int src, int dst, int outputIndex, int inputIndex);
    [LibraryImport(Lib, EntryPoint = "ga_node_disconnect")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_node_disconnect(IntPtr ctx, int src, int dst, int outputIndex, int inputIndex);
    [LibraryImport(Lib, EntryPoint = "ga_node_connect_param")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_node_connect_param(IntPtr ctx, int src, int dstNode, int dstParam, int outputIndex);
    [LibraryImport(Lib, EntryPoint = "ga_node_has_ended")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_node_has_ended(IntPtr ctx, int node);

    [LibraryImport(Lib, EntryPoint = "ga_process_blocks")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_process_blocks(IntPtr ctx, float** outPlanar, int outChannels, long blockCount, int outOnDevice);
    [LibraryImport(Lib, EntryPoint = "ga_process_blocks_interleaved")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_process_blocks_interleaved(IntPtr ctx, float* interleaved, int channels, long blockCount, int outOnDevice);

    [LibraryImport(Lib, EntryPoint = "ga_poll_ended")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_poll_ended(IntPtr ctx, int* outNodeIds, int capacity);

    [LibraryImport(Lib, EntryPoint = "ga_input_set_channel_count")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_input_set_channel_count(IntPtr ctx, int node, int input, int count);
    [LibraryImport(Lib, EntryPoint = "ga_input_set_channel_count_mode")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_input_set_channel_count_mode(IntPtr ctx, int node, int input, int mode);
    [LibraryImport(Lib, EntryPoint = "ga_input_set_channel_interpretation")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_input_set_channel_interpretation(IntPtr ctx, int node, int input, int interpretation);
    [LibraryImport(Lib, EntryPoint = "ga_destination_set_channel_count")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_destination_set_channel_count(IntPtr ctx, int channels);
    [LibraryImport(Lib, EntryPoint = "ga_destination_output_channels")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_destination_output_channels(IntPtr ctx);

    [LibraryImport(Lib, EntryPoint = "ga_param_set_value")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_param_set_value(IntPtr ctx, int node, int param, float value);
    [LibraryImport(Lib, EntryPoint = "ga_param_set_value_at_time")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_param_set_value_at_time(IntPtr ctx, int node, int param, float value, double startTime);
    [LibraryImport(Lib, EntryPoint = "ga_param_linear_ramp_to_value_at_time")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_param_linear_ramp_to_value_at_time(IntPtr ctx, int node, int param, float value, double endTime);
    [LibraryImport(Lib, EntryPoint = "ga_param_exponential_ramp_to_value_at_time")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_param_exponential_ramp_to_value_at_time(IntPtr ctx, int node, int param, float value, double endTime);
    [LibraryImport(Lib, EntryPoint = "ga_param_set_target_at_time")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_param_set_target_at_time(IntPtr ctx, int node, int param, float target, double startTime, double timeConstant);
    [LibraryImport(Lib, EntryPoint = "ga_param_cancel_scheduled_values")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_param_cancel_scheduled_values(IntPtr ctx, int node, int param, double cancelTime);

    [LibraryImport(Lib, EntryPoint = "ga_source_set_buffer")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_source_set_buffer(IntPtr ctx, int node, int bufferId);
    [LibraryImport(Lib, EntryPoint = "ga_source_set_loop")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_source_set_loop(IntPtr ctx, int node, int loop, double loopStart, double loopEnd);
    [LibraryImport(Lib, EntryPoint = "ga_source_start")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_source_start(IntPtr ctx, int node, double when, double offset, double duration);
    [LibraryImport(Lib, EntryPoint = "ga_source_stop")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_source_stop(IntPtr ctx, int node, double when);
    [LibraryImport(Lib, EntryPoint = "ga_biquad_set_type")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_biquad_set_type(IntPtr ctx, int node, int filterType);
    [LibraryImport(Lib, EntryPoint = "ga_convolver_set_normalize")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_convolver_set_normalize(IntPtr ctx, int node, int normalize);
    [LibraryImport(Lib, EntryPoint = "ga_convolver_set_enable_true_stereo")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_convolver_set_enable_true_stereo(IntPtr ctx, int node, int enable);
    [LibraryImport(Lib, EntryPoint = "ga_convolver_set_buffer")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_convolver_set_buffer(IntPtr ctx, int node, int bufferId);

    [LibraryImport(Lib, EntryPoint = "ga_render")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_render(IntPtr ctx, float** outPlanar, int outChannels, long frameCount, long startIndex);

    [LibraryImport(Lib, EntryPoint = "ga_synchronize")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_synchronize(IntPtr ctx);   // with option "async": waits for the enqueued renders
    [LibraryImport(Lib, EntryPoint = "ga_render_device")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static unsafe partial int ga_render_device(IntPtr ctx, float** outPlanarDev, int outChannels, long frameCount, long startIndex);   // rows in device memory
    [LibraryImport(Lib, EntryPoint = "ga_context_set_stream")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_context_set_stream(IntPtr ctx, IntPtr hipStream);
    [LibraryImport(Lib, EntryPoint = "ga_set_option", StringMarshalling = StringMarshalling.Utf8)] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_set_option(IntPtr ctx, string key, double value);   // "async", "max_chunk_blocks", ... (DESIGN.md)
    [LibraryImport(Lib, EntryPoint = "ga_version")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    private static partial IntPtr ga_version();
    public static string Version => Marshal.PtrToStringUTF8(ga_version()) ?? "";
    [LibraryImport(Lib, EntryPoint = "ga_node_disconnect_param")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_node_disconnect_param(IntPtr ctx, int src, int dstNode, int dstParam, int outputIndex);   // AudioNode.Disconnect(AudioParam, int)
    [LibraryImport(Lib, EntryPoint = "ga_param_get_value")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_param_get_value(IntPtr ctx, int node, int param, out float value);                        // AudioParam.Value getter

    // ---- AudioStreamNodeBase with an explicit queue (GraphAudio.IO/AudioStreamSourceNodeBase.cs; node type 11) ----
    [LibraryImport(Lib, EntryPoint = "ga_stream_queue_buffer")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_stream_queue_buffer(IntPtr ctx, int node, int bufferId);          // QueueBuffer
    [LibraryImport(Lib, EntryPoint = "ga_stream_set_state")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_stream_set_state(IntPtr ctx, int node, int state);                // Play / Pause / Stop = 0 / 1 / 2
    [LibraryImport(Lib, EntryPoint = "ga_stream_dequeue_processed")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_stream_dequeue_processed(IntPtr ctx, int node, out int bufferId); // TryDequeueProcessedBuffer: 1 / 0
    [LibraryImport(Lib, EntryPoint = "ga_stream_queued_count")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_stream_queued_count(IntPtr ctx, int node);
    [LibraryImport(Lib, EntryPoint = "ga_stream_processed_count")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_stream_processed_count(IntPtr ctx, int node);

    // ---- sharded render: one context per GPU (threads of this process or one process per GPU), voices split with
    //      ga_shard_range, ONE RCCL sum of the destination bus per Render inside the library (include/graphaudio_hip.h) ----
    public const int GA_COMM_ID_BYTES = 128;
    [LibraryImport(Lib, EntryPoint = "ga_comm_unique_id")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_comm_unique_id(byte* idOut);                 // rank 0; hand the 128 bytes to every rank
    [LibraryImport(Lib, EntryPoint = "ga_comm_init")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_comm_init(IntPtr ctx, byte* id, int nRanks, int rank);   // collective
    [LibraryImport(Lib, EntryPoint = "ga_comm_destroy")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_comm_destroy(IntPtr ctx);
    [LibraryImport(Lib, EntryPoint = "ga_comm_info")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_comm_info(IntPtr ctx, out int nRanks, out int rank, out int usesRccl);   // what RCCL itself reports
    [LibraryImport(Lib, EntryPoint = "ga_shard_range")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_shard_range(long nVoices, int nRanks, int rank, out long first, out long count);
    [LibraryImport(Lib, EntryPoint = "ga_render_reduce")] [UnmanagedCallConv(CallConvs = new[] { typeof(CallConvCdecl) })]
    public static partial int ga_render_reduce(IntPtr ctx, float** outPlanar, int outChannels, long frameCount, long startIndex, int root);

    /// <summary>Maps a negative result code to the exception the stock CPU context throws in the same situation.</summary>
    public static void Check(IntPtr ctx, int code)
    {
        if (code >= 0) return;
        string msg = Marshal.PtrToStringAnsi(ctx != IntPtr.Zero ? ga_last_error(ctx) : ga_strerror(code)) ?? "graphaudio_hip error";
        throw code switch
        {
            GA_ERR_INVALID_ARGUMENT => new ArgumentException(msg),
            GA_ERR_OUT_OF_RANGE => new ArgumentOutOfRangeException(null, msg),
            GA_ERR_DISPOSED => new ObjectDisposedException(msg),
            GA_ERR_UNSUPPORTED => new NotSupportedException(msg),   // caller falls back to OfflineAudioContext
            _ => new InvalidOperationException(msg),
        };
    }
}
