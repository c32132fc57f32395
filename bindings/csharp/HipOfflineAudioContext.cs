// HipOfflineAudioContext.cs -- drop-in replacement for OfflineAudioContext (OfflineAudioContext.cs:8-158) that renders
// the graph on an MI355X through libgraphaudio_hip.so.  User code is unchanged: nodes are created against this context,
// connected with AudioNode.Connect, scheduled with Start/Stop and AudioParam automation, then
//     ctx.Render(float[][] output, int frameCount, int startIndex = 0)      // same signature as OfflineAudioContext.cs:30
// Lives INSIDE the GraphAudio.Core assembly (AllowUnsafeBlocks is already on, GraphAudio.Core.csproj:8) because the
// snapshot below reads internal/private state: AudioNode.Params (Nodes/AudioNode.cs:26), AudioParam._events
// (AudioParam.cs:22), AudioBufferSourceNode._startTime/_stopTime/_offset/_duration (AudioBufferSourceNode.cs:19-22),
// AudioNodeInput mode/interpretation (AudioNodeInput.cs:20-21).  INTEGRATION.md lists the five one-line `internal`
// accessors that must be added next to those fields.
//
// Design: the managed side stays the single source of truth for the graph.  Before every Render it (1) drains the
// command queue exactly like AudioContextBase.ProcessBlock does (AudioContextBase.cs:57), (2) walks the graph from
// Destination (same traversal as GetAllNodes, AudioContextBase.cs:191-218), (3) replays what changed since the last
// call through the flat C ABI -- O(graph delta) calls -- and (4) makes ONE ga_render call for the whole frame range.
// Unsupported node types (anything but AudioBufferSourceNode, GainNode, BiQuadFilterNode, ConvolverNode, ChannelSplitterNode,
// ChannelMergerNode, ConstantSourceNode, StereoPannerNode, OscillatorNode, DelayNode and the
// destination) make EnsureSynced throw NotSupportedException; callers fall back to the stock CPU context.
// NOTE: not compiled in this repository's build image (no .NET); reviewed by reading.  The Python host in
// graphaudio_amd/core.py drives the very same C ABI calls in the same order and IS tested.
using System;
using System.Collections.Generic;
using GraphAudio.Nodes;

namespace GraphAudio.Core.Hip;

public sealed unsafe class HipOfflineAudioContext : AudioContextBase
{
    private IntPtr _native;
    private readonly Dictionary<AudioNode, int> _nodeIds = new();
    private readonly Dictionary<PlayableAudioBuffer, int> _bufferIds = new();
    private readonly Dictionary<AudioNode, NodeShadow> _shadow = new();

    public HipOfflineAudioContext(int sampleRate = 48000, int device = 0) : base(sampleRate)
    {
        GraphAudioHip.Check(IntPtr.Zero, GraphAudioHip.ga_context_create(sampleRate, device, out _native));
        _nodeIds[Destination] = 0;
    }

    /// <summary>The native context handle (ga_comm_init and other calls that have no managed wrapper).</summary>
    public IntPtr Handle => _native;

    /// <summary>Render of a voice-sharded graph (INTEGRATION.md section 4): this rank renders its share, the library sums the
    /// destination buses of all ranks with one RCCL reduce; <paramref name="output"/> is written on <paramref name="root"/> only.</summary>
    public void RenderReduce(float[][] output, int frameCount, int startIndex = 0, int root = 0) => RenderCore(output, frameCount, startIndex, root);

    /// <summary>Same contract as OfflineAudioContext.Render (OfflineAudioContext.cs:30-102).</summary>
    public void Render(float[][] output, int frameCount, int startIndex = 0) => RenderCore(output, frameCount, startIndex, -1);

    private void RenderCore(float[][] output, int frameCount, int startIndex, int reduceRoot)
    {
        if (output.Length == 0) throw new ArgumentException("Output buffer must have at least one channel.", nameof(output));
        if (frameCount <= 0) throw new ArgumentOutOfRangeException(nameof(frameCount), "Frame count must be positive.");
        if (startIndex < 0) throw new ArgumentOutOfRangeException(nameof(startIndex), "Start index must be non-negative.");
        for (int ch = 0; ch < output.Length; ch++)
        {
            if (output[ch] is null) throw new ArgumentException($"Channel {ch} buffer is null.", nameof(output));
            if (output[ch].Length < startIndex + frameCount) throw new ArgumentException($"Channel {ch} buffer is too small.", nameof(output));
        }
        EnsureSynced();
        // pin every channel for the duration of the call only (SteamAudioNodeBase.cs:86-95 does the same)
        var handles = new System.Runtime.InteropServices.GCHandle[output.Length];
        float** ptrs = stackalloc float*[output.Length];
        try
        {
            for (int ch = 0; ch < output.Length; ch++)
            {
                handles[ch] = System.Runtime.InteropServices.GCHandle.Alloc(output[ch], System.Runtime.InteropServices.GCHandleType.Pinned);
                ptrs[ch] = (float*)handles[ch].AddrOfPinnedObject();
            }
            GraphAudioHip.Check(_native, reduceRoot < 0
                ? GraphAudioHip.ga_render(_native, ptrs, output.Length, frameCount, startIndex)
                : GraphAudioHip.ga_render_reduce(_native, ptrs, output.Length, frameCount, startIndex, reduceRoot));
        }
        finally
        {
            foreach (var h in handles) if (h.IsAllocated) h.Free();
        }
        RaiseEndedEvents();
    }

    public float[][] Render(int frameCount)   // OfflineAudioContext.cs:108-124
    {
        if (frameCount <= 0) throw new ArgumentOutOfRangeException(nameof(frameCount), "Frame count must be positive.");
        int channels = GraphAudioHip.ga_destination_output_channels(_native);
        var output = new float[channels][];
        for (int ch = 0; ch < channels; ch++) output[ch] = new float[frameCount];
        Render(output, frameCount);
        return output;
    }

    // ---- graph snapshot -> C ABI delta ------------------------------------------------------------------------
    private sealed class NodeShadow
    {
        public List<(AudioNode src, int outIdx)>[] Inputs = Array.Empty<List<(AudioNode, int)>>();
        public int[] ParamVersions = Array.Empty<int>();
        public object? Buffer;
        public bool Started;
    }

    private void EnsureSynced()
    {
        DrainCommandsInternal();                       // internal accessor for DrainCommands (AudioContextBase.cs:272-284)
        foreach (var node in GetAllNodes())            // Destination first, then upstream (AudioContextBase.cs:191-218)
        {
            if (!_nodeIds.TryGetValue(node, out int id)) id = CreateNative(node);
            SyncSettings(node, id);
        }
        foreach (var node in GetAllNodes()) SyncConnections(node, _nodeIds[node]);
    }

    private int CreateNative(AudioNode node)
    {
        // (node type, constructor argument): the argument is the output / input count of a splitter / merger
        // (ChannelSplitterNode.cs:14, ChannelMergerNode.cs:14) or DelayNode's maxDelayTime (DelayNode.cs:22, internal accessor)
        (int type, double arg) = node switch
        {
            AudioBufferSourceNode => (GraphAudioHip.NodeBufferSource, 0.0),
            GainNode => (GraphAudioHip.NodeGain, 0.0),
            BiQuadFilterNode => (GraphAudioHip.NodeBiquad, 0.0),
            ConvolverNode => (GraphAudioHip.NodeConvolver, 0.0),
            ChannelSplitterNode sp => (GraphAudioHip.NodeChannelSplitter, (double)sp.Outputs.Count),
            ChannelMergerNode mg => (GraphAudioHip.NodeChannelMerger, (double)mg.Inputs.Count),
            ConstantSourceNode => (GraphAudioHip.NodeConstantSource, 0.0),
            StereoPannerNode => (GraphAudioHip.NodeStereoPanner, 0.0),
            OscillatorNode => (GraphAudioHip.NodeOscillator, 0.0),
            DelayNode d => (GraphAudioHip.NodeDelay, d.MaxDelayTimeInternal),
            _ => throw new NotSupportedException($"{node.GetType().Name} is not on the HIP render path; use OfflineAudioContext"),
        };
        GraphAudioHip.Check(_native, GraphAudioHip.ga_node_create_ex(_native, type, arg, out int id));
        _nodeIds[node] = id;
        _shadow[node] = new NodeShadow();
        return id;
    }

    private int BufferId(PlayableAudioBuffer b)
    {
        if (_bufferIds.TryGetValue(b, out int id)) return id;
        float** ch = stackalloc float*[b.NumberOfChannels];
        var pins = new System.Runtime.InteropServices.GCHandle[b.NumberOfChannels];
        try
        {
            for (int c = 0; c < b.NumberOfChannels; c++)
            {
                pins[c] = System.Runtime.InteropServices.GCHandle.Alloc(b.GetChannelArrayInternal(c), System.Runtime.InteropServices.GCHandleType.Pinned);
                ch[c] = (float*)pins[c].AddrOfPinnedObject();
            }
            GraphAudioHip.Check(_native, GraphAudioHip.ga_buffer_create(_native, ch, b.NumberOfChannels, b.Length, b.SampleRate, out id));
        }
        finally { foreach (var p in pins) if (p.IsAllocated) p.Free(); }
        _bufferIds[b] = id;
        return id;
    }

    private void SyncSettings(AudioNode node, int id)
    {
        var sh = _shadow.TryGetValue(node, out var s) ? s : (_shadow[node] = new NodeShadow());
        // input channel configuration (AudioNodeInput.cs:41-58): cheap, always replayed
        for (int i = 0; i < node.Inputs.Count; i++)
        {
            var inp = node.Inputs[i];
            GraphAudioHip.Check(_native, GraphAudioHip.ga_input_set_channel_count(_native, id, i, inp.ChannelCount));
            GraphAudioHip.Check(_native, GraphAudioHip.ga_input_set_channel_count_mode(_native, id, i, (int)inp.ChannelCountModeInternal));
        }
        // params: value + event list replayed when the copy-on-write array changed (AudioParam.cs:333-352)
        var ps = node.Params;
        if (sh.ParamVersions.Length != ps.Count) sh.ParamVersions = new int[ps.Count];
        for (int p = 0; p < ps.Count; p++)
        {
            var ev = ps[p].EventsInternal;              // AutomationEvent[] snapshot
            int version = HashCode.Combine(ev, ps[p].Value);
            if (version == sh.ParamVersions[p]) continue;
            sh.ParamVersions[p] = version;
            GraphAudioHip.Check(_native, GraphAudioHip.ga_param_set_value(_native, id, p, ps[p].Value));   // also clears native events
            foreach (var e in ev)
            {
                int rc = e.Type switch
                {
                    0 => GraphAudioHip.ga_param_set_value_at_time(_native, id, p, e.Value, e.Time),
                    1 => GraphAudioHip.ga_param_linear_ramp_to_value_at_time(_native, id, p, e.Value, e.Time),
                    2 => GraphAudioHip.ga_param_exponential_ramp_to_value_at_time(_native, id, p, e.Value, e.Time),
                    _ => GraphAudioHip.ga_param_set_target_at_time(_native, id, p, e.Target, e.Time, e.TimeConstant),
                };
                GraphAudioHip.Check(_native, rc);
            }
        }
        switch (node)
        {
            case BiQuadFilterNode bq:
                GraphAudioHip.Check(_native, GraphAudioHip.ga_biquad_set_type(_native, id, (int)bq.Type));
                break;
            case ConvolverNode cv when !ReferenceEquals(sh.Buffer, cv.Buffer):
                GraphAudioHip.Check(_native, GraphAudioHip.ga_convolver_set_normalize(_native, id, cv.Normalize ? 1 : 0));
                GraphAudioHip.Check(_native, GraphAudioHip.ga_convolver_set_enable_true_stereo(_native, id, cv.EnableTrueStereo ? 1 : 0));
                GraphAudioHip.Check(_native, GraphAudioHip.ga_convolver_set_buffer(_native, id, cv.Buffer is null ? -1 : BufferId(cv.Buffer)));
                sh.Buffer = cv.Buffer;
                break;
            case OscillatorNode osc:
                GraphAudioHip.Check(_native, GraphAudioHip.ga_oscillator_set_type(_native, id, (int)osc.Type));
                SyncScheduled(osc, id, sh);
                break;
            case ConstantSourceNode cs:
                SyncScheduled(cs, id, sh);
                break;
            case AudioBufferSourceNode src:
                if (!ReferenceEquals(sh.Buffer, src.Buffer))
                {
                    GraphAudioHip.Check(_native, GraphAudioHip.ga_source_set_buffer(_native, id, src.Buffer is null ? -1 : BufferId(src.Buffer)));
                    sh.Buffer = src.Buffer;
                }
                GraphAudioHip.Check(_native, GraphAudioHip.ga_source_set_loop(_native, id, src.Loop ? 1 : 0, src.LoopStart, src.LoopEnd));
                if (src.HasStartedInternal && !sh.Started)
                {
                    GraphAudioHip.Check(_native, GraphAudioHip.ga_source_start(_native, id, src.StartTimeInternal, src.OffsetInternal, src.DurationInternal));
                    sh.Started = true;
                }
                if (!double.IsNaN(src.StopTimeInternal) && double.IsPositiveInfinity(src.DurationInternal))
                    GraphAudioHip.Check(_native, GraphAudioHip.ga_source_stop(_native, id, src.StopTimeInternal));
                break;
        }
    }

    // Start / Stop of ConstantSourceNode and OscillatorNode (IAudioScheduledSourceNode): NaN duration = none
    private void SyncScheduled(IAudioScheduledSourceNode src, int id, NodeShadow sh)
    {
        if (src.HasStartedInternal && !sh.Started)
        {
            GraphAudioHip.Check(_native, GraphAudioHip.ga_source_start(_native, id, src.StartTimeInternal, 0.0, double.NaN));
            sh.Started = true;
        }
        if (!double.IsNaN(src.StopTimeInternal))
            GraphAudioHip.Check(_native, GraphAudioHip.ga_source_stop(_native, id, src.StopTimeInternal));
    }

    private void SyncConnections(AudioNode node, int id)
    {
        var sh = _shadow.TryGetValue(node, out var s) ? s : (_shadow[node] = new NodeShadow());
        if (sh.Inputs.Length != node.Inputs.Count)
        {
            sh.Inputs = new List<(AudioNode, int)>[node.Inputs.Count];
            for (int i = 0; i < sh.Inputs.Length; i++) sh.Inputs[i] = new();
        }
        for (int i = 0; i < node.Inputs.Count; i++)
        {
            var now = new List<(AudioNode, int)>();
            foreach (var o in node.Inputs[i].ConnectedOutputs) now.Add((o.Owner, o.Index));
            var before = sh.Inputs[i];
            bool same = before.Count == now.Count;
            for (int k = 0; same && k < now.Count; k++) same = ReferenceEquals(before[k].src, now[k].Item1) && before[k].outIdx == now[k].Item2;
            if (same) continue;
            // connection ORDER is semantic (sequential float32 sum, AudioNodeInput.cs:121-132): rebuild the list in order
            foreach (var (src, outIdx) in before) GraphAudioHip.Check(_native, GraphAudioHip.ga_node_disconnect(_native, _nodeIds[src], id, outIdx, i));
            foreach (var (src, outIdx) in now) GraphAudioHip.Check(_native, GraphAudioHip.ga_node_connect(_native, _nodeIds[src], id, outIdx, i));
            sh.Inputs[i] = now;
        }
    }

    private void RaiseEndedEvents()
    {
        foreach (var (node, id) in _nodeIds)
            if (node is AudioBufferSourceNode src && GraphAudioHip.ga_node_has_ended(_native, id) == 1)
                src.RaiseEndedFromNative();            // raises Ended once and Dispose()s, like TryRaiseEndedEvent (:378-389)
    }

    protected override void Dispose(bool disposing)
    {
        if (_native != IntPtr.Zero)
        {
            GraphAudioHip.ga_context_destroy(_native);
            _native = IntPtr.Zero;
        }
        base.Dispose(disposing);
    }
}
