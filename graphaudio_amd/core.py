"""Host-side mirror of the reference's AudioContextBase / AudioNode / AudioParam plugin surface.

The reference host language is C# (no .NET toolchain in this image), so this is the Python host above the
C ABI (include/graphaudio_hip.h).  Names, argument meaning and error behaviour follow the reference so that
graph-building code and tests read like reference code:

    ctx  = OfflineAudioContext(48000)                       # OfflineAudioContext.cs:18
    src  = AudioBufferSourceNode(ctx); src.Buffer = PlayableAudioBuffer.FromMonoArray(x, 48000)
    conv = ConvolverNode(ctx);         conv.Buffer = PlayableAudioBuffer.FromChannelArrays(ir, 48000)
    src.Connect(conv).Connect(ctx.Destination); src.Start()
    out  = ctx.Render(480000)                               # OfflineAudioContext.cs:108

The host is deliberately thin: every call is forwarded to the native library, which owns all semantics
(command queue, clamping, event ordering, scheduling).  The C# binding in bindings/csharp/ has the same shape.
"""
from __future__ import annotations

import ctypes as C
import enum
import math
import weakref
from typing import List, Optional, Sequence

import numpy as np

from . import _capi
from ._capi import (ArgumentException, ArgumentOutOfRangeException, CApi, InvalidOperationException,
                    ObjectDisposedException)

FramesPerBlock = 128  # AudioBuffer.FramesPerBlock, AudioBuffer.cs:10


class FilterType(enum.IntEnum):  # BiQuadFilterNode.cs:288-298
    Lowpass = 0
    Highpass = 1
    Bandpass = 2
    Notch = 3
    Allpass = 4
    Peaking = 5
    Lowshelf = 6
    Highshelf = 7


class ChannelCountMode(enum.IntEnum):  # AudioNodeInput.cs:258-272
    Max = 0
    ClampedMax = 1
    Explicit = 2


class ChannelInterpretation(enum.IntEnum):  # AudioNodeInput.cs:246-256
    Speakers = 0
    Discrete = 1


class AutomationRate(enum.IntEnum):  # AudioParam.cs:381-392
    ARate = 0
    KRate = 1


class PlayableAudioBuffer:
    """Immutable planar sample storage (PlayableAudioBuffer.cs:11-175).  Uploaded to a context on first use."""

    def __init__(self, channels: Sequence[np.ndarray], sampleRate: int):
        if len(channels) == 0:
            raise ArgumentException("Channel data cannot be or empty")
        if len(channels) > 32:
            raise ArgumentOutOfRangeException("Channel count must be between 1 and 32")
        if sampleRate <= 0:
            raise ArgumentOutOfRangeException("Sample rate must be positive")
        chans = [np.ascontiguousarray(c, dtype=np.float32).reshape(-1) for c in channels]
        n = chans[0].shape[0]
        for c in chans[1:]:
            if c.shape[0] != n:
                raise ArgumentException("All channels must have the same length")
        self._channels = chans
        self._sampleRate = int(sampleRate)
        self._ids = {}  # id(context) -> (weakref to the context, native buffer id)

    @staticmethod
    def FromChannelArrays(channelData, sampleRate: int) -> "PlayableAudioBuffer":  # :122-145
        return PlayableAudioBuffer(list(channelData), sampleRate)

    @staticmethod
    def FromMonoArray(audioData, sampleRate: int) -> "PlayableAudioBuffer":  # :150-159
        return PlayableAudioBuffer([audioData], sampleRate)

    @staticmethod
    def FromStereoArrays(left, right, sampleRate: int) -> "PlayableAudioBuffer":  # :164-174
        if len(left) != len(right):
            raise ArgumentException("Left and right channels must have the same length")
        return PlayableAudioBuffer([left, right], sampleRate)

    NumberOfChannels = property(lambda self: len(self._channels))
    Length = property(lambda self: int(self._channels[0].shape[0]))
    SampleRate = property(lambda self: self._sampleRate)
    Duration = property(lambda self: self.Length / float(self._sampleRate))
    IsInitialized = property(lambda self: True)

    def GetChannelData(self, channelIndex: int) -> np.ndarray:
        if channelIndex < 0 or channelIndex >= len(self._channels):
            raise ArgumentOutOfRangeException("channelIndex")
        return self._channels[channelIndex]

    def _native_id(self, ctx: "AudioContextBase") -> int:
        ent = self._ids.get(id(ctx))
        if ent is not None and ent[0]() is ctx:
            return ent[1]
        ptrs = (C.c_void_p * len(self._channels))(*[c.ctypes.data for c in self._channels])
        out = C.c_int(-1)
        ctx._call("buffer_create", ptrs, len(self._channels), self.Length, self._sampleRate, C.byref(out))
        self._ids[id(ctx)] = (weakref.ref(ctx), out.value)
        return out.value

    def __del__(self):
        # the managed object is gone: tell every live context that still holds a device copy (ga_buffer_release); nodes that
        # still play / convolve with it keep the storage alive natively
        try:
            for ref, bid in list(self._ids.values()):
                ctx = ref()
                if ctx is not None and getattr(ctx, "_h", None):
                    ctx._api.buffer_release(ctx._h, bid)
        except Exception:  # interpreter shutdown
            pass


class AudioParam:
    """AudioParam (AudioParam.cs:11-392): value + automation timeline, evaluated natively."""

    def __init__(self, owner: "AudioNode", index: int, name: str, defaultValue: float, minValue: float,
                 maxValue: float, rate: AutomationRate):
        self._owner, self._index = owner, index
        self.Name, self.DefaultValue, self.MinValue, self.MaxValue = name, defaultValue, minValue, maxValue
        self.AutomationRate = rate

    def _c(self, fn, *args):
        return self._owner.Context._call(fn, self._owner._id, self._index, *args)

    @property
    def Value(self) -> float:  # :34-36
        out = C.c_float(0)
        self._c("param_get_value", C.byref(out))
        return out.value

    @Value.setter
    def Value(self, v: float):  # :37-48 -- clamps, cancels all scheduled events
        self._c("param_set_value", float(v))

    def SetValueAtTime(self, value: float, startTime: float):  # :252-261
        self._c("param_set_value_at_time", float(value), float(startTime))

    def LinearRampToValueAtTime(self, value: float, endTime: float):  # :266-275
        self._c("param_linear_ramp_to_value_at_time", float(value), float(endTime))

    def ExponentialRampToValueAtTime(self, value: float, endTime: float):  # :280-292
        self._c("param_exponential_ramp_to_value_at_time", float(value), float(endTime))

    def SetTargetAtTime(self, target: float, startTime: float, timeConstant: float):  # :297-307
        self._c("param_set_target_at_time", float(target), float(startTime), float(timeConstant))

    def CancelScheduledValues(self, cancelTime: float):  # :312-331
        self._c("param_cancel_scheduled_values", float(cancelTime))


class AudioNodeInput:
    """AudioNodeInput (AudioNodeInput.cs:11-98): public channel configuration of one input port."""

    def __init__(self, owner: "AudioNode", index: int):
        self.Owner, self.Index = owner, index

    def SetChannelCount(self, count: int):  # :41-48
        self.Owner.Context._call("input_set_channel_count", self.Owner._id, self.Index, int(count))

    def SetChannelCountMode(self, mode: ChannelCountMode):  # :55-58
        self.Owner.Context._call("input_set_channel_count_mode", self.Owner._id, self.Index, int(mode))

    def SetChannelInterpretation(self, interpretation: ChannelInterpretation):  # :50-53
        self.Owner.Context._call("input_set_channel_interpretation", self.Owner._id, self.Index, int(interpretation))


class AudioNode:
    """AudioNode (Nodes/AudioNode.cs:10-239)."""

    _node_type: int = -1
    _input_count = 1
    _output_count = 1
    _ctor_arg = None   # constructor argument of nodes that take one (node_create_ex)

    def __init__(self, context: "AudioContextBase", name: Optional[str] = None, _id: Optional[int] = None):
        self.Context = context
        self.Name = name or type(self).__name__
        if _id is None:
            out = C.c_int(-1)
            if self._ctor_arg is None:
                context._call("node_create", self._node_type, C.byref(out))
            else:
                context._call("node_create_ex", self._node_type, float(self._ctor_arg), C.byref(out))
            _id = out.value
        self._id = _id
        self.NodeId = _id
        self.Inputs: List[AudioNodeInput] = [AudioNodeInput(self, i) for i in range(self._input_count)]
        self._params: List[AudioParam] = []
        context._nodes[_id] = self

    def _param(self, name, default, mn, mx, rate) -> AudioParam:
        p = AudioParam(self, len(self._params), name, default, mn, mx, rate)
        self._params.append(p)
        return p

    def Connect(self, destination, outputIndex: int = 0, inputIndex: int = 0):
        """Connect(AudioNode) returns the destination for chaining (:68-73); Connect(AudioParam) (:86-92)."""
        if isinstance(destination, AudioParam):
            self.Context._call("node_connect_param", self._id, destination._owner._id, destination._index, outputIndex)
            return None
        self.Context._call("node_connect", self._id, destination._id, outputIndex, inputIndex)
        return destination

    def Disconnect(self, destination=None, outputIndex: int = 0, inputIndex: int = 0):  # :78-81, :97-103
        if isinstance(destination, AudioParam):
            self.Context._call("node_disconnect_param", self._id, destination._owner._id, destination._index, outputIndex)
            return
        self.Context._call("node_disconnect", self._id, -1 if destination is None else destination._id, outputIndex,
                           inputIndex)

    def Dispose(self):  # :207-238
        self.Context._call("node_dispose", self._id)
        self.Context._forget(self)


class AudioDestinationNode(AudioNode):  # Nodes/AudioDestinationNode.cs:9-75
    _node_type = 0
    _output_count = 0

    def SetChannelCount(self, channels: int):  # :23-32
        self.Context._call("destination_set_channel_count", int(channels))


class GainNode(AudioNode):  # Nodes/GainNode.cs:9-71
    _node_type = 2

    def __init__(self, context):
        super().__init__(context, "Gain")
        self.Gain = self._param("gain", 1.0, -3.4028235e38, 3.4028235e38, AutomationRate.ARate)


class BiQuadFilterNode(AudioNode):  # Nodes/BiQuadFilterNode.cs:10-298
    _node_type = 3

    def __init__(self, context):
        super().__init__(context, "BiQuadFilter")
        self._type = FilterType.Lowpass
        self.Frequency = self._param("frequency", 1000.0, 1.0, context.SampleRate / 2.0, AutomationRate.ARate)
        self.Q = self._param("Q", 1.0, 0.001, 1000.0, AutomationRate.ARate)
        self.Gain = self._param("gain", 0.0, -60.0, 60.0, AutomationRate.KRate)

    @property
    def Type(self) -> FilterType:
        return self._type

    @Type.setter
    def Type(self, value: FilterType):  # :21-37
        self.Context._call("biquad_set_type", self._id, int(value))
        self._type = FilterType(value)


class ConvolverNode(AudioNode):  # Nodes/ConvolverNode.cs:10-176
    _node_type = 4

    def __init__(self, context):
        super().__init__(context, "Convolver")
        self._buffer: Optional[PlayableAudioBuffer] = None
        self._normalize = True
        self._true_stereo = True

    @property
    def Normalize(self) -> bool:  # :87
        return self._normalize

    @Normalize.setter
    def Normalize(self, v: bool):
        self.Context._call("convolver_set_normalize", self._id, 1 if v else 0)
        self._normalize = bool(v)

    @property
    def EnableTrueStereo(self) -> bool:  # :95
        return self._true_stereo

    @EnableTrueStereo.setter
    def EnableTrueStereo(self, v: bool):
        self.Context._call("convolver_set_enable_true_stereo", self._id, 1 if v else 0)
        self._true_stereo = bool(v)

    @property
    def Buffer(self) -> Optional[PlayableAudioBuffer]:
        return self._buffer

    @Buffer.setter
    def Buffer(self, value: Optional[PlayableAudioBuffer]):  # :25-79
        # the `_buffer == value` early-out (:30) is decided natively: it compares with the last EXECUTED swap
        bid = -1 if value is None else value._native_id(self.Context)
        self.Context._call("convolver_set_buffer", self._id, bid)
        self._buffer = value


class AudioBufferSourceNode(AudioNode):  # Nodes/AudioBufferSourceNode.cs:13-415
    _node_type = 1
    _input_count = 0

    def __init__(self, context):
        super().__init__(context, "AudioBufferSource")
        self.PlaybackRate = self._param("playbackRate", 1.0, 0.001, 1000.0, AutomationRate.KRate)
        self._buffer: Optional[PlayableAudioBuffer] = None
        self._loop, self._loop_start, self._loop_end = False, 0.0, 0.0
        self.Ended = []  # list of callables(sender) -- the C# event (:34)

    def _push_loop(self):
        self.Context._call("source_set_loop", self._id, 1 if self._loop else 0, self._loop_start, self._loop_end)

    Loop = property(lambda self: self._loop)
    LoopStart = property(lambda self: self._loop_start)
    LoopEnd = property(lambda self: self._loop_end)

    @Loop.setter
    def Loop(self, v: bool):  # :39-43
        self._loop = bool(v)
        self._push_loop()

    @LoopStart.setter
    def LoopStart(self, v: float):  # :48-52
        self._loop_start = max(0.0, float(v))
        self._push_loop()

    @LoopEnd.setter
    def LoopEnd(self, v: float):  # :57-61
        self._loop_end = max(0.0, float(v))
        self._push_loop()

    @property
    def Buffer(self) -> Optional[PlayableAudioBuffer]:
        return self._buffer

    @Buffer.setter
    def Buffer(self, value: Optional[PlayableAudioBuffer]):  # :67-71
        bid = -1 if value is None else value._native_id(self.Context)
        self.Context._call("source_set_buffer", self._id, bid)
        self._buffer = value

    def Start(self, when: float = 0.0, offset: float = 0.0, duration: float = math.inf):  # :79-114
        self.Context._call("source_start", self._id, float(when), float(offset), float(duration))

    def Stop(self, when: float = 0.0):  # :116-129
        self.Context._call("source_stop", self._id, float(when))


class StreamState(enum.IntEnum):  # GraphAudio.IO/AudioStreamSourceNodeBase.cs:12-17
    Playing = 0
    Paused = 1
    Stopped = 2


class AudioStreamSourceNode(AudioNode):
    """AudioStreamNodeBase (GraphAudio.IO/AudioStreamSourceNodeBase.cs:19-329) with an explicit queue: the host calls
    QueueBuffer (protected in the reference, where a decoder thread feeds it) so that renders are deterministic."""
    _node_type = 11
    _input_count = 0

    def __init__(self, context):
        super().__init__(context, "AudioStreamSource")
        self.PlaybackRate = self._param("playbackRate", 1.0, 0.001, 1000.0, AutomationRate.KRate)
        self._buffers = {}   # native id -> PlayableAudioBuffer (keeps queued buffers alive on the host side)

    @property
    def State(self) -> "StreamState":
        return self._state if hasattr(self, "_state") else StreamState.Stopped

    def _set(self, st):
        self.Context._call("stream_set_state", self._id, int(st))
        self._state = StreamState(st)

    def Play(self):   # :70-73
        self._set(StreamState.Playing)

    def Pause(self):  # :78-81
        self._set(StreamState.Paused)

    def Stop(self):   # :86-89
        self._set(StreamState.Stopped)

    def QueueBuffer(self, buffer: "PlayableAudioBuffer"):  # :118-124
        bid = buffer._native_id(self.Context)
        self._buffers[bid] = buffer
        self.Context._call("stream_queue_buffer", self._id, bid)

    def TryDequeueProcessedBuffer(self):  # :126-129
        out = C.c_int(-1)
        got = self.Context._call("stream_dequeue_processed", self._id, C.byref(out))
        return self._buffers.get(out.value) if got else None

    @property
    def QueuedBufferCount(self) -> int:
        return self.Context._call("stream_queued_count", self._id)

    @property
    def ProcessedBufferCount(self) -> int:
        return self.Context._call("stream_processed_count", self._id)


class OscillatorType(enum.IntEnum):  # OscillatorNode.cs:207-213
    Sine = 0
    Square = 1
    Sawtooth = 2
    Triangle = 3


class ChannelSplitterNode(AudioNode):  # Nodes/ChannelSplitterNode.cs:9-71
    _node_type = 5

    def __init__(self, context, numberOfOutputs: int = 2):
        self._ctor_arg = int(numberOfOutputs)
        self._output_count = int(numberOfOutputs)
        super().__init__(context, "ChannelSplitter")


class ChannelMergerNode(AudioNode):  # Nodes/ChannelMergerNode.cs:9-65
    _node_type = 6

    def __init__(self, context, numberOfInputs: int = 2):
        self._ctor_arg = int(numberOfInputs)
        self._input_count = max(int(numberOfInputs), 0)
        super().__init__(context, "ChannelMerger")


class _ScheduledSource(AudioNode):
    """IAudioScheduledSourceNode members shared by ConstantSourceNode and OscillatorNode."""
    _input_count = 0

    def Start(self, when: float = 0.0, offset: float = 0.0, duration: float = math.nan):
        self.Context._call("source_start", self._id, float(when), float(offset), float(duration))

    def Stop(self, when: float = 0.0):
        self.Context._call("source_stop", self._id, float(when))


class ConstantSourceNode(_ScheduledSource):  # Nodes/ConstantSourceNode.cs:15-163
    _node_type = 7

    def __init__(self, context):
        super().__init__(context, "ConstantSource")
        fmax = float(np.finfo(np.float32).max)
        self.Offset = self._param("offset", 1.0, -fmax, fmax, AutomationRate.ARate)
        self.Ended = []


class StereoPannerNode(AudioNode):  # Nodes/StereoPannerNode.cs:9-163
    _node_type = 8

    def __init__(self, context):
        super().__init__(context, "StereoPanner")
        self.Pan = self._param("pan", 0.0, -1.0, 1.0, AutomationRate.ARate)


class OscillatorNode(_ScheduledSource):  # Nodes/OscillatorNode.cs:12-214
    _node_type = 9

    def __init__(self, context):
        super().__init__(context, "Oscillator")
        self.Frequency = self._param("frequency", 440.0, 0.0, context.SampleRate / 2.0, AutomationRate.ARate)
        self._type = OscillatorType.Sine
        self.Ended = []

    @property
    def Type(self) -> OscillatorType:
        return self._type

    @Type.setter
    def Type(self, value: OscillatorType):  # :33-42
        self.Context._call("oscillator_set_type", self._id, int(value))
        self._type = OscillatorType(int(value))


class DelayNode(AudioNode):  # Nodes/DelayNode.cs:9-150
    _node_type = 10

    def __init__(self, context, maxDelayTime: float = 1.0):
        self._ctor_arg = float(maxDelayTime)
        super().__init__(context, "Delay")
        self.DelayTime = self._param("delayTime", 0.0, 0.0, float(maxDelayTime), AutomationRate.ARate)


class AudioContextBase:
    """AudioContextBase (AudioContextBase.cs:14-306) over a native context handle."""

    def __init__(self, sampleRate: int = 48000, device: int = 0, _api: Optional[CApi] = None):
        if sampleRate <= 0:
            raise ArgumentOutOfRangeException("sampleRate")  # :37-38
        self._api = _api if _api is not None else _capi.product_api()
        self.SampleRate = int(sampleRate)
        self._h = C.c_void_p()
        self._api.check(None, self._api.context_create(int(sampleRate), int(device), C.byref(self._h)))
        self._nodes = {}
        self._ended_seen = set()
        self.Destination = AudioDestinationNode(self, "AudioDestination", _id=0)

    def _call(self, fn: str, *args):
        if not self._h:
            raise ObjectDisposedException(type(self).__name__)
        return self._api.check(self._h, getattr(self._api, fn)(self._h, *args))

    @property
    def CurrentTime(self) -> float:  # :28
        return self._api.current_time(self._h)

    @property
    def CurrentBlock(self) -> int:  # :223
        return self._api.current_block(self._h)

    def ProcessBlocks(self, outputBuffers, blockCount: int):  # AudioContextBase.cs:163-186
        """Render `blockCount` whole blocks into planar float32 arrays (list of 1-D arrays or one 2-D array); per block only the
        channels the destination buffer has are written, `None` entries are skipped."""
        if blockCount < 0:
            raise ArgumentOutOfRangeException("blockCount")
        chans = list(outputBuffers)
        ptrs = (C.c_void_p * max(len(chans), 1))()
        for i, a in enumerate(chans):
            if a is None:
                ptrs[i] = None
                continue
            if a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"] or a.ndim != 1:
                raise ArgumentException("outputBuffers must hold contiguous 1-D float32 arrays")
            if a.shape[0] < blockCount * FramesPerBlock:
                raise ArgumentException("Destination is too short.")   # Span.CopyTo
            ptrs[i] = a.ctypes.data
        self._call("process_blocks", ptrs, len(chans), int(blockCount), 0)
        self._raise_ended()

    def ProcessBlockInterleaved(self, interleavedBuffer: np.ndarray, channels: int):  # AudioContextBase.cs:88-157
        """One block, frame-major interleaved: interleavedBuffer[frame * channels + ch]."""
        self.ProcessBlocksInterleaved(interleavedBuffer, channels, 1)

    def ProcessBlocksInterleaved(self, interleavedBuffer: np.ndarray, channels: int, blockCount: int):
        """`blockCount` consecutive ProcessBlockInterleaved calls in one native call (device-side interleave)."""
        if channels < 1 or channels > 32:
            raise ArgumentOutOfRangeException("channels")
        a = interleavedBuffer
        if a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"]:
            raise ArgumentException("interleavedBuffer must be a contiguous float32 array")
        if a.size < FramesPerBlock * channels * blockCount:
            raise ArgumentException("Buffer too small for interleaved output.")
        self._call("process_blocks_interleaved", a.ctypes.data_as(C.POINTER(C.c_float)), int(channels), int(blockCount), 0)
        self._raise_ended()

    def FramesToSeconds(self, frames: int) -> float:  # :228-231
        return frames / float(self.SampleRate)

    def SecondsToFrames(self, seconds: float) -> int:  # :236-239
        return int(seconds * self.SampleRate)

    def SetOption(self, key: str, value: float):
        self._call("set_option", key.encode(), float(value))
        if key == "async":
            self._async = bool(value)   # asynchronous renders: output arrays are kept alive until Synchronize()
            if not value:
                self._pending_outputs = []

    def GetStats(self) -> dict:
        st = _capi.Stats()
        self._call("get_stats", C.byref(st))
        return st.as_dict()

    def _forget(self, node):
        """A disposed node no longer needs to be reachable from the context (it only is for Ended dispatch); dropping it lets its
        PlayableAudioBuffer be collected, which releases the device copy."""
        if self._nodes.get(node._id) is node:
            del self._nodes[node._id]
        if hasattr(node, "_buffer"):
            node._buffer = None

    def _raise_ended(self):
        """Raise the Ended event (AudioBufferSourceNode.cs:378-389) of sources that finished during the last render."""
        buf = (C.c_int * 256)()
        while True:
            n = self._api.poll_ended(self._h, buf, 256)
            if n <= 0:
                break
            for i in range(n):
                node = self._nodes.get(buf[i])
                if node is not None and buf[i] not in self._ended_seen:
                    self._ended_seen.add(buf[i])
                    for cb in getattr(node, "Ended", []):
                        cb(node)
                    self._forget(node)   # Ended is followed by Dispose() (AudioBufferSourceNode.cs:386)
            if n < 256:
                break

    def Dispose(self):  # :243-260
        if self._h:
            self._api.context_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.Dispose()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.Dispose()


class OfflineAudioContext(AudioContextBase):
    """OfflineAudioContext (OfflineAudioContext.cs:8-158) rendered by the MI355X HIP path.

    ``Render(output, frameCount, startIndex=0)`` fills a list/array of per-channel float32 arrays
    (OfflineAudioContext.cs:30); ``Render(frameCount)`` allocates and returns ``[channels][frames]``
    (OfflineAudioContext.cs:108).
    """

    def Render(self, output, frameCount: Optional[int] = None, startIndex: int = 0):
        if frameCount is None:  # Render(int frameCount) overload
            frameCount = int(output)
            if frameCount <= 0:
                raise ArgumentOutOfRangeException("Frame count must be positive.")
            channels = self._api.destination_output_channels(self._h)
            out = np.zeros((channels, frameCount), dtype=np.float32)
            self.Render(out, frameCount)
            if getattr(self, "_async", False):   # the copy into `out` may still be in flight: never hand back a half-written array
                self._call("synchronize")
            return out
        if len(output) == 0:
            raise ArgumentException("Output buffer must have at least one channel.")
        if frameCount <= 0:
            raise ArgumentOutOfRangeException("Frame count must be positive.")
        if startIndex < 0:
            raise ArgumentOutOfRangeException("Start index must be non-negative.")
        rows = []
        for ch in range(len(output)):
            row = output[ch]
            if row is None:
                raise ArgumentException(f"Channel {ch} buffer is null.")
            if not (isinstance(row, np.ndarray) and row.dtype == np.float32 and row.flags["C_CONTIGUOUS"]):
                raise ArgumentException(f"Channel {ch} buffer must be a contiguous float32 array.")
            if row.shape[0] < startIndex + frameCount:
                raise ArgumentException(
                    f"Channel {ch} buffer is too small. Required: {startIndex + frameCount}, Available: {row.shape[0]}")
            rows.append(row)
        ptrs = (C.c_void_p * len(rows))(*[r.ctypes.data for r in rows])
        if getattr(self, "_async", False):
            # the device writes these rows after the call has returned: the context holds a reference until the next
            # Synchronize() / GetStats() so that a caller dropping its array cannot free memory a copy is still aimed at
            self._hold_output(rows)
        self._call("render", ptrs, len(rows), int(frameCount), int(startIndex))
        self._raise_ended()
        return output

    def RenderDevice(self, device_ptrs: Sequence[int], frameCount: int, startIndex: int = 0):
        """Render into DEVICE memory (one pointer per channel, e.g. ``tensor[ch].data_ptr()``)."""
        ptrs = (C.c_void_p * len(device_ptrs))(*[int(p) for p in device_ptrs])
        self._call("render_device", ptrs, len(device_ptrs), int(frameCount), int(startIndex))
        self._raise_ended()

    # ---- sharded render (include/graphaudio_hip.h "sharded render"): one context per GPU, one RCCL sum per Render ----
    COMM_ID_BYTES = 128

    def CommUniqueId(self) -> bytes:
        """The 128-byte communicator id rank 0 creates and hands to every rank (any channel)."""
        buf = C.create_string_buffer(self.COMM_ID_BYTES)
        self._api.check(self._h, self._api.comm_unique_id(buf))
        return buf.raw

    def CommInit(self, comm_id: Optional[bytes], n_ranks: int, rank: int):
        buf = C.create_string_buffer(bytes(comm_id), self.COMM_ID_BYTES) if comm_id is not None else None
        self._call("comm_init", buf, int(n_ranks), int(rank))

    def CommDestroy(self):
        self._call("comm_destroy")

    def CommInfo(self) -> dict:
        """What the communicator itself reports (ncclCommCount / ncclCommUserRank), not what CommInit was told."""
        n, r, u = C.c_int(0), C.c_int(0), C.c_int(0)
        self._call("comm_info", C.byref(n), C.byref(r), C.byref(u))
        return {"ranks": n.value, "rank": r.value, "uses_rccl": bool(u.value)}

    def RenderReduce(self, output, frameCount: int, startIndex: int = 0, root: int = 0):
        """Render(output, frameCount, startIndex) of a voice-sharded graph: every rank renders its share, the destination
        buses are summed on the device (RCCL), `output` (per-channel float32 arrays) is written on `root` only."""
        n = len(output)
        ptrs = (C.c_void_p * n)()
        for ch in range(n):
            row = output[ch]
            if not (isinstance(row, np.ndarray) and row.dtype == np.float32 and row.flags["C_CONTIGUOUS"]):
                raise ArgumentException(f"Channel {ch} buffer must be a contiguous float32 array.")
            if row.shape[0] < startIndex + frameCount:
                raise ArgumentException(f"Channel {ch} buffer is too small.")
            ptrs[ch] = row.ctypes.data
        if getattr(self, "_async", False):
            self._hold_output(output)   # alive until Synchronize()
        self._call("render_reduce", ptrs, n, int(frameCount), int(startIndex), int(root))
        self._raise_ended()

    def _hold_output(self, rows):
        """Asynchronous renders: EVERY output still in flight stays referenced until Synchronize() -- the device (a kernel that
        writes page-locked rows, or a copy) is aimed at that memory.  The same arrays rendered into again are held once; a
        caller that never synchronises is synchronised here every 64 distinct outputs instead of losing references."""
        pend = self.__dict__.setdefault("_pending_outputs", [])
        key = tuple(id(r) for r in rows)
        if any(k == key for k, _ in pend):
            return
        if len(pend) >= 64:
            self.Synchronize()
            pend = self._pending_outputs
        pend.append((key, rows))

    def SetStream(self, hip_stream: int):
        self._call("context_set_stream", C.c_void_p(int(hip_stream)))

    def Synchronize(self):
        """Wait for the renders enqueued under ``SetOption("async", 1)`` (include/graphaudio_hip.h, ga_synchronize)."""
        self._call("synchronize")
        self._pending_outputs = []


# The reference's class name for the stock CPU context is OfflineAudioContext; the drop-in replacement a C# user
# instantiates is HipOfflineAudioContext (bindings/csharp/HipOfflineAudioContext.cs).  Same object here.
HipOfflineAudioContext = OfflineAudioContext
