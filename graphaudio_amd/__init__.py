"""graphaudio_amd -- MI355X-native offline render path for GraphAudio's per-block DSP hot path.

Host-side mirror of the reference's AudioContextBase / AudioNode / AudioParam surface over the C ABI of
libgraphaudio_hip.so (include/graphaudio_hip.h).  No CPU fallback: the HIP library must be present.
"""
from ._capi import (ArgumentException, ArgumentOutOfRangeException, DeviceException, GraphAudioLibraryError,
                    InvalidOperationException, NotSupportedException, ObjectDisposedException, library_path,
                    product_api)
from .core import (AudioStreamSourceNode, StreamState, AudioBufferSourceNode, AudioContextBase, AudioDestinationNode, AudioNode, AudioNodeInput,
                   AudioParam, AutomationRate, BiQuadFilterNode, ChannelCountMode, ChannelInterpretation,
                   ChannelMergerNode, ChannelSplitterNode, ConstantSourceNode, ConvolverNode, DelayNode, FilterType,
                   FramesPerBlock, GainNode, HipOfflineAudioContext, OfflineAudioContext, OscillatorNode,
                   OscillatorType, PlayableAudioBuffer, StereoPannerNode)

__all__ = [n for n in dir() if not n.startswith("_")]
