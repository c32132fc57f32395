"""ctypes binding of include/graphaudio_hip.h.

The product library is ``libgraphaudio_hip.so`` next to this file (built by ``__graft_entry__.build()`` /
``graphaudio_amd/csrc/Makefile``).  There is NO fallback: if the HIP library is missing or does not load,
importing a context raises ``GraphAudioLibraryError``.

``CApi`` is parameterised by (library, prefix) only so that tests can drive the CPU oracle -- which exports
the same header under the ``gao_`` prefix -- through the very same host code.  Nothing in this package
references ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libgraphaudio_hip.so"


class GraphAudioLibraryError(ImportError):
    """The native HIP library could not be loaded."""


# --- exception types named after the reference's (.NET) exceptions they mirror -------------------------
class ArgumentException(ValueError):
    """System.ArgumentException (OfflineAudioContext.cs:32-51)."""


class ArgumentOutOfRangeException(ArgumentException):
    """System.ArgumentOutOfRangeException (AudioBuffer.cs:18, AudioNodeInput.cs:43)."""


class InvalidOperationException(RuntimeError):
    """System.InvalidOperationException (ConvolverNode.cs:45-49, Nodes/AudioNode.cs:157-160)."""


class ObjectDisposedException(InvalidOperationException):
    """System.ObjectDisposedException (AudioContextBase.cs:54)."""


class NotSupportedException(RuntimeError):
    """The graph uses a feature outside the accelerated path (GA_ERR_UNSUPPORTED)."""


class DeviceException(RuntimeError):
    """HIP runtime / device failure (GA_ERR_DEVICE, GA_ERR_OUT_OF_MEMORY, GA_ERR_NO_DEVICE)."""


_CODE_TO_EXC = {
    -1: ArgumentException,
    -2: ArgumentOutOfRangeException,
    -3: InvalidOperationException,
    -4: ObjectDisposedException,
    -5: InvalidOperationException,
    -6: NotSupportedException,
    -7: DeviceException,
    -8: DeviceException,
    -9: DeviceException,
}


class Stats(C.Structure):
    _fields_ = [
        ("blocks_rendered", C.c_int64),
        ("chunks", C.c_int64),
        ("segments", C.c_int64),
        ("kernel_launches", C.c_int64),
        ("device_ms_total", C.c_double),
        ("mac_launches", C.c_int64),
        ("mac_ms_total", C.c_double),
        ("mac_flops_total", C.c_double),
        ("mac_bytes_total", C.c_double),
        ("fft_ms_total", C.c_double),
        ("other_ms_total", C.c_double),
        ("device_bytes_in_use", C.c_int64),
        ("n_nodes", C.c_int32),
        ("n_conv_rows", C.c_int32),
        ("stage_ms", C.c_double * 16),
        ("stage_launches", C.c_int64 * 16),
        ("stage_bytes", C.c_double * 16),
        ("profiled_chunks", C.c_int64),
        ("coarse_carried_outputs", C.c_int64),
        ("stage_flops", C.c_double * 16),
        ("stage_kernel", (C.c_char * 64) * 16),
        ("coarse_premixed_signals", C.c_int64),
        ("deferred_handovers", C.c_int64),
        ("biquad_split_cascades", C.c_int64),
        ("ref_order_rows", C.c_int64),
        ("sim_replays", C.c_int64),
        ("twin_rows", C.c_int64),
    ]
    STAGES = ("other", "mix", "rfft_fwd", "mac", "rfft_inv", "coarse_fwd", "coarse_mac", "coarse_inv", "coarse_hist", "coarse_section",
              "coarse_premix")   # index = GA_STAGE_* (tests/test_capi.py checks the length against GA_STAGE_COUNT)

    def as_dict(self):
        d = {}
        for name, _ in self._fields_:
            v = getattr(self, name)
            if name == "stage_kernel":
                d[name] = [bytes(row.value).decode("ascii", "replace") for row in v]
            else:
                d[name] = list(v) if hasattr(v, "__len__") else v
        return d


_vp, _i, _i64, _f, _d = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); one row per GA_FN() declaration in include/graphaudio_hip.h
SIGNATURES = {
    "strerror": (C.c_char_p, [_i]),
    "version": (C.c_char_p, []),
    "device_count": (_i, []),
    "context_create": (_i, [_i, _i, C.POINTER(_vp)]),
    "context_destroy": (_i, [_vp]),
    "last_error": (C.c_char_p, [_vp]),
    "current_time": (_d, [_vp]),
    "current_block": (_i64, [_vp]),
    "set_option": (_i, [_vp, C.c_char_p, _d]),
    "get_stats": (_i, [_vp, C.POINTER(Stats)]),
    "buffer_create": (_i, [_vp, _pp, _i, _i64, _i, C.POINTER(_i)]),
    "buffer_release": (_i, [_vp, _i]),
    "node_create": (_i, [_vp, _i, C.POINTER(_i)]),
    "node_create_ex": (_i, [_vp, _i, _d, C.POINTER(_i)]),
    "node_dispose": (_i, [_vp, _i]),
    "node_connect": (_i, [_vp, _i, _i, _i, _i]),
    "node_disconnect": (_i, [_vp, _i, _i, _i, _i]),
    "node_connect_param": (_i, [_vp, _i, _i, _i, _i]),
    "node_disconnect_param": (_i, [_vp, _i, _i, _i, _i]),
    "node_has_ended": (_i, [_vp, _i]),
    "poll_ended": (_i, [_vp, C.POINTER(_i), _i]),
    "input_set_channel_count": (_i, [_vp, _i, _i, _i]),
    "input_set_channel_count_mode": (_i, [_vp, _i, _i, _i]),
    "input_set_channel_interpretation": (_i, [_vp, _i, _i, _i]),
    "destination_set_channel_count": (_i, [_vp, _i]),
    "destination_output_channels": (_i, [_vp]),
    "param_set_value": (_i, [_vp, _i, _i, _f]),
    "param_get_value": (_i, [_vp, _i, _i, C.POINTER(_f)]),
    "param_set_value_at_time": (_i, [_vp, _i, _i, _f, _d]),
    "param_linear_ramp_to_value_at_time": (_i, [_vp, _i, _i, _f, _d]),
    "param_exponential_ramp_to_value_at_time": (_i, [_vp, _i, _i, _f, _d]),
    "param_set_target_at_time": (_i, [_vp, _i, _i, _f, _d, _d]),
    "param_cancel_scheduled_values": (_i, [_vp, _i, _i, _d]),
    "source_set_buffer": (_i, [_vp, _i, _i]),
    "source_set_loop": (_i, [_vp, _i, _i, _d, _d]),
    "source_start": (_i, [_vp, _i, _d, _d, _d]),
    "source_stop": (_i, [_vp, _i, _d]),
    "oscillator_set_type": (_i, [_vp, _i, _i]),
    "biquad_set_type": (_i, [_vp, _i, _i]),
    "convolver_set_normalize": (_i, [_vp, _i, _i]),
    "convolver_set_enable_true_stereo": (_i, [_vp, _i, _i]),
    "convolver_set_buffer": (_i, [_vp, _i, _i]),
    "render": (_i, [_vp, _pp, _i, _i64, _i64]),
    "render_device": (_i, [_vp, _pp, _i, _i64, _i64]),
    "process_blocks": (_i, [_vp, _pp, _i, _i64, _i]),
    "process_blocks_interleaved": (_i, [_vp, C.POINTER(C.c_float), _i, _i64, _i]),
    "context_set_stream": (_i, [_vp, _vp]),
    "synchronize": (_i, [_vp]),
    "stream_queue_buffer": (_i, [_vp, _i, _i]),
    "stream_set_state": (_i, [_vp, _i, _i]),
    "stream_dequeue_processed": (_i, [_vp, _i, C.POINTER(_i)]),
    "stream_queued_count": (_i, [_vp, _i]),
    "stream_processed_count": (_i, [_vp, _i]),
    "comm_unique_id": (_i, [_vp]),
    "comm_init": (_i, [_vp, _vp, _i, _i]),
    "comm_destroy": (_i, [_vp]),
    "comm_info": (_i, [_vp, C.POINTER(_i), C.POINTER(_i), C.POINTER(_i)]),
    "shard_range": (_i, [_i64, _i, _i, C.POINTER(_i64), C.POINTER(_i64)]),
    "render_reduce": (_i, [_vp, _pp, _i, _i64, _i64, _i]),
}


class CApi:
    """Typed access to one shared library exporting the graphaudio_hip.h surface under ``prefix``."""

    def __init__(self, lib: C.CDLL, prefix: str = "ga_"):
        self.lib = lib
        self.prefix = prefix
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(lib, prefix + name)  # AttributeError if a declared symbol is not exported
            fn.restype = restype
            fn.argtypes = argtypes
            setattr(self, name, fn)

    def check(self, ctx, code: int):
        """Turn a negative result code into the exception the reference would have thrown."""
        if code >= 0:
            return code
        msg = ""
        if ctx:
            raw = self.last_error(ctx)
            msg = raw.decode("utf-8", "replace") if raw else ""
        if not msg:
            msg = self.strerror(code).decode()
        raise _CODE_TO_EXC.get(code, RuntimeError)(msg)


_product_api = None


_library_override = None


def use_library(path: str) -> None:
    """Load a differently built library (kernel-variant measurements: tools/build_variant.sh, bench.py --library).  An explicit
    call before the first context -- no environment variable swaps the product library."""
    global _library_override, _product_api
    if _product_api is not None:
        raise RuntimeError("use_library() has to be called before the library is first used")
    _library_override = path


def library_path() -> str:
    return _library_override or os.path.join(_HERE, LIB_NAME)


def product_api() -> CApi:
    """The HIP product library.  Fails loudly when it is missing -- there is no CPU fallback."""
    global _product_api
    if _product_api is None:
        path = library_path()
        if not os.path.exists(path):
            raise GraphAudioLibraryError(
                f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). graphaudio_amd has no CPU fallback."
            )
        try:
            lib = C.CDLL(path)
        except OSError as e:  # pragma: no cover - depends on the machine
            raise GraphAudioLibraryError(f"cannot load {path}: {e}") from e
        _product_api = CApi(lib, "ga_")
    return _product_api
