"""ctypes binding of include/speechless_hip.h.  There is NO fallback: if the HIP library is missing or a call
fails, this raises -- the product path never computes on the CPU."""
import ctypes
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from pathlib import Path

# torch FIRST: its wheel bundles its own libamdhip64.so.7 / libhsa-runtime64.  libspeechless_hip.so needs the same
# SONAME, and a process must hold exactly ONE HIP runtime -- the one that owns the tensors' memory and streams.
# Loading ours first would pull /opt/rocm's runtime in and torch would then find "no ROCm-capable device".
import torch  # noqa: F401  (plumbing: device memory, streams, torch.distributed)

import os

# SL_LIB_PATH: experiments only (a probe build of the library next to the real one)
LIB_PATH = Path(os.environ.get("SL_LIB_PATH") or Path(__file__).resolve().parent / "libspeechless_hip.so")

SL_BF16 = 0
SL_F32 = 1
SL_F16 = 2

EPI_NONE = 0
EPI_BIAS = 1
EPI_BIAS_RELU = 2
EPI_RELU_MASK = 3
EPI_BIAS_ELU = 4
EPI_ELU_MASK = 5


class AdamLayer(ctypes.Structure):
    """Mirror of sl_adam_layer (include/speechless_hip.h)."""
    _fields_ = [("offset", c_int64), ("w_fwd", c_void_p), ("w_dgrad", c_void_p), ("k", c_int32), ("cin_pad", c_int32),
                ("cout_pad", c_int32)]


class BgwLayer(ctypes.Structure):
    """Mirror of sl_bgw_layer (include/speechless_hip.h)."""
    _fields_ = [("w_off", c_int64), ("b_off", c_int64), ("k", c_int32), ("cin_pad", c_int32), ("cout_pad", c_int32),
                ("tap", c_int32)]


class ConvGeom(ctypes.Structure):
    """Mirror of sl_conv_geom (include/speechless_hip.h)."""
    _fields_ = [
        ("batch", c_int32),
        ("t_out", c_int32),
        ("taps", c_int32),
        ("cin", c_int32),
        ("cout", c_int32),
        ("x_row0", c_int32),
        ("x_row_stride", c_int32),
        ("x_batch_stride", c_int64),
        ("y_row0", c_int32),
        ("y_row_stride", c_int32),
        ("y_batch_stride", c_int64),
        ("acc_scale", c_float),
    ]


class WgradJob(ctypes.Structure):
    """Mirror of sl_wgrad_job (include/speechless_hip.h)."""
    _fields_ = [("x", c_void_p), ("g", c_void_p), ("dw", c_void_p), ("geom", ConvGeom)]


# name -> (restype, argtypes); every symbol include/speechless_hip.h declares
SIGNATURES = {
    "sl_version": (c_int, []),
    "sl_profile_next_kernel": (c_int, [c_void_p, c_void_p]),
    "sl_output_softmax_supported": (c_int, [POINTER(ConvGeom), c_int, c_int]),
    "sl_output_softmax_select": (c_int, [c_int]),
    "sl_output_softmax": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvGeom), c_int,
                                  c_int, c_int64, c_float, c_int, c_void_p]),
    "sl_last_error": (c_char_p, []),
    "sl_conv1d_nt_workspace_bytes": (c_size_t, [POINTER(ConvGeom), c_int, c_int]),
    "sl_conv1d_nt": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvGeom), c_int, c_int, c_int,
                             c_int, c_void_p, c_size_t, c_void_p]),
    "sl_conv1d_chain_supported": (c_int, [POINTER(ConvGeom), c_int, c_int]),
    "sl_conv1d_chain_select": (c_int, [c_int]),
    "sl_conv1d_chain": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                POINTER(ConvGeom), c_int, c_int, c_int, c_void_p]),
    "sl_conv1d_wgrad_workspace_bytes": (c_size_t, [POINTER(ConvGeom), c_int, c_int]),
    "sl_conv1d_wgrad": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(ConvGeom), c_int, c_int, c_void_p, c_size_t,
                                c_void_p]),
    "sl_conv1d_wgrad_grouped_workspace_bytes": (c_size_t, [POINTER(ConvGeom), c_int, c_int]),
    "sl_conv1d_wgrad_grouped": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(ConvGeom), c_int, c_int64, c_int64,
                                        c_int64, c_int, c_void_p, c_size_t, c_void_p]),
    "sl_bias_grad_workspace_bytes": (c_size_t, [POINTER(ConvGeom)]),
    "sl_bias_grad": (c_int, [c_void_p, c_void_p, POINTER(ConvGeom), c_int, c_void_p, c_size_t, c_void_p]),
    "sl_pack_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "sl_dropout": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_float, ctypes.c_uint64, c_void_p]),
    "sl_scale": (c_int, [c_void_p, c_size_t, c_int, c_float, c_void_p]),
    "sl_elu_dropout_backward": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_float, ctypes.c_uint64, c_void_p]),
    "sl_pack_input": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64, c_int, c_void_p]),
    "sl_wave_frames": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int64, c_int, c_void_p]),
    "sl_pack_input_ones": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64, c_int, c_int, c_void_p]),
    "sl_softmax_logq": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_float,
                                c_void_p]),
    "sl_ctc_select": (c_int, [c_int]),
    "sl_ctc_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "sl_ctc_loss_grad": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int,
                                 c_int, c_int, c_int, c_int, c_int64, c_int, c_float, c_float, c_void_p, c_size_t,
                                 c_void_p]),
    "sl_greedy_decode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                 c_void_p]),
    "sl_adam_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_float, c_float, c_float,
                             c_float, c_void_p]),
    "sl_adam_pack_layer": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int, c_int, c_float, c_float, c_float, c_float, c_void_p]),
    "sl_stft_power_db": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64,
                                 c_float, c_void_p]),
    "sl_z_normalize_workspace_bytes": (c_size_t, [c_int]),
    "sl_z_normalize": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_void_p, c_size_t,
                               c_void_p]),
    "sl_bias_grad_from_wgrad": (c_int, [c_void_p, POINTER(BgwLayer), c_int, c_int, c_void_p]),
    "sl_adam_pack_layers": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(AdamLayer), c_int, c_int, c_int,
                                    c_float, c_float, c_float, c_float, c_void_p]),
    "sl_set_available_cus": (c_int, [c_int]),
    "sl_conv1d_backward_1x1_supported": (c_int, [POINTER(ConvGeom), c_int, c_int]),
    "sl_conv1d_backward_1x1_workspace_bytes": (c_size_t, [POINTER(ConvGeom), c_int, c_int, c_int]),
    "sl_conv1d_backward_1x1": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvGeom), c_int, c_int,
                                       c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "sl_conv1d_backward_1x1_part": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(ConvGeom), c_int, c_int,
                                            c_int, c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "sl_split3": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int64, c_int, c_void_p]),
    "sl_split3_pack_input": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "sl_split3_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "sl_split3_assemble": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "sl_split3_pack_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "sl_split3_adam_pack_layers": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(AdamLayer), c_int, c_int, c_float,
                                           c_float, c_float, c_float, c_void_p]),
    "sl_split3_wgrad_combine": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                        c_void_p]),
    "sl_split3_bias_grad_workspace_bytes": (c_size_t, [c_int]),
    "sl_split3_dropout": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, ctypes.c_uint64, c_void_p]),
    "sl_split3_bias_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_void_p, c_size_t, c_void_p]),
    "sl_conv1d_wgrad_multi_workspace_bytes": (c_size_t, [POINTER(WgradJob), c_int, c_int]),
    "sl_conv1d_wgrad_multi": (c_int, [POINTER(WgradJob), c_int, c_int, c_void_p, c_size_t, c_void_p]),
    "sl_pack_layers": (c_int, [c_void_p, POINTER(AdamLayer), c_int, c_int, c_void_p]),
    "sl_splitf16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int64, c_int, c_int64, c_int, c_void_p]),
    "sl_splitf16_pack_input": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int64, c_void_p]),
    "sl_splitf16_pack_weights": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "sl_splitf16_adam_pack_layers": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, POINTER(AdamLayer), c_int, c_int, c_float,
                                             c_float, c_float, c_float, c_float, c_void_p]),
    "sl_split3_wgrad_combine_scaled": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                               c_int, c_float, c_void_p]),
    "sl_splitf16_bias_grad": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int64, c_float, c_void_p, c_size_t,
                                      c_void_p]),
    "sl_splitf16_dropout": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_float, ctypes.c_uint64, c_void_p]),
}


class HipLibraryError(RuntimeError):
    pass


class HipLibrary:
    """Loaded libspeechless_hip.so with typed entry points.  `call(name, *args)` raises on a non-zero status."""

    def __init__(self, path=LIB_PATH):
        path = Path(path)
        if not path.exists():
            raise HipLibraryError(
                "{} not found: build it with `python -m speechless_amd.build` (or __graft_entry__.build()). "
                "The speechless_amd hot path has no CPU fallback.".format(path))
        self.path = path
        # PyDLL: the GIL is NOT released around a call.  Every entry point only enqueues work (microseconds); with CDLL
        # each of the ~60 launches of a step hands the GIL to the input pipeline's worker threads and has to win it back
        # (up to the interpreter's 5 ms switch interval) before the next launch can be issued.
        self._dll = ctypes.PyDLL(str(path)) if os.environ.get("SL_RELEASE_GIL", "0") != "1" else ctypes.CDLL(str(path))
        self._fn = {}
        for name, (restype, argtypes) in SIGNATURES.items():
            fn = getattr(self._dll, name)  # AttributeError if the symbol is missing -> loud
            fn.restype = restype
            fn.argtypes = argtypes
            self._fn[name] = fn
        if self._fn["sl_version"]() != 1:
            raise HipLibraryError("libspeechless_hip.so version mismatch")

    def raw(self, name):
        return self._fn[name]

    def last_error(self):
        msg = self._fn["sl_last_error"]()
        return msg.decode("utf8", "replace") if msg else ""

    def call(self, name, *args):
        rc = self._fn[name](*args)
        if rc != 0:
            raise HipLibraryError("{} failed with status {}: {}".format(name, rc, self.last_error()))


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        _LIB = HipLibrary()
    return _LIB
