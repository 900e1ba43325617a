"""Timing events for the measurement modes of the engine (Engine.timeline, bench.py's per-launch table): HIP events created
with hipEventDisableSystemFence.  A torch.cuda.Event performs a system-scope release when it is recorded -- between every two
launches of a step that means an L2 write-back and every kernel starting on a colder cache: the MFMA kernels measured that
way came out 8-17 % longer than their own dispatch timestamps (round 5).  These events only take timestamps.  Same interface
as the part of torch.cuda.Event the engine uses (record, elapsed_time, cuda_event)."""
import ctypes
import os

import torch

_HIP = None
_DISABLE_SYSTEM_FENCE = 0x20000000  # hip_runtime_api.h: hipEventDisableSystemFence


def _hip():
    """the HIP runtime torch itself loaded (the process must hold exactly one): dlopen of the same file returns its handle"""
    global _HIP
    if _HIP is None:
        path = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        try:
            lib = ctypes.CDLL(path)
        except OSError:
            # a torch build linked against the system ROCm ships no copy of its own: the soname resolves to the runtime the
            # process already holds
            lib = ctypes.CDLL("libamdhip64.so")
        lib.hipEventCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
        lib.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lib.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
        lib.hipEventDestroy.argtypes = [ctypes.c_void_p]
        lib.hipEventSynchronize.argtypes = [ctypes.c_void_p]
        _HIP = lib
    return _HIP


class TimingEvent:
    def __init__(self, enable_timing=True):
        handle = ctypes.c_void_p()
        rc = _hip().hipEventCreateWithFlags(ctypes.byref(handle), _DISABLE_SYSTEM_FENCE)
        if rc != 0:
            raise RuntimeError("hipEventCreateWithFlags failed with status {}".format(rc))
        self.cuda_event = handle.value

    def record(self, stream=None):
        stream = stream if stream is not None else torch.cuda.current_stream()
        rc = _hip().hipEventRecord(self.cuda_event, stream.cuda_stream)
        if rc != 0:
            raise RuntimeError("hipEventRecord failed with status {}".format(rc))

    def elapsed_time(self, other):
        ms = ctypes.c_float()
        _hip().hipEventSynchronize(other.cuda_event)
        rc = _hip().hipEventElapsedTime(ctypes.byref(ms), self.cuda_event, other.cuda_event)
        if rc != 0:
            raise RuntimeError("hipEventElapsedTime failed with status {}".format(rc))
        return float(ms.value)

    def __del__(self):
        if getattr(self, "cuda_event", None) and _HIP is not None:
            _HIP.hipEventDestroy(self.cuda_event)
            self.cuda_event = None
