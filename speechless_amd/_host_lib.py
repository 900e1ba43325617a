"""ctypes binding of include/speechless_host.h (libspeechless_host.so: batch packer, n-gram model, CTC beam search).
No torch here: these are plain host helpers.  Missing library or symbol -> raises."""
import ctypes
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int32, c_void_p
from pathlib import Path

LIB_PATH = Path(__file__).resolve().parent / "libspeechless_host.so"

# name -> (restype, argtypes); every symbol include/speechless_host.h declares
HOST_SIGNATURES = {
    "sl_host_version": (c_int, []),
    "sl_host_pack_batch": (c_int, [POINTER(c_void_p), POINTER(c_int32), c_int, c_int, c_int, c_int, c_void_p, c_int]),
    "sl_host_lm_load_arpa": (c_void_p, [c_char_p, c_char_p, c_int]),
    "sl_host_lm_free": (None, [c_void_p]),
    "sl_host_lm_order": (c_int, [c_void_p]),
    "sl_host_lm_score_sentence": (c_double, [c_void_p, c_char_p]),
    "sl_host_scorer_create": (c_void_p, [c_void_p, c_void_p, c_int, c_float, c_float, c_float]),
    "sl_host_scorer_free": (None, [c_void_p]),
    "sl_host_ctc_beam_search": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_int]),
}

_HOST_LIB = None


def host_lib():
    """The loaded library with typed entry points.  CDLL: ctypes releases the GIL while a call runs, so packing and
    beam search do not compete with the training thread."""
    global _HOST_LIB
    if _HOST_LIB is None:
        if not LIB_PATH.exists():
            raise RuntimeError("{} is missing: run `python -m speechless_amd.build`".format(LIB_PATH))
        lib = ctypes.CDLL(str(LIB_PATH))
        for name, (restype, argtypes) in HOST_SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is missing -> loud
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.sl_host_version() != 1:
            raise RuntimeError("libspeechless_host.so version mismatch")
        _HOST_LIB = lib
    return _HOST_LIB
