"""ctypes binding of libdvslam_rccl.so (include/dvslam_rccl.h): RCCL sum all-reduce of the gradient arena on a stream of
the caller's choice.  Loaded only when the direct all-reduce path is selected (dp.RcclComm); no fallback."""
import ctypes as C
import os

import torch  # noqa: F401  (torch's librccl.so.1 / libamdhip64 first: one runtime per process)

from ._lib import DvsError

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libdvslam_rccl.so")
UNIQUE_ID_BYTES = 128
_vp = C.c_void_p

_SIGNATURES = {
    "dvs_rccl_last_error": (C.c_char_p, []),
    "dvs_allreduce_unique_id": (C.c_int, [_vp]),
    "dvs_allreduce_init": (C.c_int, [C.POINTER(_vp), _vp, C.c_int, C.c_int]),
    "dvs_allreduce_run": (C.c_int, [_vp, _vp, C.c_size_t, _vp]),
    "dvs_allreduce_run_ranges": (C.c_int, [_vp, _vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_int, _vp]),
    "dvs_allreduce_world": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dvs_allreduce_destroy": (C.c_int, [_vp]),
}
_lib = None


def exported_symbols():
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DvsError("libdvslam_rccl.so is not built (%s): run __graft_entry__.build()" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype, fn.argtypes = res, args
        _lib = l
    return _lib


def check(rc, what):
    if rc != 0:
        raise DvsError("%s failed (%d): %s" % (what, rc, lib().dvs_rccl_last_error().decode()))
