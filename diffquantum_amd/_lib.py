"""ctypes loader for libdiffqc_hip.so (the C-ABI declared in include/diffqc_hip.h).

The shared object is built in-tree by ``__graft_entry__.build()`` /
``make -C diffquantum_amd/csrc``.  There is no fallback: if the library is missing the
import of anything that needs it raises, and if no GPU is present ``dq_create`` returns
DQ_ERR_NO_DEVICE which surfaces as ``DiffqcNoDevice``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdiffqc_hip.so")

DQ_C128, DQ_C64 = 0, 1
DQ_OK, DQ_ERR_INVALID, DQ_ERR_STATE, DQ_ERR_HIP, DQ_ERR_NO_DEVICE, DQ_ERR_UNSUPPORTED, DQ_ERR_NOMEM = range(7)
PROF_CLASSES = ("zgemm", "assemble", "lincomb", "gradc", "other", "rgemm", "rsym", "zherm", "spmm", "statecol_fwd", "statecol_bwd")


class DiffqcError(RuntimeError):
    pass


class DiffqcNoDevice(DiffqcError):
    pass


class DqProfile(C.Structure):
    _fields_ = [
        ("launches", C.c_longlong * len(PROF_CLASSES)),
        ("ms", C.c_double * len(PROF_CLASSES)),
        ("flops", C.c_double * len(PROF_CLASSES)),
        ("bytes", C.c_double * len(PROF_CLASSES)),
    ]


_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); every entry must be exported by the library
SIGNATURES = {
    "dq_version": (C.c_char_p, []),
    "dq_last_error": (C.c_char_p, []),
    "dq_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "dq_step_polynomial": (C.c_int, [C.c_double, C.c_double, C.c_int, C.POINTER(C.c_int), _dp]),
    "dq_quadrature_rule": (C.c_int, [C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), _dp, _dp, _dp, _dp]),
    "dq_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int]),
    "dq_create_ex": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int]),
    "dq_destroy": (C.c_int, [_vp]),
    "dq_set_stream": (C.c_int, [_vp, _vp]),
    "dq_synchronize": (C.c_int, [_vp]),
    "dq_set_hamiltonians": (C.c_int, [_vp, _vp, _vp]),
    "dq_set_observable": (C.c_int, [_vp, _vp]),
    "dq_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "dq_get_info": (C.c_int, [_vp, C.c_char_p, _dp]),
    "dq_propagate": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp, _vp, C.c_int]),
    "dq_energy": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "dq_energy_grad": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "dq_gram": (C.c_int, [_vp, _vp, C.c_int, _vp]),
    "dq_energy_grad_columns": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "dq_fidelity_grad": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "dq_mc_shift_grad": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp, C.c_int, C.c_double, _vp, C.c_int, C.c_double,
                                   _vp, _vp]),
    "dq_slice_forward": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp]),
    "dq_slice_terminal": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "dq_slice_backward": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp, _vp, _vp]),
    "dq_build_propagators": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp]),
    "dq_propagator_vjp": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp, _vp]),
    "dq_zgemm": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_long, C.c_long,
                           _vp, C.c_long, C.c_long, C.c_double, _vp, C.c_long, C.c_long, C.c_int]),
    "dq_zgemm_ex": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_long, C.c_long,
                              _vp, C.c_long, C.c_long, C.c_double, _vp, C.c_long, C.c_long, C.c_int, C.c_int]),
    "dq_rgemm": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _vp, C.c_long, C.c_long,
                           _vp, C.c_long, C.c_long, C.c_double, _vp, C.c_long, C.c_long, C.c_int, C.c_int]),
    "dq_expm": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "dq_expm_vjp": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int]),
    "dq_profile_enable": (C.c_int, [_vp, C.c_int]),
    "dq_profile_reset": (C.c_int, [_vp]),
    "dq_profile_get": (C.c_int, [_vp, C.POINTER(DqProfile)]),
    "dq_sctx_last_error": (C.c_char_p, []),
    "dq_sctx_create": (C.c_int, [C.POINTER(_vp), C.c_int, C.c_int, C.c_int, C.c_int]),
    "dq_sctx_destroy": (C.c_int, [_vp]),
    "dq_sctx_set_stream": (C.c_int, [_vp, _vp]),
    "dq_sctx_set_terms": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp]),
    "dq_sctx_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "dq_sctx_get_info": (C.c_int, [_vp, C.c_char_p, _dp]),
    "dq_sctx_propagate": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp, _vp, C.c_int]),
    "dq_sctx_energy_grad": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "dq_sctx_apply": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, C.c_int]),
}

_lib = None


def load():
    """Load the HIP library or raise (never substitutes another implementation)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DiffqcError(
            "HIP extension missing: {} (build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C diffquantum_amd/csrc`); there is no CPU fallback".format(LIB_PATH))
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, structured: bool = False):
    if status == DQ_OK:
        return
    lib = load()
    msg = (lib.dq_sctx_last_error() if structured else lib.dq_last_error()).decode("utf-8", "replace")
    if status == DQ_ERR_INVALID:
        raise ValueError(msg)
    if status == DQ_ERR_NO_DEVICE:
        raise DiffqcNoDevice(msg)
    if status == DQ_ERR_NOMEM:
        raise MemoryError(msg)
    if status == DQ_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise DiffqcError("diffqc_hip status {}: {}".format(status, msg))
