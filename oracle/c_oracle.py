"""ORACLE -- TEST INFRASTRUCTURE ONLY.  ctypes access to oracle/_build/libdiffqc_oracle.so
(the plain-C restatement of /root/reference/diffqc.cc, see oracle/diffqc_oracle.c)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libdiffqc_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
        _lib.dqo_expit.restype = C.c_double
        _lib.dqo_expit.argtypes = [C.c_double]
        _lib.dqo_bspline.restype = C.c_double
        _lib.dqo_bspline.argtypes = [C.c_int, C.c_int, C.c_double]
        _lib.dqo_legendre.restype = C.c_double
        _lib.dqo_legendre.argtypes = [C.c_int, C.c_double]
        _lib.dqo_expm.restype = C.c_int
        _lib.dqo_expm.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        _lib.dqo_propagate.restype = C.c_int
        _lib.dqo_propagate.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_double,
                                       C.c_void_p, C.c_int]
        _lib.dqo_trotter.restype = C.c_int
        _lib.dqo_trotter.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double,
                                     C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_double, C.c_double,
                                     C.c_int]
    return _lib


def expm(A):
    A = np.ascontiguousarray(A, dtype=np.complex128)
    U = np.empty_like(A)
    rc = lib().dqo_expm(A.shape[0], A.ctypes.data, U.ctypes.data)
    assert rc == 0, rc
    return U


def propagate(H0, Hs, u, dt, psi):
    H0 = np.ascontiguousarray(H0, dtype=np.complex128)
    d = H0.shape[0]
    Hs = np.ascontiguousarray(Hs, dtype=np.complex128).reshape(-1, d, d)
    u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1, Hs.shape[0])
    out = np.array(psi, dtype=np.complex128).reshape(d, -1).copy()
    rc = lib().dqo_propagate(d, Hs.shape[0], H0.ctypes.data, Hs.ctypes.data, u.ctypes.data, u.shape[0], float(dt),
                             out.ctypes.data, out.shape[1])
    assert rc == 0, rc
    return out


def trotter(H0, Hs, channels, duration, func_type, psi0, T0, T, per_step, vv):
    H0 = np.ascontiguousarray(H0, dtype=np.complex128)
    d = H0.shape[0]
    Hs = np.ascontiguousarray(Hs, dtype=np.complex128).reshape(-1, d, d)
    off = np.zeros(len(channels) + 1, dtype=np.int32)
    rows = []
    for h, ch in enumerate(channels):
        off[h + 1] = off[h] + len(ch)
        rows += [list(r[:4]) for r in ch]
    rows = np.ascontiguousarray(rows, dtype=np.float64).reshape(-1, 4)
    vv = np.ascontiguousarray(vv, dtype=np.float64)
    psi = np.array(psi0, dtype=np.complex128).reshape(-1).copy()
    rc = lib().dqo_trotter(d, Hs.shape[0], H0.ctypes.data, Hs.ctypes.data, off.ctypes.data, rows.ctypes.data,
                           float(duration), int(func_type), vv.ctypes.data, vv.shape[1], vv.shape[2],
                           psi.ctypes.data, float(T0), float(T), int(per_step))
    assert rc == 0, rc
    return psi
