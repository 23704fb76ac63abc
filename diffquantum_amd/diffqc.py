"""Host-side mirror of the reference's ``diffqc`` pybind11 module (/root/reference/diffqc.cc:210-228).

Same functions, argument meaning and process-global state as the C++ module:

    set_H(H0, Hs, channels, duration, func_type)             diffqc.cc:43-73
    trotter(psi0, T0, T, per_step, vv) -> list[complex]      diffqc.cc:173-205
    print_test(), complex_test(v), test_eigen(m), __version__ = "dev"   diffqc.cc:27-38,222-227

but the time loop (``dH`` assembly, ``dH.exp()``, ``exp * psi``) runs in libdiffqc_hip.so on the GPU.
The pulse envelope ``f_u`` (diffqc.cc:95-135) is evaluated on the host for the whole time grid
(diffquantum_amd.pulse.f_u_table) and handed to the engine as the u[t][h] table.

When the compiled pybind11 module ``diffquantum_amd._diffqc_native`` (csrc/diffqc_module.cpp: the reference's module
re-built over the C-ABI, pulse model included) is present, ``set_H`` / ``trotter`` / ``trotter_batch`` and the three
test functions go through it; ``DQ_DIFFQC_PURE_PYTHON=1`` keeps everything on this file's Python twin. Both call the
same HIP engine; the extensions ``energy_grad`` / ``fidelity_grad`` always use the Python-side handle.

Differences from the reference, all deliberate: shape errors raise ValueError instead of being
undefined behaviour (diffqc.cc:101-103), and ``trotter_batch`` evolves many states at once.
"""
from __future__ import annotations

import os

import numpy as np

from . import pulse
from .engine import Engine

__version__ = "dev"

native = None
if not os.environ.get("DQ_DIFFQC_PURE_PYTHON"):
    try:
        from . import _diffqc_native as native  # the compiled module (make -C diffquantum_amd/csrc)
    except ImportError:
        native = None

_state = {"engine": None, "channels": None, "duration": None, "func_type": None, "dim": 0, "H": None, "native_set": False}


def print_test():
    if native is not None:
        return native.print_test()
    print("hello")


def complex_test(psi0):
    if native is not None:
        return native.complex_test([complex(z) for z in psi0])
    return [complex(z) for z in psi0]


def test_eigen(v):
    if native is not None:
        return native.test_eigen([[float(x) for x in row] for row in v])
    return [[float(x) for x in row] for row in v]


def _engine():
    """The Python-side engine handle.  When set_H went through the compiled module it BORROWS that module's context (one
    dq_ctx, one device copy of the Hamiltonians and of their sparsity tables); otherwise it owns one."""
    if _state["engine"] is None:
        if _state["H"] is None:
            raise RuntimeError("set_H must be called first")
        d, K = _state["H"]
        if _state["native_set"]:
            _state["engine"] = Engine.borrow(native.ctx_handle(), d, K)
        else:
            raise RuntimeError("set_H must be called first")
    return _state["engine"]


def close():
    """Release the engine context (the compiled module's included); set_H must be called again afterwards."""
    if _state["engine"] is not None:
        _state["engine"].close()
    if native is not None and _state["native_set"]:
        native.close()
    _state.update(engine=None, H=None, native_set=False, dim=0)


def set_H(H0, Hs, channels, duration, func_type):
    H0 = np.asarray(H0, dtype=np.complex128)
    if H0.ndim != 2 or H0.shape[0] != H0.shape[1] or H0.shape[0] == 0:
        raise ValueError("H0 must be a non-empty square matrix")
    d = H0.shape[0]
    Hs = np.asarray(Hs, dtype=np.complex128).reshape(-1, d, d) if len(Hs) else np.zeros((0, d, d), np.complex128)
    if len(channels) != Hs.shape[0]:
        raise ValueError("channels must have one entry per control Hamiltonian ({} != {})".format(len(channels), Hs.shape[0]))
    chans = [[[float(x) for x in row] for row in ch] for ch in channels]
    for ch in chans:
        for row in ch:
            if len(row) < 4:
                raise ValueError("each channel row needs at least 4 entries (unused, omega, w, idx)")
    if _state["engine"] is not None:
        _state["engine"].close()          # (a borrowed handle is only dropped; the module replaces its context below)
    _state.update(engine=None, channels=chans, duration=float(duration), func_type=int(func_type), dim=d, H=(d, Hs.shape[0]),
                  native_set=False)
    if native is not None:
        native.set_H(H0, Hs, chans, float(duration), int(func_type))      # numpy buffers, read in place
        _state["native_set"] = True
    else:
        eng = Engine(d, Hs.shape[0])
        eng.set_hamiltonians(H0, Hs)
        _state["engine"] = eng


def _table(T0, T, per_step, vv):
    if _state["H"] is None:
        raise RuntimeError("set_H must be called before trotter")
    n_steps = pulse.n_steps_diffqc(int(per_step), float(T0), float(T))
    ts, dt = pulse.time_grid(float(T0), float(T), n_steps)
    u = pulse.f_u_table(vv, _state["channels"], _state["duration"], _state["func_type"], ts)
    return u, dt


def trotter(psi0, T0, T, per_step, vv):
    psi = np.asarray(psi0, dtype=np.complex128).reshape(-1)
    if psi.shape[0] != _state["dim"]:
        raise ValueError("psi0 has length {} but set_H was given dimension {}".format(psi.shape[0], _state["dim"]))
    if _state["native_set"]:
        v = np.asarray(vv, dtype=np.float64)
        if v.ndim != 3 or v.shape[0] != 2:
            raise ValueError("vv must have shape [2][n_param][n_basis]")
        return native.trotter(psi.tolist(), float(T0), float(T), int(per_step), v.tolist())   # (the reference's list-in / list-out signature)
    u, dt = _table(T0, T, per_step, vv)
    out = _engine().propagate(u, dt, psi.reshape(-1, 1))
    return [complex(z) for z in out[:, 0]]


def energy_grad(psi0, T0, T, per_step, vv, M):
    """Extension: the channel model made end-to-end differentiable.

    Returns (loss, dL/dvv with the shape of vv, psi_final) for L = sum_b <psi_b(T)|M|psi_b(T)>; psi0 is [dim] or
    [dim][batch]. The time chain and its adjoint run on the GPU, the chain rule through f_u (diffqc.cc:95-135) on the host."""
    if _state["H"] is None:
        raise RuntimeError("set_H must be called before energy_grad")
    n_steps = pulse.n_steps_diffqc(int(per_step), float(T0), float(T))
    ts, dt = pulse.time_grid(float(T0), float(T), n_steps)
    u = pulse.f_u_table(vv, _state["channels"], _state["duration"], _state["func_type"], ts)
    eng = _engine()
    eng.set_observable(np.asarray(M, dtype=np.complex128))
    psi = np.asarray(psi0, dtype=np.complex128)
    loss, grad_u, fin, _ = eng.energy_grad(u, dt, psi.reshape(_state["dim"], -1))
    gvv = pulse.grad_vv_from_grad_u(grad_u, vv, _state["channels"], _state["duration"], _state["func_type"], ts)
    return loss, gvv, fin


def fidelity_grad(psi0, targets, T0, T, per_step, vv):
    """Extension: L = sum_b (1 - |<target_b|psi_b(T)>|^2) and dL/dvv for a batch of (initial, target) pairs."""
    if _state["H"] is None:
        raise RuntimeError("set_H must be called before fidelity_grad")
    n_steps = pulse.n_steps_diffqc(int(per_step), float(T0), float(T))
    ts, dt = pulse.time_grid(float(T0), float(T), n_steps)
    u = pulse.f_u_table(vv, _state["channels"], _state["duration"], _state["func_type"], ts)
    d = _state["dim"]
    loss, grad_u, fin, fid = _engine().fidelity_grad(u, dt, np.asarray(psi0, dtype=np.complex128).reshape(d, -1),
                                                            np.asarray(targets, dtype=np.complex128).reshape(d, -1))
    gvv = pulse.grad_vv_from_grad_u(grad_u, vv, _state["channels"], _state["duration"], _state["func_type"], ts)
    return loss, gvv, fin, fid


def trotter_batch(psi0, T0, T, per_step, vv):
    """Extension: psi0 is [dim][batch] (numpy or CUDA torch complex128); returns the same kind."""
    u, dt = _table(T0, T, per_step, vv)
    return _engine().propagate(u, dt, psi0)
