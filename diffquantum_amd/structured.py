"""Python handle on the structured (matrix-free) engine of libdiffqc_hip.so (``dq_sctx_*`` in
include/diffqc_hip.h): Hamiltonians given as real-weighted sums of Pauli strings, for qubit counts where the dense
matrices of the reference (`diffqc.set_H` diffqc.cc:43-67, `SimulatorPlain.trotter` sim_plain.py:126-131) cannot
exist.  Same step semantics as the dense engine: psi <- exp(-i dt (H0 + sum_k u_k H_k)) psi.

Qubit order follows the reference's kron construction (demo_maxcut.py:49-56, sim_plain.multi_kron): character q of
a Pauli string acts on the q-th tensor factor, i.e. on bit (n-1-q) of the basis-state index.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .engine import _cx, _f64, _is_torch

DRIFT, OBSERVABLE = -1, -2


def pauli_masks(string: str):
    """(xmask, zmask) of a Pauli string such as "XIZY" (leftmost character = most significant index bit)."""
    n = len(string)
    x = z = 0
    for q, ch in enumerate(string.upper()):
        bit = 1 << (n - 1 - q)
        if ch == "X":
            x |= bit
        elif ch == "Z":
            z |= bit
        elif ch == "Y":
            x |= bit
            z |= bit
        elif ch != "I":
            raise ValueError("Pauli strings may contain only I, X, Y, Z (got {!r})".format(ch))
    return x, z


def pauli_dense(string: str):
    """Dense matrix of a Pauli string (for parity tests at small n)."""
    P = {"I": np.eye(2), "X": np.array([[0, 1], [1, 0]]), "Y": np.array([[0, -1j], [1j, 0]]),
         "Z": np.array([[1, 0], [0, -1]])}
    out = np.array([[1.0 + 0j]])
    for ch in string.upper():
        out = np.kron(out, P[ch])
    return out


def terms_to_dense(terms, n_qubits: int, n_controls: int):
    """(H0, Hs[K], M) dense complex128 matrices of a term list [(owner, string, weight), ...]."""
    d = 1 << n_qubits
    H0 = np.zeros((d, d), complex)
    Hs = np.zeros((n_controls, d, d), complex)
    M = np.zeros((d, d), complex)
    for owner, string, w in terms:
        mat = w * pauli_dense(string)
        if owner == DRIFT:
            H0 += mat
        elif owner == OBSERVABLE:
            M += mat
        else:
            Hs[owner] += mat
    return H0, Hs, M


class StructuredEngine:
    def __init__(self, n_qubits: int, n_controls: int, device: int = 0, dtype: str = "c64"):
        if dtype not in ("c128", "c64"):
            raise ValueError("dtype must be 'c128' or 'c64'")
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        self.n_qubits, self.n_controls, self.device, self.dtype = int(n_qubits), int(n_controls), int(device), dtype
        self.dim = 1 << self.n_qubits
        self.cdtype = np.complex128 if dtype == "c128" else np.complex64
        self._ck(self._lib.dq_sctx_create(C.byref(self._ctx), self.n_qubits, self.n_controls, self.device,
                                          _lib.DQ_C128 if dtype == "c128" else _lib.DQ_C64))

    @staticmethod
    def _ck(status):
        _lib.check(status, structured=True)

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self._lib.dq_sctx_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def use_torch_stream(self):
        import torch
        self._ck(self._lib.dq_sctx_set_stream(self._ctx, C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))

    def set_terms(self, terms):
        """terms: iterable of (owner, pauli_string, weight); owner DRIFT (-1), OBSERVABLE (-2) or a control index."""
        terms = list(terms)
        owner = np.empty(len(terms), dtype=np.int32)
        xm = np.empty(len(terms), dtype=np.uint32)
        zm = np.empty(len(terms), dtype=np.uint32)
        w = np.empty(len(terms), dtype=np.float64)
        for i, (o, s, wt) in enumerate(terms):
            if len(s) != self.n_qubits:
                raise ValueError("Pauli string {!r} must have one character per qubit ({})".format(s, self.n_qubits))
            owner[i] = o
            xm[i], zm[i] = pauli_masks(s)
            w[i] = wt
        self._ck(self._lib.dq_sctx_set_terms(self._ctx, len(terms), owner.ctypes.data, xm.ctypes.data, zm.ctypes.data,
                                             w.ctypes.data))

    def set_option(self, name, value):
        """dq_sctx_set_option (include/diffqc_hip.h): "tol", "force_degree", "checkpoint_bytes", "plan_steps", "two_level", "profile",
        and the kernel-form switches "dma", "dma_workgroups", "chain", "reverse_chain", "step_handover", "reverse_keep_last", "fuse_overlaps", "interleave_groups" (the last
        one before set_terms).  info(): "degree", "theta", "groups", "checkpoint_stride", "two_level", "passes", "dma_passes", ..."""
        self._ck(self._lib.dq_sctx_set_option(self._ctx, name.encode(), float(value)))

    def info(self, name):
        v = C.c_double()
        self._ck(self._lib.dq_sctx_get_info(self._ctx, name.encode(), C.byref(v)))
        return v.value

    def _u(self, u):
        if _is_torch(u):
            if u.dim() != 2 or u.shape[1] != self.n_controls:
                raise ValueError("u must have shape [n_steps][n_controls]")
            p, k = _f64(u, None, "u")
            return p, k, int(u.shape[0])
        a = np.ascontiguousarray(u, dtype=np.float64)
        if a.ndim != 2 or a.shape[1] != self.n_controls:
            raise ValueError("u must have shape [n_steps][n_controls={}], got {}".format(self.n_controls, a.shape))
        return a.ctypes.data, a, int(a.shape[0])

    def _psi(self, psi, name="psi"):
        if _is_torch(psi):
            if psi.dim() != 2 or psi.shape[0] != self.dim:
                raise ValueError("{} must have shape [2^n][batch]".format(name))
            p, k = _cx(psi, None, name, self.cdtype)
            return p, k, int(psi.shape[1])
        a = np.asarray(psi)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        a = np.ascontiguousarray(a, dtype=self.cdtype)
        if a.ndim != 2 or a.shape[0] != self.dim:
            raise ValueError("{} must have shape [2^n={}][batch], got {}".format(name, self.dim, a.shape))
        return a.ctypes.data, a, int(a.shape[1])

    def _like(self, psi, B):
        if _is_torch(psi):
            import torch
            return torch.empty_like(psi)
        return np.empty((self.dim, B), dtype=self.cdtype)

    def apply(self, psi, u_row=None, observable=False):
        """H(u_row) psi, or M psi: one matrix-free application."""
        pp, kp, B = self._psi(psi)
        out = self._like(psi, B)
        po, ko, _ = self._psi(out, "out")
        if observable:
            self._ck(self._lib.dq_sctx_apply(self._ctx, 1, 0, pp, po, B))
        else:
            ur = np.ascontiguousarray(u_row if u_row is not None else np.zeros(self.n_controls), dtype=np.float64)
            if ur.shape != (self.n_controls,):
                raise ValueError("u_row must have n_controls entries")
            self._ck(self._lib.dq_sctx_apply(self._ctx, 0, ur.ctypes.data, pp, po, B))
        return out

    def propagate(self, u, dt, psi):
        pu, ku, S = self._u(u)
        pp, kp, B = self._psi(psi)
        out = self._like(psi, B)
        po, ko, _ = self._psi(out, "out")
        self._ck(self._lib.dq_sctx_propagate(self._ctx, pu, S, float(dt), pp, po, B))
        return out

    def energy_grad(self, u, dt, psi0, want_final=True):
        pu, ku, S = self._u(u)
        pp, kp, B = self._psi(psi0)
        grad = np.zeros((S, self.n_controls), dtype=np.float64)
        loss = C.c_double()
        en = np.empty(B, dtype=np.float64)
        fin = self._like(psi0, B) if want_final else None
        pf = self._psi(fin, "fin")[0] if want_final else 0
        self._ck(self._lib.dq_sctx_energy_grad(self._ctx, pu, S, float(dt), pp, B, C.addressof(loss), grad.ctypes.data,
                                               pf, en.ctypes.data))
        return loss.value, grad, fin, en


def ising_x_terms(n: int, seed: int = 0):
    """The D2 synthetic problem as strings: H0 = sum J_ij Z_i Z_j + sum h_i Z_i, H_k = X_k, M = H0."""
    rng = np.random.default_rng(seed)
    J = rng.uniform(-1, 1, (n, n))
    h = rng.uniform(-1, 1, n)
    terms = []

    def s(pos):
        return "".join(pos.get(q, "I") for q in range(n))

    for i in range(n):
        terms.append((DRIFT, s({i: "Z"}), float(h[i])))
        terms.append((OBSERVABLE, s({i: "Z"}), float(h[i])))
        for j in range(i + 1, n):
            terms.append((DRIFT, s({i: "Z", j: "Z"}), float(J[i, j])))
            terms.append((OBSERVABLE, s({i: "Z", j: "Z"}), float(J[i, j])))
    for k in range(n):
        terms.append((k, s({k: "X"}), 1.0))
    return terms
