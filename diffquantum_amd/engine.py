"""Python handle on one ``dq_ctx`` of libdiffqc_hip.so.

Thin plumbing over the C-ABI (include/diffqc_hip.h): argument checking, pointer extraction
from numpy arrays (host) or torch CUDA tensors (device, zero-copy), status -> exception.
The arithmetic of the reference's hot path -- ``dH`` assembly, ``expm`` and the chained
``psi = U psi`` (/root/reference/diffqc.cc:190-200, /root/reference/sim_plain.py:135-150) --
runs entirely inside the HIP library.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def _is_torch(x):
    return type(x).__module__.split(".")[0] == "torch" and hasattr(x, "data_ptr")


def _cx(x, shape=None, name="array", dtype=np.complex128):
    """Return (pointer, keepalive) of a complex buffer of `dtype` with `shape` (host numpy or CUDA torch)."""
    if _is_torch(x):
        import torch
        want = torch.complex128 if dtype == np.complex128 else torch.complex64
        if x.dtype != want:
            raise ValueError("{}: torch tensors must be {}".format(name, want))
        if not x.is_contiguous():
            raise ValueError("{}: torch tensors must be contiguous".format(name))
        if shape is not None and tuple(x.shape) != tuple(shape):
            raise ValueError("{}: expected shape {}, got {}".format(name, tuple(shape), tuple(x.shape)))
        return x.data_ptr(), x
    a = np.ascontiguousarray(x, dtype=dtype)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("{}: expected shape {}, got {}".format(name, tuple(shape), tuple(a.shape)))
    return a.ctypes.data, a


def _f64(x, shape=None, name="array"):
    if _is_torch(x):
        import torch
        if x.dtype != torch.float64 or not x.is_contiguous():
            raise ValueError("{}: torch tensors must be contiguous float64".format(name))
        if shape is not None and tuple(x.shape) != tuple(shape):
            raise ValueError("{}: expected shape {}, got {}".format(name, tuple(shape), tuple(x.shape)))
        return x.data_ptr(), x
    a = np.ascontiguousarray(x, dtype=np.float64)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError("{}: expected shape {}, got {}".format(name, tuple(shape), tuple(a.shape)))
    return a.ctypes.data, a


class Engine:
    """One Hilbert-space dimension + one Hamiltonian set on one GPU.

    dtype "c128" (default; fp64 complex, the reference's `typedef double Scalar`) or "c64" (fp32 complex, the
    reference's commented-out `typedef float Scalar`, diffqc.cc:13-19)."""

    def __init__(self, dim: int, n_controls: int, device: int = 0, dtype: str = "c128"):
        if dtype not in ("c128", "c64"):
            raise ValueError("dtype must be 'c128' or 'c64'")
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        self.dim = int(dim)
        self.n_controls = int(n_controls)
        self.device = int(device)
        self.dtype = dtype
        self.cdtype = np.complex128 if dtype == "c128" else np.complex64
        _lib.check(self._lib.dq_create_ex(C.byref(self._ctx), self.dim, self.n_controls, self.device,
                                          _lib.DQ_C128 if dtype == "c128" else _lib.DQ_C64))

        self._owns = True

    @classmethod
    def borrow(cls, handle: int, dim: int, n_controls: int, device: int = 0, dtype: str = "c128"):
        """Wrap an existing dq_ctx (e.g. the compiled `diffqc` module's, `_diffqc_native.ctx_handle()`) without owning it:
        one context, one device copy of the Hamiltonians; close() then only drops the reference."""
        if not handle:
            raise ValueError("null context handle")
        self = cls.__new__(cls)
        self._lib = _lib.load()
        self._ctx = C.c_void_p(int(handle))
        self.dim, self.n_controls, self.device, self.dtype = int(dim), int(n_controls), int(device), dtype
        self.cdtype = np.complex128 if dtype == "c128" else np.complex64
        self._owns = False
        return self

    def _c(self, x, shape=None, name="array"):
        return _cx(x, shape, name, self.cdtype)

    # -- lifetime -----------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            if getattr(self, "_owns", True):
                self._lib.dq_destroy(self._ctx)
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

    # -- configuration ------------------------------------------------------
    def set_stream(self, stream_handle):
        _lib.check(self._lib.dq_set_stream(self._ctx, C.c_void_p(int(stream_handle) if stream_handle else 0)))

    def use_torch_stream(self):
        import torch
        self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)

    def synchronize(self):
        _lib.check(self._lib.dq_synchronize(self._ctx))

    def set_hamiltonians(self, H0, Hs):
        d, K = self.dim, self.n_controls
        p0, k0 = self._c(H0, (d, d), "H0")
        if K > 0:
            if isinstance(Hs, (list, tuple)):
                Hs = np.stack([np.asarray(h, dtype=self.cdtype) for h in Hs])
            ps, ks = self._c(Hs, (K, d, d), "Hs")
        else:
            ps, ks = 0, None
        _lib.check(self._lib.dq_set_hamiltonians(self._ctx, p0, ps))

    def set_observable(self, M):
        p, k = self._c(M, (self.dim, self.dim), "M")
        _lib.check(self._lib.dq_set_observable(self._ctx, p))

    def set_option(self, name: str, value: float):
        _lib.check(self._lib.dq_set_option(self._ctx, name.encode(), float(value)))

    def info(self, name: str) -> float:
        v = C.c_double()
        _lib.check(self._lib.dq_get_info(self._ctx, name.encode(), C.byref(v)))
        return v.value

    # -- hot path -----------------------------------------------------------
    def _u(self, u):
        K = self.n_controls
        if _is_torch(u):
            if u.dim() != 2 or u.shape[1] != K:
                raise ValueError("u must have shape [n_steps][n_controls={}]".format(K))
            p, keep = _f64(u, None, "u")
            return p, keep, int(u.shape[0])
        a = np.ascontiguousarray(u, dtype=np.float64)
        if a.ndim != 2 or a.shape[1] != K:
            raise ValueError("u must have shape [n_steps][n_controls={}], got {}".format(K, a.shape))
        return a.ctypes.data, a, int(a.shape[0])

    def _psi(self, psi, name="psi"):
        if _is_torch(psi):
            if psi.dim() != 2 or psi.shape[0] != self.dim:
                raise ValueError("{} must have shape [dim={}][batch]".format(name, self.dim))
            p, keep = self._c(psi, None, name)
            return p, keep, int(psi.shape[1])
        a = np.asarray(psi)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        a = np.ascontiguousarray(a, dtype=self.cdtype)
        if a.ndim != 2 or a.shape[0] != self.dim:
            raise ValueError("{} must have shape [dim={}][batch], got {}".format(name, self.dim, a.shape))
        return a.ctypes.data, a, int(a.shape[1])

    def propagate(self, u, dt: float, psi, out=None):
        """psi_out = prod_t exp(-i dt (H0 + sum_k u[t][k] H_k)) psi  for all batch columns."""
        pu, ku, S = self._u(u)
        pp, kp, B = self._psi(psi)
        if out is None:
            if _is_torch(psi):
                import torch
                out = torch.empty_like(psi)
            else:
                out = np.empty((self.dim, B), dtype=self.cdtype)
        po, ko, Bo = self._psi(out, "out")
        if Bo != B or (not _is_torch(out) and ko is not out):
            raise ValueError("out must be a contiguous [dim][batch] buffer of the engine dtype")
        _lib.check(self._lib.dq_propagate(self._ctx, pu, S, float(dt), pp, po, B))
        return out

    def energy(self, psi):
        pp, kp, B = self._psi(psi)
        e = np.empty(B, dtype=np.float64)
        tot = C.c_double()
        _lib.check(self._lib.dq_energy(self._ctx, pp, B, e.ctypes.data, C.addressof(tot)))
        return e, tot.value

    def energy_grad(self, u, dt: float, psi0, want_final=True, want_energies=True, grad_out=None):
        """Forward chain + adjoint sweep.

        Returns (loss, grad_u[n_steps][n_controls], psi_final or None, energies[batch] or None).
        """
        pu, ku, S = self._u(u)
        pp, kp, B = self._psi(psi0)
        K = self.n_controls
        if grad_out is None:
            grad = np.zeros((S, K), dtype=np.float64)
            pg = grad.ctypes.data
        else:
            grad = grad_out
            pg, _kg = _f64(grad, (S, K), "grad_out")
        loss = C.c_double()
        fin = None
        pf = 0
        if want_final:
            if _is_torch(psi0):
                import torch
                fin = torch.empty_like(psi0)
                pf = fin.data_ptr()
            else:
                fin = np.empty((self.dim, B), dtype=self.cdtype)
                pf = fin.ctypes.data
        en = None
        pe = 0
        if want_energies:
            en = np.empty(B, dtype=np.float64)
            pe = en.ctypes.data
        _lib.check(self._lib.dq_energy_grad(self._ctx, pu, S, float(dt), pp, B, C.addressof(loss), pg, pf, pe))
        return loss.value, grad, fin, en

    def fidelity_grad(self, u, dt: float, psi0, targets, want_final=True):
        """Forward chain + adjoint sweep of L = sum_b (1 - |<target_b|psi_b(T)>|^2).

        Returns (loss, grad_u[n_steps][n_controls], psi_final or None, fidelities[batch])."""
        pu, ku, S = self._u(u)
        pp, kp, B = self._psi(psi0)
        pt, kt, Bt = self._psi(targets, "targets")
        if Bt != B:
            raise ValueError("targets must have the same batch width as psi0")
        grad = np.zeros((S, self.n_controls), dtype=np.float64)
        loss = C.c_double()
        fid = np.empty(B, dtype=np.float64)
        fin, pf = None, 0
        if want_final:
            if _is_torch(psi0):
                import torch
                fin = torch.empty_like(psi0)
                pf = fin.data_ptr()
            else:
                fin = np.empty((self.dim, B), dtype=self.cdtype)
                pf = fin.ctypes.data
        _lib.check(self._lib.dq_fidelity_grad(self._ctx, pu, S, float(dt), pp, pt, B, C.addressof(loss),
                                              grad.ctypes.data, pf, fid.ctypes.data))
        return loss.value, grad, fin, fid

    def mc_shift_grad(self, u_a, dt_a: float, u_b, dt_b: float, psi0, r: float = 0.5, want_phi=False):
        """The reference's Monte-Carlo estimator core (sim_plain.py:186-220) in one call: evolve psi0 through chain a
        (0 -> s), apply the 2K gates (I +- i r H_k)/sqrt(1 + r^2), evolve the shifted batch through chain b (s -> T),
        measure M. Returns ps[batch][K] = (1 + r^2)/(2r) (<M>_- - <M>_+) (and phi = the state at s, if asked)."""
        pa, ka, Sa = self._u(u_a)
        pb, kb, Sb = self._u(u_b)
        pp, kp, B = self._psi(psi0)
        ps = np.zeros((B, self.n_controls), dtype=np.float64)
        phi, pphi = None, 0
        if want_phi:
            phi = np.empty((self.dim, B), dtype=self.cdtype)
            pphi = phi.ctypes.data
        _lib.check(self._lib.dq_mc_shift_grad(self._ctx, pa, Sa, float(dt_a), pb, Sb, float(dt_b), pp, B, float(r),
                                              ps.ctypes.data, pphi))
        return (ps, phi) if want_phi else ps

    # -- time-sliced operator form (phases of energy_grad; see diffquantum_amd/distributed.py) ---------
    def _mat_like(self, like):
        """An uninitialised [dim][dim] complex matrix of the same kind as `like` (CUDA torch tensor or numpy)."""
        if _is_torch(like):
            import torch
            return torch.empty((self.dim, self.dim), dtype=like.dtype if like.is_complex() else
                               (torch.complex128 if self.dtype == "c128" else torch.complex64), device=like.device)
        return np.empty((self.dim, self.dim), dtype=self.cdtype)

    # ---- column-sharded form of the wide-batch energy gradient (distributed.column_sharded_energy_grad) ----
    def gram(self, psi0):
        """R = Psi0 Psi0^H ([dim][dim]) of these states: what the loss of a batch depends on (include/diffqc_hip.h, dq_gram).
        A matrix of the kind of psi0 (torch tensor on its device, or numpy)."""
        pp, kp, B = self._psi(psi0)
        R = self._mat_like(psi0)
        pr, kr = self._c(R, (self.dim, self.dim), "R")
        _lib.check(self._lib.dq_gram(self._ctx, pp, B, pr))
        return R

    def energy_grad_columns(self, u, dt: float, R, col0: int, n_cols: int, P=None):
        """The share of (loss, dL/du[n_steps][n_controls]) that the columns [col0, col0 + n_cols) of (P R, 2 M P) carry, R the Gram
        matrix of the WHOLE batch (dq_energy_grad_columns); summed over a partition of [0, dim) it is energy_grad of that batch.
        P: the chain's product U_{S-1} ... U_0 when it was made elsewhere (time slices); None: the call builds it."""
        pu, ku, S = self._u(u)
        pr, kr = self._c(R, (self.dim, self.dim), "R")
        pp = 0
        if P is not None:
            pp, kp = self._c(P, (self.dim, self.dim), "P")
        grad = np.zeros((S, self.n_controls), dtype=np.float64)
        loss = C.c_double()
        _lib.check(self._lib.dq_energy_grad_columns(self._ctx, pu, S, float(dt), pr, pp, int(col0), int(n_cols), C.addressof(loss),
                                                    grad.ctypes.data))
        return loss.value, grad

    def slice_forward(self, u, dt: float, like=None):
        """P = U_{n-1} ... U_0 of the slice's own pulse rows u[n][K] (identity for n = 0); the propagator
        intermediates stay cached for slice_backward. Returns a [dim][dim] matrix of the kind of `like`."""
        pu, ku, n = self._u(u)
        P = self._mat_like(like)
        pp, kp = self._c(P, (self.dim, self.dim), "P")
        _lib.check(self._lib.dq_slice_forward(self._ctx, pu, n, float(dt), pp))
        return P

    def slice_terminal(self, Q, psi0, targets=None, want_final=False, want_energies=False):
        """Psi_S = Q Psi_0 for this rank's columns; returns (loss, W_S, psi_final or None, energies or None) with
        W_S = Lambda_S Psi_S^H (anti-Hermitian part if info('slice_skew'))."""
        pq, kq = self._c(Q, (self.dim, self.dim), "Q")
        pp, kp, B = self._psi(psi0)
        pt = 0
        if targets is not None:
            pt, kt, Bt = self._psi(targets, "targets")
            if Bt != B:
                raise ValueError("targets must have the same batch width as psi0")
        W = self._mat_like(Q)
        pw, kw = self._c(W, (self.dim, self.dim), "W")
        loss = C.c_double()
        fin, pf = None, 0
        if want_final:
            if _is_torch(psi0):
                import torch
                fin = torch.empty_like(psi0)
                pf = fin.data_ptr()
            else:
                fin = np.empty((self.dim, B), dtype=self.cdtype)
                pf = fin.ctypes.data
        en, pe = None, 0
        if want_energies:
            en = np.empty(B, dtype=np.float64)
            pe = en.ctypes.data
        _lib.check(self._lib.dq_slice_terminal(self._ctx, pq, pp, B, pt, C.addressof(loss), pe, pf, pw))
        return loss.value, W, fin, en

    def slice_backward(self, u, dt: float, W_end, want_w_start=False):
        """Operator-form backward sweep over the slice's own steps from W at the slice's end; returns
        grad_u[n][K] (numpy; rows of dL/du summed over the whole batch) and, if asked, W at the slice's start."""
        pu, ku, n = self._u(u)
        pw, kw = self._c(W_end, (self.dim, self.dim), "W_end")
        grad = np.zeros((n, self.n_controls), dtype=np.float64)
        Ws, ps = None, 0
        if want_w_start:
            Ws = self._mat_like(W_end)
            ps, ks = self._c(Ws, (self.dim, self.dim), "W_start")
        _lib.check(self._lib.dq_slice_backward(self._ctx, pu, n, float(dt), pw, grad.ctypes.data, ps))
        return (grad, Ws) if want_w_start else grad

    # -- explicit propagators in and out (SURVEY.md 8-F2 in its literal form; see distributed.propagator_sharded_energy_grad) --
    def build_propagators(self, u, dt: float, like=None):
        """U[t] = exp(-i dt (H0 + sum_k u[t][k] H_k)) for the rows of u: an [n][dim][dim] array of the kind of `like`
        (CUDA torch tensor -> the propagators stay on the device; otherwise numpy)."""
        pu, ku, n = self._u(u)
        if _is_torch(like):
            import torch
            U = torch.empty((n, self.dim, self.dim), dtype=torch.complex128 if self.dtype == "c128" else torch.complex64,
                            device=like.device)
            pU = U.data_ptr()
        else:
            U = np.empty((n, self.dim, self.dim), dtype=self.cdtype)
            pU = U.ctypes.data
        _lib.check(self._lib.dq_build_propagators(self._ctx, pu, n, float(dt), pU))
        return U

    def propagator_vjp(self, u, dt: float, G):
        """grad_u[t][k] = Re <G[t], dU_t/du_k> for cotangents G [n][dim][dim] of the propagators of the rows of u."""
        pu, ku, n = self._u(u)
        pG, kG = self._c(G, (n, self.dim, self.dim), "G")
        grad = np.zeros((n, self.n_controls), dtype=np.float64)
        _lib.check(self._lib.dq_propagator_vjp(self._ctx, pu, n, float(dt), pG, grad.ctypes.data))
        return grad

    def gemm(self, A, B, opA=0, opB=0, alpha=1.0, out=None):
        """alpha op(A) @ op(B) for 2-D complex arrays of either kind (CUDA torch tensors stay on the device); op 1 =
        conjugate transpose. Shapes are taken from the arrays."""
        ar, ac = int(A.shape[0]), int(A.shape[1])
        br, bc = int(B.shape[0]), int(B.shape[1])
        M, K = (ac, ar) if opA == 1 else (ar, ac)
        K2, N = (bc, br) if opB == 1 else (br, bc)
        if K != K2:
            raise ValueError("inner dimensions differ: {} vs {}".format(K, K2))
        pa, ka = self._c(A, (ar, ac), "A")
        pb, kb = self._c(B, (br, bc), "B")
        if out is None:
            if _is_torch(A):
                import torch
                out = torch.empty((M, N), dtype=A.dtype, device=A.device)
            else:
                out = np.empty((M, N), dtype=self.cdtype)
        po, ko = self._c(out, (M, N), "out")
        if not _is_torch(out) and ko is not out:
            raise ValueError("out must be a contiguous buffer of the engine dtype")
        _lib.check(self._lib.dq_zgemm_ex(self._ctx, int(opA), int(opB), M, N, K, float(alpha), pa, ac, ar * ac, pb, bc, br * bc,
                                         0.0, po, N, M * N, 1, 0))
        return out

    def matmul(self, A, B, opA=0, opB=0, herm=0, out=None):
        """op(A) @ op(B) for [dim][dim] matrices of either kind (CUDA torch tensors stay on the device);
        op 1 = conjugate transpose; herm = +1/-1: (anti-)Hermitian output form."""
        d = self.dim
        pa, ka = self._c(A, (d, d), "A")
        pb, kb = self._c(B, (d, d), "B")
        if out is None:
            out = self._mat_like(A)
        po, ko = self._c(out, (d, d), "out")
        if not _is_torch(out) and ko is not out:
            raise ValueError("out must be a contiguous [dim][dim] buffer of the engine dtype")
        _lib.check(self._lib.dq_zgemm_ex(self._ctx, int(opA), int(opB), d, d, d, 1.0, pa, d, d * d, pb, d, d * d, 0.0,
                                         po, d, d * d, 1, int(herm)))
        return out

    def apply(self, Q, psi, out=None):
        """Q @ psi for a [dim][dim] operator and a [dim][batch] state batch of either kind (device tensors in place)."""
        d = self.dim
        pq, kq = self._c(Q, (d, d), "Q")
        pp, kp, B = self._psi(psi)
        if out is None:
            if _is_torch(psi):
                import torch
                out = torch.empty_like(psi)
            else:
                out = np.empty((d, B), dtype=self.cdtype)
        po, ko, Bo = self._psi(out, "out")
        if Bo != B or (not _is_torch(out) and ko is not out):
            raise ValueError("out must be a contiguous [dim][batch] buffer of the engine dtype")
        _lib.check(self._lib.dq_zgemm_ex(self._ctx, 0, 0, d, B, d, 1.0, pq, d, d * d, pp, B, d * B, 0.0, po, B, d * B, 1, 0))
        return out

    # -- building blocks ----------------------------------------------------
    def zgemm(self, A, B, opA=0, opB=0, alpha=1.0, beta=0.0, C_in=None, herm=0):
        """alpha*op(A)@op(B) + beta*C on the GPU kernel; A, B: 2-D or batched 3-D complex128 (host).
        herm=+1/-1: the product is known to be Hermitian / anti-Hermitian (upper tiles + mirror)."""
        A = np.ascontiguousarray(A, dtype=self.cdtype)
        B = np.ascontiguousarray(B, dtype=self.cdtype)
        batched = A.ndim == 3
        if not batched:
            A, B = A[None], B[None]
        nb = A.shape[0]
        if B.shape[0] != nb:
            raise ValueError("batch mismatch")
        M, K = (A.shape[2], A.shape[1]) if opA == 1 else (A.shape[1], A.shape[2])  # opA 2/3/4: as stored + symmetry hint
        K2, N = (B.shape[2], B.shape[1]) if opB else (B.shape[1], B.shape[2])
        if K != K2:
            raise ValueError("inner dimensions differ: {} vs {}".format(K, K2))
        if C_in is None:
            Cm = np.zeros((nb, M, N), dtype=self.cdtype)
        else:
            Cm = np.array(C_in, dtype=self.cdtype, order="C").reshape(nb, M, N)
        _lib.check(self._lib.dq_zgemm_ex(self._ctx, int(opA), int(opB), M, N, K, float(alpha), A.ctypes.data,
                                         A.shape[2], A.shape[1] * A.shape[2], B.ctypes.data, B.shape[2],
                                         B.shape[1] * B.shape[2], float(beta), Cm.ctypes.data, N, M * N, nb, int(herm)))
        return Cm if batched else Cm[0]

    def rgemm(self, A, B, opA=0, opB=0, alpha=1.0, beta=0.0, C_in=None, sym=False):
        """Real GEMM alpha*op(A)@op(B) + beta*C on the GPU kernel (op: 0 = N, 1 = transpose); A, B: 2-D or batched
        3-D real arrays (host). sym=True/1: symmetric-output form (commuting symmetric operands); sym=2: the
        two-product form alpha*(A@B + B@A) + beta*C for symmetric A, B."""
        rdtype = np.float64 if self.dtype == "c128" else np.float32
        A = np.ascontiguousarray(A, dtype=rdtype)
        B = np.ascontiguousarray(B, dtype=rdtype)
        batched = A.ndim == 3
        if not batched:
            A, B = A[None], B[None]
        nb = A.shape[0]
        if B.shape[0] != nb:
            raise ValueError("batch mismatch")
        ta, tb = (1, 0) if sym else (int(opA), int(opB))
        M, K = (A.shape[2], A.shape[1]) if ta else (A.shape[1], A.shape[2])
        K2, N = (B.shape[2], B.shape[1]) if tb else (B.shape[1], B.shape[2])
        if K != K2:
            raise ValueError("inner dimensions differ: {} vs {}".format(K, K2))
        if C_in is None:
            Cm = np.zeros((nb, M, N), dtype=rdtype)
        else:
            Cm = np.array(C_in, dtype=rdtype, order="C").reshape(nb, M, N)
        _lib.check(self._lib.dq_rgemm(self._ctx, ta, tb, M, N, K, float(alpha), A.ctypes.data, A.shape[2],
                                      A.shape[1] * A.shape[2], B.ctypes.data, B.shape[2], B.shape[1] * B.shape[2],
                                      float(beta), Cm.ctypes.data, N, M * N, nb, int(sym)))
        return Cm if batched else Cm[0]

    def expm(self, A):
        A = np.ascontiguousarray(A, dtype=self.cdtype)
        batched = A.ndim == 3
        Ab = A if batched else A[None]
        if Ab.shape[1:] != (self.dim, self.dim):
            raise ValueError("expm: matrices must be dim x dim")
        U = np.empty_like(Ab)
        _lib.check(self._lib.dq_expm(self._ctx, Ab.ctypes.data, U.ctypes.data, Ab.shape[0]))
        return U if batched else U[0]

    def expm_vjp(self, A, Ubar):
        A = np.ascontiguousarray(A, dtype=self.cdtype)
        Ubar = np.ascontiguousarray(Ubar, dtype=self.cdtype)
        batched = A.ndim == 3
        Ab = A if batched else A[None]
        Ub = Ubar if batched else Ubar[None]
        if Ab.shape[1:] != (self.dim, self.dim) or Ub.shape != Ab.shape:
            raise ValueError("expm_vjp: matrices must be dim x dim and shapes must match")
        out = np.empty_like(Ab)
        _lib.check(self._lib.dq_expm_vjp(self._ctx, Ab.ctypes.data, Ub.ctypes.data, out.ctypes.data, Ab.shape[0]))
        return out if batched else out[0]

    # -- profiling ----------------------------------------------------------
    def profile_enable(self, on=True):
        _lib.check(self._lib.dq_profile_enable(self._ctx, 1 if on else 0))

    def profile_reset(self):
        _lib.check(self._lib.dq_profile_reset(self._ctx))

    def profile(self):
        p = _lib.DqProfile()
        _lib.check(self._lib.dq_profile_get(self._ctx, C.byref(p)))
        return {
            name: {"launches": int(p.launches[i]), "ms": float(p.ms[i]), "flops": float(p.flops[i]),
                   "bytes": float(p.bytes[i])}
            for i, name in enumerate(_lib.PROF_CLASSES)
        }


def device_count() -> int:
    n = C.c_int()
    _lib.check(_lib.load().dq_device_count(C.byref(n)))
    return n.value
