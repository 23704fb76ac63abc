"""Test infrastructure: a numpy stand-in for the three phase calls of the time-sliced operator form
(Engine.slice_forward / slice_terminal / slice_backward / matmul), written with the ORACLE's arithmetic
(scipy expm + expm_frechet on the reference's step generator, sim_plain.py:137-146).  It lets the CPU suite drive
diffquantum_amd.distributed's rank program through real torch.distributed collectives (gloo) without a GPU.
Never imported by the product."""
import numpy as np
import scipy.linalg as sla

from oracle import trotter_oracle as orc


class NumpySliceEngine:
    def __init__(self, H0, Hs, M, skew=True):
        self.H0 = np.asarray(H0, dtype=np.complex128)
        self.Hs = [np.asarray(h, dtype=np.complex128) for h in Hs]
        self.M = np.asarray(M, dtype=np.complex128)
        self.dim = self.H0.shape[0]
        self.n_controls = len(self.Hs)
        self.skew = skew

    def info(self, name):
        if name == "slice_skew":
            return 1.0 if self.skew else 0.0
        raise KeyError(name)

    def slice_forward(self, u, dt, like=None):
        P = np.eye(self.dim, dtype=np.complex128)
        for row in np.asarray(u, dtype=np.float64).reshape(-1, self.n_controls):
            P = sla.expm(orc.step_generator(self.H0, self.Hs, row, dt)) @ P
        return P

    def matmul(self, A, B, opA=0, opB=0, herm=0, out=None):
        A = A.conj().T if opA else A
        B = B.conj().T if opB else B
        return A @ B

    def slice_terminal(self, Q, psi0, targets=None, want_final=False, want_energies=False):
        psi = Q @ psi0
        lam = 2.0 * (self.M @ psi)
        en = np.real(np.sum(psi.conj() * (self.M @ psi), axis=0))
        W = lam @ psi.conj().T
        if self.skew:
            W = 0.5 * (W - W.conj().T)
        return float(en.sum()), W, (psi if want_final else None), (en if want_energies else None)

    def slice_backward(self, u, dt, W_end, want_w_start=False):
        u = np.asarray(u, dtype=np.float64).reshape(-1, self.n_controls)
        W = np.array(W_end)
        grad = np.zeros_like(u)
        for t in range(u.shape[0] - 1, -1, -1):
            A = orc.step_generator(self.H0, self.Hs, u[t], dt)
            U = sla.expm(A)
            G = W @ U
            for k, Hk in enumerate(self.Hs):
                dU = sla.expm_frechet(A, -1j * dt * Hk, compute_expm=False)
                grad[t, k] = np.real(np.vdot(G, dU))  # Re tr(G^H dU)
            W = U.conj().T @ G
        return (grad, W) if want_w_start else grad

    # ---- explicit propagators in and out (Engine.build_propagators / propagator_vjp / gemm / energy) ----
    def build_propagators(self, u, dt, like=None):
        u = np.asarray(u, dtype=np.float64).reshape(-1, self.n_controls)
        out = np.empty((u.shape[0], self.dim, self.dim), dtype=np.complex128)
        for t, row in enumerate(u):
            out[t] = sla.expm(orc.step_generator(self.H0, self.Hs, row, dt))
        return out

    def propagator_vjp(self, u, dt, G):
        u = np.asarray(u, dtype=np.float64).reshape(-1, self.n_controls)
        grad = np.zeros_like(u)
        for t, row in enumerate(u):
            A = orc.step_generator(self.H0, self.Hs, row, dt)
            for k, Hk in enumerate(self.Hs):
                grad[t, k] = np.real(np.vdot(G[t], sla.expm_frechet(A, -1j * dt * Hk, compute_expm=False)))
        return grad

    def gemm(self, A, B, opA=0, opB=0, alpha=1.0, out=None):
        res = alpha * ((A.conj().T if opA else A) @ (B.conj().T if opB else B))
        if out is not None:
            out[...] = res
            return out
        return res

    def energy(self, psi):
        en = np.real(np.sum(psi.conj() * (self.M @ psi), axis=0))
        return en, float(en.sum())

    # ---- column-sharded wide-batch gradient (Engine.gram / energy_grad_columns), with the oracle's arithmetic ----
    def gram(self, psi0):
        psi0 = np.asarray(psi0, dtype=np.complex128)
        return psi0 @ psi0.conj().T

    def energy_grad_columns(self, u, dt, R, col0, n_cols, P=None):
        """The share of (L, dL/du) of the columns [col0, col0 + n_cols) of (P R, 2 M P): L = tr(M P R P^H) = 1/2 Re sum_i <lam_i, y_i>,
        dL/du_k[t] = Re sum_i <lam_i(t+1)| dU_t |y_i(t)> with y_i(t) = U_t^H y_i(t+1), lam_i(t) = U_t^H lam_i(t+1)."""
        u = np.asarray(u, dtype=np.float64).reshape(-1, self.n_controls)
        S = u.shape[0]
        gens = [orc.step_generator(self.H0, self.Hs, row, dt) for row in u]
        Us = [sla.expm(A) for A in gens]
        if P is None:
            P = np.eye(self.dim, dtype=np.complex128)
            for U in Us:
                P = U @ P
        sl = slice(col0, col0 + n_cols)
        Y = (P @ np.asarray(R, dtype=np.complex128))[:, sl]
        Lam = (2.0 * self.M @ P)[:, sl]
        loss = 0.5 * float(np.real(np.sum(Lam.conj() * Y)))
        grad = np.zeros((S, self.n_controls))
        for t in range(S - 1, -1, -1):
            Yt = Us[t].conj().T @ Y
            for k, Hk in enumerate(self.Hs):
                dU = sla.expm_frechet(gens[t], -1j * dt * Hk, compute_expm=False)
                grad[t, k] = np.real(np.sum(Lam.conj() * (dU @ Yt)))
            Y = Yt
            Lam = Us[t].conj().T @ Lam
        return loss, grad
