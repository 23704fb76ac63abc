"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product.

CPU restatement (numpy + scipy) of the reference's Schroedinger-propagator hot path and of the
pulse model that feeds it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import this module, and only as the checker / reported baseline.  The
product path (``diffquantum_amd``) never imports it and has no CPU fallback.

Pinning status
  * The ``sim_plain.py`` path (basis functions, ``generate_u``, ``trotter``,
    ``compute_energy_grad_MC``, the demo's training trace) is PINNED: ``tests/test_oracle_golden.py``
    checks this file against golden vectors captured by importing /root/reference/sim_plain.py in
    place (``tests/golden/make_goldens.py``; fixtures ``tests/golden/*.npz``).
  * The ``diffqc.cc`` pulse model ``f_u`` (carrier cos/sin channels) is PINNED since round 3:
    ``tests/golden/make_fu_goldens.py`` slices ``my_expit`` / ``bspline`` / ``f_u`` out of
    /root/reference/diffqc.cc at generation time, compiles them with g++ beside a driver and stores
    the tables only (``tests/golden/fu_goldens.npz``: carrier w != 0, both quadratures, several
    channels, the 1e-6 cutoff, the clamp at 32); ``tests/test_oracle_golden.py`` checks ``f_u`` here
    against them at 2e-15.  (The whole of diffqc.cc cannot be compiled here: Eigen is absent.)
  * The adjoint gradient does not exist in the reference; its oracle is the exact derivative of the
    reference's discrete chain via ``scipy.linalg.expm_frechet``, cross-checked against central
    finite differences of the (pinned) forward path (golden G6).

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg
from scipy.special import legendre as _legendre

# --------------------------------------------------------------------------------------
# pulse model
# --------------------------------------------------------------------------------------


def sigmoid(x: float) -> float:
    """sim_plain.py:49-50 (math.exp, no clamping)."""
    return 1.0 / (1.0 + math.exp(-x))


def my_expit(x: float) -> float:
    """diffqc.cc:75-80 (clamped at +-32)."""
    if x > 32.0:
        return 1.0
    if x < -32.0:
        return 0.0
    return 1.0 / (1.0 + math.exp(-x))


def bspline_py(b: int, n_basis: int, t: float) -> float:
    """sim_plain.py:52-70: bump b, zero when t >= r or t <= l."""
    tau = 1.0 / (n_basis - 2)
    tau_b = tau * (b - 1.5)
    norm_factor = -((1.5 * tau) ** 2)
    l = tau_b - 1.5 * tau
    r = tau_b + 1.5 * tau
    if t >= r or t <= l:
        return 0.0
    return (t - l) * (t - r) / norm_factor


def bspline_cc(b: int, n_basis: int, t: float) -> float:
    """diffqc.cc:82-93: same bump, `t < r and t > l`, norm -(1.5 tau)*(1.5 tau)."""
    tau = 1.0 / (n_basis - 2.0)
    tau_b = tau * (b - 1.5)
    l = tau_b - 1.5 * tau
    r = tau_b + 1.5 * tau
    ans = 0.0
    if t < r and t > l:
        norm_factor = -(1.5 * tau) * (1.5 * tau)
        ans = (t - l) * (t - r) / norm_factor
    return ans


def basis_value(basis: str, j: int, n_basis: int, t: float, T: float, legendre_ps=None) -> float:
    """One term phi_j(t) of sim_plain.py:85-94 (without the coefficient)."""
    if basis == "poly":
        return (t - 0.5) ** j
    if basis == "Legendre":
        p = legendre_ps[j] if legendre_ps is not None else _legendre(j)
        return float(p(2 * t / T - 1))
    if basis == "BSpline":
        return bspline_py(j, n_basis, t / T)
    raise ValueError("basis_value: unsupported basis " + basis)


def generate_u_value(coeff_i, omega_i: float, t: float, T: float, basis: str, n_basis: int, legendre_ps=None) -> float:
    """sim_plain.py:73-99 `_u(t, args)`: omega_i * (2 sigmoid(sum_j c_ij phi_j(t)) - 1)."""
    u = 0
    n = int(n_basis / 2) if basis == "Fourier" else n_basis
    for j in range(n):
        if basis == "Fourier":
            u += coeff_i[j] * np.cos(2 * np.pi * j * t) + coeff_i[j + n] * np.sin(2 * np.pi * j * t)
        else:
            u += coeff_i[j] * basis_value(basis, j, n_basis, t, T, legendre_ps)
    return (sigmoid(u) * 2 - 1) * omega_i


def n_steps_py(per_step: int, T0: float, T: float) -> int:
    """sim_plain.py:123 (no abs)."""
    return int(per_step * ((T - T0) + 1))


def n_steps_cc(per_step: int, T0: float, T: float) -> int:
    """diffqc.cc:182 (with abs)."""
    return int(per_step * (abs(T - T0) + 1))


def time_grid(T0: float, T: float, n_steps: int):
    """Left-endpoint samples by repeated addition `t += dt` (sim_plain.py:133-134,150; diffqc.cc:183-184,199)."""
    dt = (T - T0) / n_steps
    ts = np.empty(n_steps)
    t = T0
    for k in range(n_steps):
        ts[k] = t
        t += dt
    return ts, dt


def generate_u_table(coeff, omegas, T: float, ts, basis: str = "BSpline"):
    """u[t][k] for the sim_plain pulse model on the grid `ts`."""
    coeff = np.asarray(coeff, dtype=np.float64)
    K, n_basis = coeff.shape
    leg = [_legendre(j) for j in range(n_basis)] if basis == "Legendre" else None
    out = np.empty((len(ts), K))
    for a, t in enumerate(ts):
        for i in range(K):
            out[a, i] = generate_u_value(coeff[i], omegas[i], float(t), T, basis, n_basis, leg)
    return out


def legendre_cc(n: int, x: float) -> float:
    """std::legendre(n, x) (diffqc.cc:117) by the three-term recurrence."""
    if n == 0:
        return 1.0
    p0, p1 = 1.0, x
    for k in range(1, n):
        p0, p1 = p1, ((2 * k + 1) * x * p1 - k * p0) / (k + 1)
    return p1


def f_u(h: int, t: float, vv, channels, duration: float, func_type: int) -> float:
    """diffqc.cc:95-135."""
    ans = 0.0
    n_basis = len(vv[0][0])
    for chan in channels[h]:
        omega = chan[1]
        w = chan[2]
        idx = int(round(chan[3]))
        A = 0.0
        B = 0.0
        for j in range(n_basis):
            if func_type == 0:
                fv = legendre_cc(j, 2 * t / duration - 1)
            else:
                fv = bspline_cc(j, n_basis, t / duration)
            A += vv[0][idx][j] * fv
            B += vv[1][idx][j] * fv
        N = math.sqrt(A * A + B * B)
        if abs(N - 0.0) < 0.000001:
            ans += 0.0
        else:
            ans += omega * (2 * my_expit(N) - 1) / N * (math.cos(w * t) * A + math.sin(w * t) * B)
    return ans


def f_u_table(vv, channels, duration: float, func_type: int, ts):
    K = len(channels)
    out = np.empty((len(ts), K))
    for a, t in enumerate(ts):
        for h in range(K):
            out[a, h] = f_u(h, float(t), vv, channels, duration, func_type)
    return out


# --------------------------------------------------------------------------------------
# propagator chain
# --------------------------------------------------------------------------------------


def step_generator(H0, Hs, u_row, dt):
    """dH of one step, accumulated term by term as sim_plain.py:137-143 / diffqc.cc:192-197."""
    dH = -1.0j * dt * H0
    for k in range(len(Hs)):
        dH = dH + (-1.0j * dt * u_row[k]) * Hs[k]
    return dH


def propagate(H0, Hs, u, dt: float, psi):
    """psi <- expm(dH_t) psi for t = 0..S-1; psi is [d] or [d][B] (sim_plain.py:135-150, diffqc.cc:190-200)."""
    H0 = np.asarray(H0, dtype=np.complex128)
    Hs = [np.asarray(h, dtype=np.complex128) for h in Hs]
    psi = np.array(psi, dtype=np.complex128)
    for t in range(len(u)):
        U = scipy.linalg.expm(step_generator(H0, Hs, u[t], dt))
        psi = np.matmul(U, psi)
    return psi


def trotter_sim_plain(H0, Hs, coeff, omegas, Tpulse: float, basis: str, per_step: int, psi0, T0: float, T: float):
    """SimulatorPlain.trotter (sim_plain.py:119-153) for H = [H0, [Hs[i], generate_u(i, coeff)] ...]."""
    n_steps = n_steps_py(per_step, T0, T)
    ts, dt = time_grid(T0, T, n_steps)
    u = generate_u_table(coeff, omegas, Tpulse, ts, basis)
    psi = np.asarray(psi0, dtype=np.complex128).reshape(len(H0), -1)
    return propagate(H0, Hs, u, dt, psi), u, dt


def trotter_diffqc(H0, Hs, channels, duration, func_type, psi0, T0, T, per_step, vv):
    """diffqc.trotter after diffqc.set_H (diffqc.cc:43-73,173-205)."""
    n_steps = n_steps_cc(per_step, T0, T)
    ts, dt = time_grid(T0, T, n_steps)
    u = f_u_table(vv, channels, duration, func_type, ts)
    psi = np.asarray(psi0, dtype=np.complex128).reshape(-1)
    return propagate(H0, Hs, u, dt, psi), u, dt


def energies(M, psi):
    """Re <psi_b| M |psi_b> per column (sim_plain.py:281)."""
    psi = np.asarray(psi, dtype=np.complex128).reshape(len(M), -1)
    return np.real(np.sum(psi.conj() * (np.asarray(M) @ psi), axis=0))


# --------------------------------------------------------------------------------------
# gradients
# --------------------------------------------------------------------------------------


def energy_grad(H0, Hs, M, u, dt: float, psi0):
    """Exact derivative of L = sum_b <psi_b(S)|M|psi_b(S)> of the discrete chain w.r.t. u[t][k].

    Not in the reference.  Uses scipy.linalg.expm_frechet, i.e. a
    different algorithm from the GPU engine's reverse-mode Taylor polynomial.
    Returns (loss, grad[S][K], psi_S).
    """
    H0 = np.asarray(H0, dtype=np.complex128)
    Hs = [np.asarray(h, dtype=np.complex128) for h in Hs]
    M = np.asarray(M, dtype=np.complex128)
    d = len(H0)
    S, K = len(u), len(Hs)
    psi = np.array(psi0, dtype=np.complex128).reshape(d, -1)
    states = [psi]
    As = []
    Us = []
    for t in range(S):
        A = step_generator(H0, Hs, u[t], dt)
        U = scipy.linalg.expm(A)
        As.append(A)
        Us.append(U)
        psi = U @ psi
        states.append(psi)
    loss = float(np.sum(energies(M, psi)))
    lam = 2.0 * (M @ psi)
    grad = np.zeros((S, K))
    for t in range(S - 1, -1, -1):
        for k in range(K):
            E = -1.0j * dt * Hs[k]
            dU = scipy.linalg.expm_frechet(As[t], E, compute_expm=False)
            grad[t, k] = np.real(np.sum(lam.conj() * (dU @ states[t])))
        lam = Us[t].conj().T @ lam
    return loss, grad, states[-1]


def energy_grad_one_frechet(H0, Hs, M, u, dt: float, psi0):
    """Same quantity as energy_grad, organised for large d: ONE scipy.linalg.expm_frechet per step.

    Adjoint-Frechet identity: <G, L_exp(A, E)> = <L_exp(A^H, G), E>, so with
    G_t = Lambda_{t+1} Psi_t^H the K control gradients of a step are K contractions of Abar_t = L_exp(A_t^H, G_t).
    Per step: expm (forward), expm_frechet (also returns exp(A^H) = U^H), three (d x d)(d x B) products.
    This is the CPU baseline of bench.py for the fwd+adjoint metric.  Returns (loss, grad[S][K], psi_S)."""
    H0 = np.asarray(H0, dtype=np.complex128)
    Hs = [np.asarray(h, dtype=np.complex128) for h in Hs]
    M = np.asarray(M, dtype=np.complex128)
    d = len(H0)
    S, K = len(u), len(Hs)
    psi = np.array(psi0, dtype=np.complex128).reshape(d, -1)
    states = [psi]
    for t in range(S):
        psi = np.matmul(scipy.linalg.expm(step_generator(H0, Hs, u[t], dt)), psi)
        states.append(psi)
    loss = float(np.sum(energies(M, psi)))
    lam = 2.0 * (M @ psi)
    grad = np.zeros((S, K))
    for t in range(S - 1, -1, -1):
        A = step_generator(H0, Hs, u[t], dt)
        G = lam @ states[t].conj().T
        Uh, Abar = scipy.linalg.expm_frechet(A.conj().T, G, compute_expm=True)
        for k in range(K):
            grad[t, k] = dt * float(np.sum(np.imag(Abar.conj() * Hs[k])))  # Re<Abar, -i dt H_k>
        lam = Uh @ lam
    return loss, grad, states[-1]


def fidelity_grad(H0, Hs, u, dt: float, psi0, targets):
    """L = sum_b (1 - |<phi_b|psi_b(S)>|^2) and its exact derivative w.r.t. u[t][k] (expm_frechet).

    The batched, Trotter-chain form of train_fidelity's objective (sim_plain.py:436-455, M = |psi1><psi1|).
    Returns (loss, grad[S][K], psi_S, fidelities[B])."""
    H0 = np.asarray(H0, dtype=np.complex128)
    Hs = [np.asarray(h, dtype=np.complex128) for h in Hs]
    d = len(H0)
    S, K = len(u), len(Hs)
    psi = np.array(psi0, dtype=np.complex128).reshape(d, -1)
    phi = np.array(targets, dtype=np.complex128).reshape(d, -1)
    states, As, Us = [psi], [], []
    for t in range(S):
        A = step_generator(H0, Hs, u[t], dt)
        U = scipy.linalg.expm(A)
        As.append(A)
        Us.append(U)
        psi = U @ psi
        states.append(psi)
    c = np.sum(phi.conj() * psi, axis=0)
    fid = np.abs(c) ** 2
    loss = float(np.sum(1.0 - fid))
    lam = -2.0 * phi * c[None, :]
    grad = np.zeros((S, K))
    for t in range(S - 1, -1, -1):
        for k in range(K):
            dU = scipy.linalg.expm_frechet(As[t], -1.0j * dt * Hs[k], compute_expm=False)
            grad[t, k] = np.real(np.sum(lam.conj() * (dU @ states[t])))
        lam = Us[t].conj().T @ lam
    return loss, grad, states[-1], fid


def df_u_dvv_fd(vv, channels, duration, func_type, ts, eps=1e-6):
    """Central differences of f_u_table w.r.t. every vv entry: [S][K][2][n_param][n_basis]."""
    vv = np.array(vv, dtype=np.float64)
    out = np.zeros((len(ts), len(channels)) + vv.shape)
    for idx in np.ndindex(*vv.shape):
        vp, vm = vv.copy(), vv.copy()
        vp[idx] += eps
        vm[idx] -= eps
        out[(slice(None), slice(None)) + idx] = (np.asarray(f_u_table(vp, channels, duration, func_type, ts)) -
                                                 np.asarray(f_u_table(vm, channels, duration, func_type, ts))) / (2 * eps)
    return out


def energy_grad_fd(H0, Hs, M, u, dt: float, psi0, eps: float = 1e-6):
    """Central finite differences of the forward oracle w.r.t. every u[t][k]."""
    u = np.array(u, dtype=np.float64)
    g = np.zeros_like(u)
    for t in range(u.shape[0]):
        for k in range(u.shape[1]):
            up = u.copy(); up[t, k] += eps
            um = u.copy(); um[t, k] -= eps
            lp = float(np.sum(energies(M, propagate(H0, Hs, up, dt, psi0))))
            lm = float(np.sum(energies(M, propagate(H0, Hs, um, dt, psi0))))
            g[t, k] = (lp - lm) / (2 * eps)
    return g


def expm_vjp(A, Ubar):
    """Abar with Re tr(Abar^H E) = Re tr(Ubar^H L_exp(A, E)) for all E:  Abar = L_exp(A^H, Ubar)."""
    return scipy.linalg.expm_frechet(np.asarray(A).conj().T, np.asarray(Ubar), compute_expm=False)


def du_dcoeff(coeff, omegas, T: float, ts, basis: str = "BSpline"):
    """d u[t][k] / d coeff[k][j]  (chain rule of the squashing in sim_plain.py:96-98): [S][K][n_basis]."""
    coeff = np.asarray(coeff, dtype=np.float64)
    K, n_basis = coeff.shape
    leg = [_legendre(j) for j in range(n_basis)] if basis == "Legendre" else None
    out = np.zeros((len(ts), K, n_basis))
    for a, t in enumerate(ts):
        phi = np.array([basis_value(basis, j, n_basis, float(t), T, leg) for j in range(n_basis)])
        for i in range(K):
            s = sigmoid(float(coeff[i] @ phi))
            out[a, i] = omegas[i] * 2.0 * s * (1.0 - s) * phi
    return out


def mc_gradient(H0, Hs, M, coeff, omegas, T: float, basis: str, per_step: int, psi0, s: float, r: float = 0.5):
    """compute_energy_grad_MC (sim_plain.py:156-231) for a given sample time `s` (batch 1, no noise).

    Returns (grad_coeff[K][n_basis], ps[K], phi(s)).
    """
    H0 = np.asarray(H0, dtype=np.complex128)
    Hs = [np.asarray(h, dtype=np.complex128) for h in Hs]
    K, n_basis = np.asarray(coeff).shape
    d = len(H0)
    dDdv = du_dcoeff(coeff, omegas, T, [s], basis)[0]  # torch autograd of sim_plain.py:170-184
    phi, _, _ = trotter_sim_plain(H0, Hs, coeff, omegas, T, basis, per_step, psi0, 0, s)
    grad = np.zeros((K, n_basis))
    ps_all = np.zeros(K)
    for i in range(K):
        gate_p = (np.eye(d) + r * 1.0j * Hs[i]) / np.sqrt(1.0 + r ** 2)
        gate_m = (np.eye(d) - r * 1.0j * Hs[i]) / np.sqrt(1.0 + r ** 2)
        ket_p, _, _ = trotter_sim_plain(H0, Hs, coeff, omegas, T, basis, per_step, gate_p @ phi, s, T)
        ps_p = complex((ket_p.conj().T @ (M @ ket_p)).reshape(-1)[0])
        ket_m, _, _ = trotter_sim_plain(H0, Hs, coeff, omegas, T, basis, per_step, gate_m @ phi, s, T)
        ps_m = complex((ket_m.conj().T @ (M @ ket_m)).reshape(-1)[0])
        ps = ((1 + r ** 2) / 2 / r * (ps_m - ps_p)).real
        ps_all[i] = ps
        grad[i] = ps * dDdv[i]
    return grad, ps_all, phi


# --------------------------------------------------------------------------------------
# synthetic problems shared by tests and bench
# --------------------------------------------------------------------------------------

PAULI = {
    "I": np.eye(2, dtype=np.complex128),
    "X": np.array([[0, 1], [1, 0]], dtype=np.complex128),
    "Y": np.array([[0, -1j], [1j, 0]], dtype=np.complex128),
    "Z": np.array([[1, 0], [0, -1]], dtype=np.complex128),
}


def kron_all(ops):
    out = np.array([[1.0 + 0j]])
    for o in ops:
        out = np.kron(out, o)
    return out


def pauli_on(n: int, which: dict):
    return kron_all([PAULI[which.get(q, "I")] for q in range(n)])
