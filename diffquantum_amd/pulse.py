"""Pulse model of the propagator path: turns parameter tables into the per-step control
amplitudes u[t][k] that the HIP engine consumes, and provides their parameter derivatives for the
chain rule behind the adjoint gradient.

Two parametrisations exist in the reference and both are mirrored (vectorised over the time grid):
  * ``SimulatorPlain.generate_u`` (/root/reference/sim_plain.py:73-99): u_i(t) = w_i (2 sigmoid(a_i(t)) - 1),
    a_i(t) = sum_j c_ij phi_j(t), phi in {poly, Legendre, Fourier, BSpline-bump};
  * ``f_u`` of the C++ module (/root/reference/diffqc.cc:95-135): per channel A, B amplitudes on a
    cos/sin carrier, squashed by 2*expit(N)-1 with N = sqrt(A^2+B^2) and a 1e-6 cutoff.
Host numpy, O(S*K*n_basis) scalar work per gradient evaluation -- negligible next to the chain.
"""
from __future__ import annotations

import math

import numpy as np

BASES = ("poly", "Legendre", "Fourier", "BSpline")


def time_grid(T0: float, T: float, n_steps: int):
    """Left-endpoint sample times, built by repeated ``t += dt`` like the reference's loops
    (sim_plain.py:133-134,150; diffqc.cc:183-184,199) so that the samples agree to the last bit."""
    if n_steps <= 0:
        raise ValueError("n_steps must be positive (got {}); sim_plain.py:123 gives int(per_step*((T-T0)+1))".format(n_steps))
    dt = (T - T0) / n_steps
    ts = np.empty(n_steps, dtype=np.float64)
    t = T0
    for k in range(n_steps):
        ts[k] = t
        t += dt
    return ts, dt


def n_steps_sim_plain(per_step: int, T0: float, T: float) -> int:
    return int(per_step * ((T - T0) + 1))  # sim_plain.py:123


def n_steps_diffqc(per_step: int, T0: float, T: float) -> int:
    return int(per_step * (abs(T - T0) + 1))  # diffqc.cc:182


def bump_table(n_basis: int, x):
    """phi[j][a] of the reference's "BSpline" bumps at x[a] in units of the pulse length
    (sim_plain.py:52-70, diffqc.cc:82-93): support (tau(b-3), tau b), open at both ends."""
    x = np.asarray(x, dtype=np.float64)
    tau = 1.0 / (n_basis - 2)
    out = np.zeros((n_basis, x.size))
    norm = -((1.5 * tau) ** 2)
    for b in range(n_basis):
        c = tau * (b - 1.5)
        l, r = c - 1.5 * tau, c + 1.5 * tau
        inside = (x < r) & (x > l)
        out[b] = np.where(inside, (x - l) * (x - r) / norm, 0.0)
    return out


def legendre_table(n_basis: int, x):
    """P_j(x[a]) for j < n_basis by the Bonnet recurrence (scipy.special.legendre / std::legendre)."""
    x = np.asarray(x, dtype=np.float64)
    out = np.empty((n_basis, x.size))
    if n_basis > 0:
        out[0] = 1.0
    if n_basis > 1:
        out[1] = x
    for k in range(1, n_basis - 1):
        out[k + 1] = ((2 * k + 1) * x * out[k] - k * out[k - 1]) / (k + 1)
    return out


def basis_table(basis: str, n_basis: int, ts, T: float):
    """phi[j][a]: the functions multiplying coeff[:, j] in generate_u, at times ts[a].
    For 'Fourier' the table has n_basis rows: cos(2 pi j t) for j < n/2 then sin(2 pi j t)."""
    ts = np.asarray(ts, dtype=np.float64)
    if basis == "BSpline":
        return bump_table(n_basis, ts / T)
    if basis == "Legendre":
        return legendre_table(n_basis, 2 * ts / T - 1)
    if basis == "poly":
        return np.stack([(ts - 0.5) ** j for j in range(n_basis)])
    if basis == "Fourier":
        n = int(n_basis / 2)
        rows = [np.cos(2 * np.pi * j * ts) for j in range(n)] + [np.sin(2 * np.pi * j * ts) for j in range(n)]
        rows += [np.zeros_like(ts)] * (n_basis - 2 * n)
        return np.stack(rows)
    raise ValueError("unknown basis {!r}; expected one of {}".format(basis, BASES))


def _sigmoid(a):
    a = np.asarray(a, dtype=np.float64)
    if np.any(a < -709.0):
        # the reference's math.exp(-x) overflows here (sim_plain.py:49-50)
        raise OverflowError("math range error")
    return 1.0 / (1.0 + np.exp(-a))


def u_table(coeff, omegas, T: float, ts, basis: str = "BSpline"):
    """u[a][i] = omegas[i] * (2 sigmoid(coeff[i] . phi(ts[a])) - 1)."""
    coeff = np.asarray(coeff, dtype=np.float64)
    phi = basis_table(basis, coeff.shape[1], ts, T)
    a = coeff @ phi
    return np.ascontiguousarray(((2.0 * _sigmoid(a) - 1.0) * np.asarray(omegas, dtype=np.float64)[:, None]).T)


def du_dcoeff(coeff, omegas, T: float, ts, basis: str = "BSpline"):
    """d u[a][i] / d coeff[i][j] -> [S][K][n_basis]."""
    coeff = np.asarray(coeff, dtype=np.float64)
    phi = basis_table(basis, coeff.shape[1], ts, T)  # [n_basis][S]
    s = _sigmoid(coeff @ phi)  # [K][S]
    w = np.asarray(omegas, dtype=np.float64)[:, None] * 2.0 * s * (1.0 - s)  # [K][S]
    return np.einsum("ka,ja->akj", w, phi)


def grad_coeff_from_grad_u(grad_u, coeff, omegas, T: float, ts, basis: str = "BSpline"):
    """Chain rule: dL/dcoeff[i][j] = sum_a dL/du[a][i] * du[a][i]/dcoeff[i][j]."""
    return np.einsum("ak,akj->kj", np.asarray(grad_u, dtype=np.float64), du_dcoeff(coeff, omegas, T, ts, basis))


# ---- the diffqc.cc channel model --------------------------------------------------------------


def _expit32(x):
    x = np.asarray(x, dtype=np.float64)
    out = 1.0 / (1.0 + np.exp(-np.clip(x, -32.0, 32.0)))
    return np.where(x > 32.0, 1.0, np.where(x < -32.0, 0.0, out))  # diffqc.cc:75-80


def f_u_table(vv, channels, duration: float, func_type: int, ts):
    """u[a][h] = f_u(h, ts[a], vv) of diffqc.cc:95-135.

    vv: [2][n_param][n_basis]; channels[h]: list of rows (unused, omega, w, idx)."""
    vv = np.asarray(vv, dtype=np.float64)
    if vv.ndim != 3 or vv.shape[0] != 2:
        raise ValueError("vv must have shape [2][n_param][n_basis]")
    ts = np.asarray(ts, dtype=np.float64)
    n_basis = vv.shape[2]
    if func_type == 0:
        phi = legendre_table(n_basis, 2 * ts / duration - 1)
    else:
        phi = bump_table(n_basis, ts / duration)
    out = np.zeros((ts.size, len(channels)))
    for h, chans in enumerate(channels):
        for chan in chans:
            if len(chan) < 4:
                raise ValueError("each channel row needs at least 4 entries (unused, omega, w, idx)")
            omega, w, idx = float(chan[1]), float(chan[2]), int(round(chan[3]))
            if not 0 <= idx < vv.shape[1]:
                raise ValueError("channel parameter index {} out of range".format(idx))
            A = vv[0, idx] @ phi
            B = vv[1, idx] @ phi
            N = np.sqrt(A * A + B * B)
            live = np.abs(N) >= 0.000001
            Ns = np.where(live, N, 1.0)
            out[:, h] += np.where(live, omega * (2 * _expit32(N) - 1) / Ns * (np.cos(w * ts) * A + np.sin(w * ts) * B), 0.0)
    return out


def grad_vv_from_grad_u(grad_u, vv, channels, duration: float, func_type: int, ts):
    """Chain rule through the channel model of diffqc.cc:95-135: dL/dvv[c][p][j] = sum_{a,h} dL/du[a][h] du[a][h]/dvv.

    With A = vv[0][idx].phi, B = vv[1][idx].phi, N = sqrt(A^2+B^2), g(N) = omega (2 expit(N) - 1)/N and
    f = g(N) (cos(wt) A + sin(wt) B):  df/dA = g'(N) (A/N)(cos A + sin B) + g cos,  df/dB likewise with sin.
    Inside the reference's |N| < 1e-6 cutoff the function is identically 0, so is the derivative."""
    vv = np.asarray(vv, dtype=np.float64)
    grad_u = np.asarray(grad_u, dtype=np.float64)
    ts = np.asarray(ts, dtype=np.float64)
    n_basis = vv.shape[2]
    phi = legendre_table(n_basis, 2 * ts / duration - 1) if func_type == 0 else bump_table(n_basis, ts / duration)
    out = np.zeros_like(vv)
    for h, chans in enumerate(channels):
        for chan in chans:
            omega, w, idx = float(chan[1]), float(chan[2]), int(round(chan[3]))
            A = vv[0, idx] @ phi
            B = vv[1, idx] @ phi
            N = np.sqrt(A * A + B * B)
            live = np.abs(N) >= 0.000001
            Ns = np.where(live, N, 1.0)
            sg = _expit32(Ns)
            g = omega * (2 * sg - 1) / Ns
            gp = omega * (2 * sg * (1 - sg) * Ns - (2 * sg - 1)) / (Ns * Ns)
            gp = np.where(np.abs(Ns) > 32.0, -omega * (2 * sg - 1) / (Ns * Ns), gp)  # expit is clamped beyond +-32
            cs, sn = np.cos(w * ts), np.sin(w * ts)
            mix = cs * A + sn * B
            dfdA = np.where(live, gp * (A / Ns) * mix + g * cs, 0.0)
            dfdB = np.where(live, gp * (B / Ns) * mix + g * sn, 0.0)
            out[0, idx] += phi @ (grad_u[:, h] * dfdA)
            out[1, idx] += phi @ (grad_u[:, h] * dfdB)
    return out
