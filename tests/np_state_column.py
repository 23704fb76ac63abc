"""Test infrastructure: a numpy model of the state-column form (diffquantum_amd/csrc/statecol.hip), row for row the
recurrences the kernels run -- it fixes the signs and the pairing of the gradient levels before (and beside) the HIP code.
Never imported by the product.

The chain is the reference's `psi <- expm(-1j * dt * H(t)) psi` (sim_plain.py:135-150, diffqc.cc:190-200) for a Hermitian
generator that is a sum of Pauli strings, given as the table the engine extracts at dq_set_hamiltonians (engine.hip
find_xor_regular / find_zxor_regular):  H_src[i, i ^ xoff[j]] = (-1)^sgn[i]_j * cmag[j] * (1j if phase[j] else 1)  for the
XOR-diagonal j of source kid[j] (-1 = drift), and real diagonals dval[jd][i] of source dkid[jd].

Step:  psi' = sum_{m <= L} A^m psi / m!,  A = -1j dt H(u_t)  (Taylor polynomial; L from the remainder bound).
Backward step t (lam = lambda_{t+1}, psi1 = psi_{t+1}):
    psi_t = sum_m (-A)^m psi1 / m!                                   (reversal: U^H = exp(-A))
    x_l   = A^l psi_t / l!,  l = 0 .. L-1                             (the forward terms of psi_t)
    y_{L-1} = lam / L,   y_{l-1} = (lam - A y_l) / l                  (Horner, descending)
    lam_t = lam - A y_0                                               (= U^H lam)
    dL/du_k[t] = Re sum_l <y_l, dA_k x_l>,   dA_k = -1j dt H_k
with <y, dA_k x> evaluated diagonal by diagonal from the partner values y[i ^ xoff] that the sweep A y fetches anyway.
Round 4 added the quadrature form of the backward step (backward_step_quad: 2 L sweeps, the headline's kernel) and the
Chebyshev-economized step polynomial (poly_factors / economized_degree: p(A) = sum_m g[m] A^m / m!, one to four degrees below Taylor's
at the same sup-norm error on the spectrum i [-theta, theta]): every step function takes the level factors `g` (None = Taylor).
"""
from math import factorial

import numpy as np

_LD = np.longdouble


def _bessel(k, x):
    x = _LD(x)
    return sum((_LD(-1) ** j) * (x / 2) ** (k + 2 * j) / (_LD(factorial(j)) * _LD(factorial(k + j))) for j in range(40))


def poly_factors(theta, L):
    """g[m] = m! x the coefficient of z^m (z = i theta x) in J_0(theta) + 2 sum_{k <= L} i^k J_k(theta) T_k(x)  (csrc/statecol.hip
    xs_poly_factors; extended precision: g[m] = 1 - O(theta^2))."""
    if theta == 0:
        return np.ones(L + 1)
    T = [np.zeros(L + 1, dtype=_LD) for _ in range(L + 1)]
    T[0][0] = 1
    if L >= 1:
        T[1][1] = 1
    for k in range(2, L + 1):
        T[k][1:] = 2 * T[k - 1][:-1]
        T[k] -= T[k - 2]
    g = np.zeros(L + 1)
    for m in range(L + 1):
        acc = _LD(0)
        for k in range(m, L + 1, 2):
            acc += (1 if k == 0 else 2) * _bessel(k, theta) * T[k][m] * (_LD(-1) ** ((k - m) // 2))
        g[m] = float(acc / _LD(theta) ** m * _LD(factorial(m)))
    return g


def economized_degree(theta, tol=1.1102230246251565e-16):
    """The lowest L with 2 sum_{k > L} |J_k(theta)| <= tol (csrc/statecol.hip xs_poly_make)."""
    for L in range(1, 25):
        if float(2 * sum(abs(_bessel(k, theta)) for k in range(L + 1, L + 40))) <= tol:
            return L
    raise ValueError("step too large for the polynomial alone")


def table_from_dense(H0, Hs, tol=0.0):
    """The XOR-regular table of (H0, Hs) or None when the pattern is not of that kind (mirrors find_zxor_regular)."""
    H0 = np.asarray(H0)
    d = H0.shape[0]
    srcs = [(-1, H0)] + [(k, np.asarray(h)) for k, h in enumerate(Hs)]
    idx = np.arange(d)
    xoff, kid, cmag, phase, sgn, dkid, dval = [], [], [], [], [], [], []
    for k, H in srcs:
        dg = np.diag(H)
        if np.any(np.abs(dg.imag) > tol):
            return None
        if np.any(dg.real != 0):
            dkid.append(k)
            dval.append(dg.real.copy())
        for x in range(1, d):
            v = H[idx, idx ^ x]
            if not np.any(v != 0):
                continue
            if np.any(v == 0):
                return None
            ph = 1 if np.all(v.real == 0) else 0
            if ph == 0 and np.any(v.imag != 0):
                return None
            w = v.imag if ph else v.real
            if np.any(np.abs(np.abs(w) - abs(w[0])) > tol):
                return None
            xoff.append(x); kid.append(k); cmag.append(abs(w[0])); phase.append(ph); sgn.append((w < 0).astype(np.int64))
    return dict(d=d, xoff=np.array(xoff, dtype=np.int64), kid=np.array(kid, dtype=np.int64), cmag=np.array(cmag),
                phase=np.array(phase, dtype=np.int64), sgn=np.array(sgn).reshape(len(xoff), d), dkid=np.array(dkid, dtype=np.int64),
                dval=np.array(dval).reshape(len(dkid), d))


def _weights(tab, u_row, dt):
    uk = lambda k: 1.0 if k < 0 else u_row[k]
    w = np.array([dt * uk(k) * c for k, c in zip(tab["kid"], tab["cmag"])])
    dg = np.zeros(tab["d"])
    for jd, k in enumerate(tab["dkid"]):
        dg += dt * uk(k) * tab["dval"][jd]
    return w, dg


def apply_A(tab, w, dg, v, sign=1.0, dots=None, xm=None):
    """o = sign * A v for A = -1j (diag(dg) + sum_j w_j S_j), S_j[i, i ^ x_j] = sigma_ij * (1j if phase_j else 1).
    dots (list, one accumulator per XOR-diagonal) += the level's gradient pairing of the fetched partners with xm."""
    idx = np.arange(tab["d"])
    o = sign * (-1j) * dg * v
    for j, x in enumerate(tab["xoff"]):
        sig = 1.0 - 2.0 * tab["sgn"][j]
        p = v[idx ^ x]                      # the partner value the kernel reads from LDS
        if tab["phase"][j]:
            o = o + sign * w[j] * sig * p   # (-1j)(1j) = 1
        else:
            o = o + sign * (-1j) * w[j] * sig * p
        if dots is not None:
            # Re <y, dA x> for this diagonal, re-indexed onto the row that fetched the partner: entry H[i ^ x, i] = conj(H[i, i ^ x])
            if tab["phase"][j]:
                dots[j] += -np.sum(sig * (p.real * xm.real + p.imag * xm.imag))
            else:
                dots[j] += np.sum(sig * (p.real * xm.imag - p.imag * xm.real))
    return o


def forward_step(tab, u_row, dt, psi, L, g=None):
    g = np.ones(L + 1) if g is None else g
    w, dg = _weights(tab, u_row, dt)
    cur, acc = psi, g[0] * psi
    for m in range(1, L + 1):
        cur = apply_A(tab, w, dg, cur) / m
        acc = acc + g[m] * cur
    return acc


def backward_step(tab, u_row, dt, psi1, lam, L, K, g=None):
    """(psi_t, lam_t, grad_row[K]) from (psi_{t+1}, lam_{t+1}).  With level factors g the step is p(A) = sum g[m] A^m / m!:
    the Horner levels become y_{L-1} = g[L] lam / L, y_{l-1} = (g[l] lam - A y_l) / l, lam_t = g[0] lam - A y_0 (so that
    y_l = sum_{m > l} g[m] (l! / m!) (-A)^(m-1-l) lam, and dp(A)[E] pairs y_l with E x_l exactly as before)."""
    g = np.ones(L + 1) if g is None else g
    w, dg = _weights(tab, u_row, dt)
    R = len(tab["xoff"])
    cur, psi = psi1, g[0] * psi1
    for m in range(1, L + 1):                       # reversal
        cur = apply_A(tab, w, dg, cur, sign=-1.0) / m
        psi = psi + g[m] * cur
    xs = [psi]
    for l in range(1, L):                           # the forward terms of psi_t
        xs.append(apply_A(tab, w, dg, xs[-1]) / l)
    dots = [0.0] * R
    rowacc = np.zeros(tab["d"])
    y = g[L] * lam / L
    for l in range(L - 1, 0, -1):                   # sweep forming y_{l-1} from y_l pairs y_l with x_l
        rowacc += y.real * xs[l].imag - y.imag * xs[l].real
        y = (g[l] * lam - apply_A(tab, w, dg, y, dots=dots, xm=xs[l])) / l
    rowacc += y.real * xs[0].imag - y.imag * xs[0].real
    lam_t = g[0] * lam - apply_A(tab, w, dg, y, dots=dots, xm=xs[0])
    g = np.zeros(K)
    for j in range(R):
        if tab["kid"][j] >= 0:
            g[tab["kid"][j]] += dt * tab["cmag"][j] * dots[j]
    for jd, k in enumerate(tab["dkid"]):
        if k >= 0:
            g[k] += dt * np.sum(tab["dval"][jd] * rowacc)
    return psi, lam_t, g


def lobatto(npts):
    """Nodes and weights of the npts-point Gauss-Lobatto rule on [0, 1] (end points included; exact to degree 2 npts - 3)."""
    from numpy.polynomial import legendre as Lg
    c = np.zeros(npts)
    c[-1] = 1.0                                   # P_{npts-1}
    xi = np.sort(np.real(Lg.legroots(Lg.legder(c))))
    x = np.concatenate([[-1.0], xi, [1.0]])
    w = 2.0 / (npts * (npts - 1) * Lg.legval(x, c) ** 2)
    return (x + 1.0) / 2.0, w / 2.0


def backward_step_quad(tab, u_row, dt, psi1, lam, L, K, npts=4, g=None, theta=None):
    """The QUADRATURE form of the backward step (csrc/statecol_bwdq.hip): (psi_t, lam_t, grad_row[K]) from (psi_{t+1}, lam_{t+1}).

    Both vectors run backward under the same operator, phi(s) = exp(-s A) phi_{t+1}, and dexp(A)[E] = int_0^1 exp((1-s)A) E exp(sA) ds
    gives dL/du_k[t] = Re int_0^1 <lam(s), dA_k psi(s)> ds -- evaluated by a Gauss-Lobatto rule whose node values are sums of the
    Taylor terms the propagation produces anyway:
        phase 1:  z_m = (-A)^m lam / m!,  lam_t = sum_m z_m,  node values lam(s_i) = sum_m s_i^m z_m
        phase 2:  r_m = (-A)^m psi1 / m!, psi_t = sum_m r_m;  the sweep that forms r_{m+1} pairs the partner values r_m[i ^ x] it fetches
                  with the own rows of q_m = sum_i w_i s_i^m lam(s_i):  sum_i w_i <lam(s_i), dA psi(s_i)> = sum_m <q_m, dA r_m>
                  = -sum_m <r_m, dA q_m>  (dA anti-Hermitian), which is the form the pairing code of apply_A takes.
    With an economized polynomial (g, theta given) exp(-s_i A) is the degree-L polynomial of the interval s_i theta, with its own
    level factors g_i: every s_i^m carries g_i[m], the end point 1 carries g[m]."""
    w, dg = _weights(tab, u_row, dt)
    R = len(tab["xoff"])
    s, wq = lobatto(npts)
    g = np.ones(L + 1) if g is None else g
    gi = [None] + [np.ones(L + 1) if theta is None else poly_factors(theta * s[i], L) for i in range(1, npts - 1)] + [g]
    cur = lam
    nodes = [lam.copy()] + [gi[i][0] * lam for i in range(1, npts)]   # node 0 (s = 0) stays lam_{t+1}; the last (s = 1) becomes lam_t
    for m in range(1, L + 1):
        cur = apply_A(tab, w, dg, cur, sign=-1.0) / m
        for i in range(1, npts):
            nodes[i] = nodes[i] + gi[i][m] * (s[i] ** m) * cur
    lam_t = nodes[-1]
    dots = [0.0] * R
    rowacc = np.zeros(tab["d"])
    cur, psi = psi1, g[0] * psi1
    for m in range(L):
        q = sum(wq[i] * gi[i][m] * (s[i] ** m) * nodes[i] for i in range(1, npts))
        if m == 0:
            q = q + wq[0] * nodes[0]
        q = -q                                            # (the sign of <q, dA r> = -<r, dA q>)
        rowacc += cur.real * q.imag - cur.imag * q.real
        cur = apply_A(tab, w, dg, cur, sign=-1.0, dots=dots, xm=q) / (m + 1)
        psi = psi + g[m + 1] * cur
    g = np.zeros(K)
    for j in range(R):
        if tab["kid"][j] >= 0:
            g[tab["kid"][j]] += dt * tab["cmag"][j] * dots[j]
    for jd, k in enumerate(tab["dkid"]):
        if k >= 0:
            g[k] += dt * np.sum(tab["dval"][jd] * rowacc)
    return psi, lam_t, g


def taylor_degree(theta, tol=1.1102230246251565e-16):
    for L in range(1, 25):
        v = 1.0
        for j in range(1, L + 2):
            v *= theta / j
        if v <= tol:
            return L
    raise ValueError("step too large for the polynomial alone")


def energy_grad(H0, Hs, M, u, dt, psi0, L=None, quad_points=0, economized=False):
    """(loss, grad[S][K], psi_S) of L = sum_b <psi_b|M|psi_b> through the state-column recurrences (quad_points > 0: the backward
    steps in the quadrature form with that many Gauss-Lobatto points; economized: the Chebyshev-economized step polynomial)."""
    tab = table_from_dense(H0, Hs)
    assert tab is not None, "not an XOR-regular generator"
    u = np.asarray(u, dtype=np.float64)
    S, K = u.shape
    n1 = lambda A: np.abs(A).sum(axis=0).max()
    th = max(abs(dt) * (n1(H0) + sum(abs(row[k]) * n1(Hs[k]) for k in range(K))) for row in u) if S else 0.0
    if L is None:
        L = economized_degree(th) if economized else taylor_degree(th)
    g = poly_factors(th, L) if economized else None
    tq = th if economized else None
    psi0 = np.asarray(psi0, dtype=np.complex128)
    B = psi0.shape[1]
    grad = np.zeros((S, K))
    fin = np.empty_like(psi0)
    loss = 0.0
    for b in range(B):
        psi = psi0[:, b]
        for t in range(S):
            psi = forward_step(tab, u[t], dt, psi, L, g)
        fin[:, b] = psi
        mp = M @ psi
        loss += float(np.real(np.vdot(psi, mp)))
        lam = 2.0 * mp
        for t in range(S - 1, -1, -1):
            if quad_points:
                psi, lam, gr = backward_step_quad(tab, u[t], dt, psi, lam, L, K, quad_points, g, tq)
            else:
                psi, lam, gr = backward_step(tab, u[t], dt, psi, lam, L, K, g)
            grad[t] += gr
    return loss, grad, fin
