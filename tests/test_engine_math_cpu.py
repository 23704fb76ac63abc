"""CPU check of the mathematics behind the engine's real-symmetric fast path, in plain numpy:
the cos/sin evaluation scheme and its hand-written reverse mode are restated here exactly as engine.hip issues them
(same products, same symmetrisations, same coefficients) and compared with the reference's arithmetic --
scipy.linalg.expm (sim_plain.py:145) for the propagator and scipy.linalg.expm_frechet for its derivative.
The HIP implementation itself is tested on the GPU (tests/test_gpu_realpath.py); this file pins the algorithm.
"""
import math

import numpy as np
import pytest
import scipy.linalg


def coeffs(q):
    cc = [(-1.0) ** j / math.factorial(2 * j) for j in range(q + 1)]
    sc = [(-1.0) ** j / math.factorial(2 * j + 1) for j in range(q + 1)]
    return cc, sc


def forward(X, q):
    """U = cos X + i sin X by q + 1 products of commuting symmetric matrices (engine.hip real_forward)."""
    cc, sc = coeffs(q)
    d = len(X)
    Y = [None, X @ X]
    for j in range(2, q + 1):
        Y.append(Y[1] @ Y[j - 1])
    C = cc[0] * np.eye(d) + sum(cc[j] * Y[j] for j in range(1, q + 1))
    P = sc[0] * np.eye(d) + sum(sc[j] * Y[j] for j in range(1, q + 1))
    S = X @ P
    return C + 1j * S, (Y, P)


def sym2(W, V, a):
    """a (W V + V W): what one two-product symmetric-output launch of rgemm.hip computes."""
    return a * (W @ V + V @ W)


def backward(X, q, saved, G):
    """Xbar for dL = Re tr(G^H dU) (engine.hip real_backward): 2q two-product launches."""
    cc, sc = coeffs(q)
    Y, P = saved
    Cb = 0.5 * (G.real + G.real.T)
    Sb = 0.5 * (G.imag + G.imag.T)
    Xb = sym2(Sb, P, 0.5)
    Pb = sym2(Sb, X, 0.5)
    Yb = [None] + [cc[j] * Cb + sc[j] * Pb for j in range(1, q + 1)]
    for j in range(q, 2, -1):
        Yb[1] = Yb[1] + sym2(Yb[j], Y[j - 1], 0.5)
        Yb[j - 1] = Yb[j - 1] + sym2(Yb[j], Y[1], 0.5)
    if q >= 2:
        Yb[1] = Yb[1] + sym2(Yb[2], Y[1], 1.0)
    return Xb + sym2(Yb[1], X, 1.0)


def forward_squared(X, q, s):
    """cos/sin at X / 2^s, then s squarings in the real algebra: (C, S) <- (C C - S S, 2 C S)."""
    U0, saved = forward(X / 2.0 ** s, q)
    C, S = U0.real, U0.imag
    levels = [(C, S)]
    for _ in range(s):
        C, S = C @ C - S @ S, 2.0 * (C @ S)
        levels.append((C, S))
    return C + 1j * S, (saved, levels)


def backward_squared(X, q, s, saved, G):
    """Xbar through the squarings (4 two-product launches per level) and the polynomial; d(X / 2^s) = dX / 2^s."""
    poly_saved, levels = saved
    Cb = 0.5 * (G.real + G.real.T)
    Sb = 0.5 * (G.imag + G.imag.T)
    for i in range(s, 0, -1):
        C, S = levels[i - 1]
        Cb, Sb = sym2(Cb, C, 1.0) + sym2(Sb, S, 1.0), sym2(Cb, S, -1.0) + sym2(Sb, C, 1.0)
    return backward(X / 2.0 ** s, q, poly_saved, Cb + 1j * Sb) / 2.0 ** s


def random_symmetric(d, norm, seed):
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((d, d))
    A = A + A.T
    return A * (norm / np.linalg.norm(A, 1))


@pytest.mark.parametrize("q,theta", [(1, 1e-4), (2, 3e-3), (3, 0.024), (4, 0.08), (5, 0.2), (6, 0.4)])
def test_cos_sin_scheme_matches_expm_within_the_taylor_remainder(q, theta):
    X = random_symmetric(24, theta, q)
    U, _ = forward(X, q)
    ref = scipy.linalg.expm(1j * X)
    rem = theta ** (2 * q + 2) / math.factorial(2 * q + 2)
    assert np.max(np.abs(U - ref)) < rem + 5e-16
    assert rem <= 2.0 ** -53 * 1.01 or q < 3      # these (q, theta) pairs are the fp64 selection boundary of the engine
    assert np.max(np.abs(U @ U.conj().T - np.eye(24))) < 2 * rem + 1e-14


@pytest.mark.parametrize("q", [1, 2, 3, 4, 5, 6])
def test_reverse_mode_is_the_exact_adjoint_of_the_scheme(q):
    """<Xbar, E> = Re tr(G^H dU[E]) for symmetric directions E, with dU the derivative of the POLYNOMIAL (checked by
    the complex-step-free central difference of `forward`), and = the Frechet derivative of exp up to the remainder."""
    d = 12
    X = random_symmetric(d, 0.3, 10 + q)
    rng = np.random.default_rng(q)
    G = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    U, saved = forward(X, q)
    Xb = backward(X, q, saved, G)
    assert np.max(np.abs(Xb - Xb.T)) < 1e-14
    E = rng.standard_normal((d, d))
    E = E + E.T
    eps = 1e-6
    dU = (forward(X + eps * E, q)[0] - forward(X - eps * E, q)[0]) / (2 * eps)
    lhs = float(np.sum(Xb * E))
    rhs = float(np.real(np.trace(G.conj().T @ dU)))
    assert abs(lhs - rhs) < 1e-8 * max(1.0, abs(rhs))
    if q >= 5:   # remainder below 1e-10 at this norm: also equals the derivative of the true exponential
        L = scipy.linalg.expm_frechet(1j * X, 1j * E, compute_expm=False)
        assert abs(lhs - float(np.real(np.trace(G.conj().T @ L)))) < 1e-9 * max(1.0, abs(rhs))


@pytest.mark.parametrize("theta,s", [(1.5, 2), (6.0, 4), (40.0, 7)])
def test_squarings_in_the_real_algebra(theta, s):
    d, q = 10, 6
    X = random_symmetric(d, theta, int(theta))
    U, saved = forward_squared(X, q, s)
    ref = scipy.linalg.expm(1j * X)
    assert np.max(np.abs(U - ref)) < 1e-13 * 2 ** s
    rng = np.random.default_rng(s)
    G = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    Xb = backward_squared(X, q, s, saved, G)
    E = rng.standard_normal((d, d))
    E = E + E.T
    L = scipy.linalg.expm_frechet(1j * X, 1j * E, compute_expm=False)
    lhs, rhs = float(np.sum(Xb * E)), float(np.real(np.trace(G.conj().T @ L)))
    assert abs(lhs - rhs) < 1e-11 * 2 ** s * max(1.0, abs(rhs))


def test_gradient_of_a_two_step_chain_matches_the_oracle(oracle):
    """The whole construction on a tiny chain: dL/du_k[t] = <Xbar_t, -dt H_k> against oracle.energy_grad
    (scipy expm + expm_frechet on the complex generator, the reference's arithmetic)."""
    from diffquantum_amd import problems
    n, K, B, S = 3, 2, 4, 2
    H0, Hs, M = problems.random_real_symmetric_problem(n, K, seed=5)
    u = np.random.default_rng(0).uniform(-1, 1, (S, K))
    dt = 0.02
    psi0 = problems.random_states(1 << n, B, seed=1)
    q = 6
    Xs = [-dt * (H0.real + sum(u[t, k] * Hs[k].real for k in range(K))) for t in range(S)]
    fw = [forward(X, q) for X in Xs]
    psis = [psi0]
    for U, _ in fw:
        psis.append(U @ psis[-1])
    lam = 2.0 * (M @ psis[-1])
    grad = np.zeros((S, K))
    for t in range(S - 1, -1, -1):
        G = lam @ psis[t].conj().T
        Xb = backward(Xs[t], q, fw[t][1], G)
        for k in range(K):
            grad[t, k] = float(np.sum(Xb * (-dt * Hs[k].real)))
        lam = fw[t][0].conj().T @ lam
    loss = float(np.real(np.sum(psis[-1].conj() * (M @ psis[-1]))))
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    assert abs(loss - rl) < 1e-12 and np.max(np.abs(grad - rg)) < 1e-12 and np.max(np.abs(psis[-1] - rf)) < 1e-13


# ---- the time-parallel operator backward sweep (engine.hip tp_sweep), restated in numpy ---------------------------------
def tp_sweep_model(U, W):
    """All G_j = W_{j+1} U_j of a chunk and the chunk-start W from suffix products Q_j = U_{n-1} ... U_j, built with the
    engine's two-level scan: blocks of ~sqrt(n) aligned to the chunk's end, local suffixes, chained block totals."""
    n, d = len(U), U[0].shape[0]
    b = max(1, min(int(math.ceil(math.sqrt(n))), n))
    nblk = (n + b - 1) // b
    len_first = n - (nblk - 1) * b
    L = [None] * n
    for k in range(nblk):                       # 1. block ends
        e = n - 1 - k * b
        L[e] = U[e]
    for r in range(1, b):                       # 2. local suffixes, batched over the blocks longer than r
        nb_r = nblk if r < len_first else nblk - 1
        for k in range(nb_r):
            j = n - 1 - k * b - r
            L[j] = L[j + 1] @ U[j]
    start = lambda k: max(0, n - (k + 1) * b)
    R = [L[start(0)]]                           # 3. block suffix chain
    for k in range(1, nblk):
        R.append(R[k - 1] @ L[start(k)])
    Q = [None] * (n + 1)                        # 4. Q_j
    for j in range(start(0), n):
        Q[j] = L[j]
    for k in range(1, nblk):
        for j in range(start(k), n - k * b):
            Q[j] = R[k - 1] @ L[j]
    Q[n] = np.eye(d)
    Z = [W @ Q[j] for j in range(n)]            # 5-7
    G = [Q[j + 1].conj().T @ Z[j] for j in range(n)]
    return G, Q[0].conj().T @ Z[0]


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 7, 9, 11, 16, 17, 30])
def test_time_parallel_sweep_equals_the_sequential_recursion(n):
    d = 5
    rng = np.random.default_rng(n)
    U = [scipy.linalg.expm(1j * random_symmetric(d, 0.7, 100 * n + t) + 0.3 * (lambda a: a - a.T)(rng.standard_normal((d, d))))
         for t in range(n)]                      # unitary: exp of an anti-Hermitian matrix
    W = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    W = W - W.conj().T
    G, Wn = tp_sweep_model(U, W)
    Ws = W
    for t in range(n - 1, -1, -1):               # G_t = W_{t+1} U_t, W_t = U_t^H G_t
        Gt = Ws @ U[t]
        assert np.max(np.abs(G[t] - Gt)) < 1e-13
        Ws = U[t].conj().T @ Gt
    assert np.max(np.abs(Wn - Ws)) < 1e-13
    assert np.max(np.abs(Wn + Wn.conj().T)) < 1e-13   # stays anti-Hermitian
