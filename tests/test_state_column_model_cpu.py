"""The state-column recurrences (tests/np_state_column.py: the numpy twin of csrc/statecol.hip) against the oracle: forward
amplitudes, loss and dL/du at 1e-10 for real Pauli sums (Ising + X_k; reference problems: demo_maxcut.py:36-79), complex ones
(X and Y drives) and Z/ZZ-type controls with diagonal entries."""
import math

import numpy as np
import pytest

from diffquantum_amd import problems
from tests import np_state_column as nsc


def _xy(n, seed, n_x, n_y):
    d = 1 << n
    H0 = np.diag(problems.ising_diagonal(n, seed)).astype(np.complex128)
    idx = np.arange(d)
    Hs = np.zeros((n_x + n_y, d, d), dtype=np.complex128)
    for q in range(n_x):
        Hs[q, idx, idx ^ (1 << (n - 1 - q))] = 1.0
    for j in range(n_y):
        e = 1 << (n - 1 - (n - n_y + j))
        bit = ((idx & e) != 0).astype(np.float64)
        Hs[n_x + j, idx, idx ^ e] = 1j * (2.0 * bit - 1.0)
    return H0, Hs, H0.copy()


def _cases():
    n = 5
    yield "ising_x", problems.random_ising(n, 3)
    yield "xy_drives", _xy(n, 4, 3, 2)
    p = problems.qaoa_ring(4)
    yield "qaoa_ring_zz_controls", (p["H0"].astype(np.complex128), p["Hs"].astype(np.complex128), p["H_cost"].astype(np.complex128))
    # two-body X_a X_b and X_a Z_b terms (signs along an XOR-diagonal, shared offsets) and a Y Z term
    H0, Hs, M = problems.random_ising(4, 5)
    X = np.array([[0, 1], [1, 0]], dtype=complex); Y = np.array([[0, -1j], [1j, 0]]); Z = np.diag([1.0 + 0j, -1.0]); I = np.eye(2)
    kr = lambda *ops: problems.kron_chain(list(ops))
    Hs = np.stack([kr(X, X, I, I) + 0.5 * kr(X, Z, I, I), kr(I, Y, Z, I), kr(I, I, Z, Z) + kr(I, I, I, X)]).astype(np.complex128)
    yield "two_body_mixed", (H0, Hs, M)


@pytest.mark.parametrize("name,prob", list(_cases()))
def test_state_column_model_matches_oracle(oracle, name, prob):
    H0, Hs, M = prob
    K, d = len(Hs), H0.shape[0]
    S, B, dt = 5, 3, 0.02
    u = np.random.default_rng(7).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(d, B, seed=1)
    tab = nsc.table_from_dense(H0, Hs)
    assert tab is not None and len(tab["xoff"]) >= 1
    loss, grad, fin = nsc.energy_grad(H0, Hs, M, u, dt, psi0)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    assert np.max(np.abs(fin - rf)) < 1e-12
    assert abs(loss - rl) < 1e-11
    assert np.max(np.abs(grad - rg)) < 1e-11


def test_table_rejects_what_is_not_a_pauli_sum():
    H0, Hs, _ = problems.random_hermitian_problem(3, 2, seed=0)
    assert nsc.table_from_dense(H0, Hs) is None


@pytest.mark.parametrize("name,prob", list(_cases()))
def test_quadrature_form_of_the_backward_step_matches_oracle(oracle, name, prob):
    """The quadrature form (4-point Gauss-Lobatto rule on the step integral of dexp: csrc/statecol_bwdq.hip) against the oracle's exact
    Frechet derivative: at the step size the engine takes it for (theta <= 0.05) to rounding; the 5-point rule at theta ~ 0.2; and the
    error of the 4-point rule grows as the bound says ((2 theta)^6) when the step is pushed beyond that."""
    H0, Hs, M = prob
    K, d = len(Hs), H0.shape[0]
    S, B = 4, 2
    u = np.random.default_rng(8).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(d, B, seed=3)
    n1 = lambda A: np.abs(A).sum(axis=0).max()
    scale = n1(H0) + sum(n1(h) for h in Hs)
    errs = {}
    for theta, pts in ((0.04, 4), (0.2, 5), (0.2, 4)):
        dt = theta / scale
        rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
        loss, grad, fin = nsc.energy_grad(H0, Hs, M, u, dt, psi0, quad_points=pts)
        assert np.max(np.abs(fin - rf)) < 1e-12 and abs(loss - rl) < 1e-11
        errs[(theta, pts)] = np.max(np.abs(grad - rg)) / np.max(np.abs(rg))
    assert errs[(0.04, 4)] < 1e-13 and errs[(0.2, 5)] < 1e-12, errs
    assert errs[(0.2, 4)] < 6.6e-7 * 0.4 ** 6 * 50, errs           # (the bound of the rule, with room for the norms it is relative to)


@pytest.mark.parametrize("name,prob", list(_cases()))
def test_economized_step_polynomial_matches_oracle(oracle, name, prob):
    """The Chebyshev-economized step polynomial (csrc/statecol.hip xs_poly_make; the kernels' default): its degree is below Taylor's
    at the same tolerance, and the chain + both forms of the adjoint through it match the oracle (scipy expm / expm_frechet) as the
    Taylor form does -- the Horner levels of the exact form and the node sums of the quadrature form carry the level factors."""
    H0, Hs, M = prob
    K, d = len(Hs), H0.shape[0]
    S, B = 4, 2
    u = np.random.default_rng(9).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(d, B, seed=4)
    n1 = lambda A: np.abs(A).sum(axis=0).max()
    scale = n1(H0) + sum(n1(h) for h in Hs)
    for theta, pts in ((0.03, 0), (0.03, 4), (0.25, 0), (0.25, 5)):
        dt = theta / scale
        th = max(abs(dt) * (n1(H0) + sum(abs(row[k]) * n1(Hs[k]) for k in range(K))) for row in u)
        assert nsc.economized_degree(th) <= nsc.taylor_degree(th) - (1 if theta > 0.1 else 0)
        rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
        loss, grad, fin = nsc.energy_grad(H0, Hs, M, u, dt, psi0, quad_points=pts, economized=True)
        assert np.max(np.abs(fin - rf)) < 1e-12 and abs(loss - rl) < 1e-11
        assert np.max(np.abs(grad - rg)) < 1e-12 * max(1.0, np.max(np.abs(rg))), (theta, pts)


def test_library_step_polynomial_matches_the_model():
    """dq_step_polynomial (host-only entry of the C-ABI, include/diffqc_hip.h): degree and level factors against the numpy
    construction, the sup-norm error on the spectral interval against the tolerance, one to four degrees below Taylor."""
    import ctypes as C
    from math import factorial
    from diffquantum_amd import _lib
    lib = _lib.load()
    for theta in (0.0, 1e-9, 0.0177, 0.0295, 0.05, 0.2, 1.0, 2.0):
        for tol in (2.0 ** -53, 6e-8):
            L, g = C.c_int(), (C.c_double * 25)()
            Lt, gt = C.c_int(), (C.c_double * 25)()
            assert lib.dq_step_polynomial(theta, tol, 1, C.byref(L), g) == 0
            assert lib.dq_step_polynomial(theta, tol, 0, C.byref(Lt), gt) == 0
            assert 1 <= L.value <= Lt.value and all(v == 1.0 for v in gt)
            assert Lt.value == nsc.taylor_degree(theta, tol) if theta > 0 else Lt.value == 1
            if theta == 0:
                continue
            assert L.value == nsc.economized_degree(theta, tol)
            gg = np.array(g[:L.value + 1])
            assert np.max(np.abs(gg - nsc.poly_factors(theta, L.value))) < 4e-16
            z = 1j * np.longdouble(theta) * np.linspace(-1, 1, 2001).astype(np.longdouble)
            p = sum(np.longdouble(gg[m]) * z ** m / factorial(m) for m in range(L.value + 1))
            # (+ the factors' rounding to double: at most 2^-53 sum_m theta^m / m!)
            assert float(np.max(np.abs(p - np.exp(z)))) <= tol * 1.01 + 2.0 ** -53 * np.exp(theta) * 0.5
    L, g = C.c_int(), (C.c_double * 25)()
    assert lib.dq_step_polynomial(0.0295, 2.0 ** -53, 1, C.byref(L), g) == 0 and L.value == 6      # the headline's step: Taylor needs 7
    assert lib.dq_step_polynomial(1.0, 2.0 ** -53, 1, C.byref(L), g) == 0 and L.value == 14        # Taylor: 18
    assert lib.dq_step_polynomial(2.5, 2.0 ** -53, 1, C.byref(L), g) == 0 and L.value == 0         # beyond the kernels' degrees
    assert lib.dq_step_polynomial(-1.0, 1e-9, 1, C.byref(L), g) != 0 and b"theta" in lib.dq_last_error()


@pytest.mark.parametrize("name,prob", list(_cases())[:3])
def test_wide_batch_gradient_on_dim_columns_model(oracle, name, prob):
    """The identity behind engine.hip do_energy_grad_xs_wide (batches wider than the dimension): L = sum_b <psi_b|M|psi_b> =
    tr(M P R P^H), R = Psi0 Psi0^H, and dL/du from the SAME backward steps run on the d columns of (psi_S, lambda_S) = (P R, 2 M P) --
    against the oracle's gradient over the B states (scipy expm / expm_frechet)."""
    H0, Hs, M = prob
    K, d = len(Hs), H0.shape[0]
    S, B, dt = 4, 3 * d + 1, 0.02
    u = np.random.default_rng(11).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(d, B, seed=5)
    rl, rg, _ = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    tab = nsc.table_from_dense(H0, Hs)
    n1 = lambda A: np.abs(A).sum(axis=0).max()
    th = max(abs(dt) * (n1(H0) + sum(abs(row[k]) * n1(Hs[k]) for k in range(K))) for row in u)
    L = nsc.economized_degree(th)
    g = nsc.poly_factors(th, L)
    P = np.eye(d, dtype=np.complex128)
    for t in range(S):                                                   # the chain's product, column by column
        P = np.stack([nsc.forward_step(tab, u[t], dt, P[:, i], L, g) for i in range(d)], axis=1)
    R = psi0 @ psi0.conj().T
    Y, Lam = P @ R, 2.0 * (M @ P)
    loss = 0.5 * float(np.real(np.sum(Lam.conj() * Y)))
    grad = np.zeros((S, K))
    for i in range(d):
        psi, lam = Y[:, i], Lam[:, i]
        for t in range(S - 1, -1, -1):
            psi, lam, gr = nsc.backward_step(tab, u[t], dt, psi, lam, L, K, g)
            grad[t] += gr
    assert abs(loss - rl) < 1e-10 * B
    assert np.max(np.abs(grad - rg)) < 1e-11 * B, np.max(np.abs(grad - rg))
    assert np.max(np.abs(rg)) > 1e-6


def test_library_quadrature_rule_in_both_bases_matches_the_model():
    """dq_quadrature_rule (host-only entry of the C-ABI): the tables of the quadrature adjoint (csrc/statecol.hip xs_quad_make) against
    the numpy construction of tests/np_state_column.py::backward_step_quad -- node basis entry by entry; DIFFERENCE basis (round 5: the
    kernels that keep the rule's second vector in LDS as fp32) by what it must reproduce: the same q_m from (lambda_t, u_1, u_2) as
    from the node values, for random Taylor terms z_m of the sizes a step produces, with |u_1| = O(theta), |u_2| = O(theta^2)."""
    import ctypes as C
    from diffquantum_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(0)
    for theta, econ in ((0.0295, 1), (0.0295, 0), (0.05, 1), (0.011, 1)):
        L = nsc.economized_degree(theta) if econ else nsc.taylor_degree(theta)
        g = nsc.poly_factors(theta, L) if econ else np.ones(L + 1)
        s, wq = nsc.lobatto(4)
        gi = [nsc.poly_factors(theta * s[i], L) if econ else np.ones(L + 1) for i in (1, 2)]
        out = {}
        for basis in (0, 1):
            n, w0 = C.c_int(), C.c_double()
            w1, c1, c2 = (C.c_double * 25)(), (C.c_double * 50)(), (C.c_double * 50)()
            assert lib.dq_quadrature_rule(theta, 2.0 ** -40, L, econ, basis, C.byref(n), C.byref(w0), w1, c1, c2) == 0
            assert n.value == 2
            out[basis] = (w0.value, np.array(w1[:L + 1]), np.array(c1[:2 * (L + 1)]).reshape(-1, 2), np.array(c2[:2 * (L + 1)]).reshape(-1, 2))
        w0, w1, c1, c2 = out[0]
        # node basis: c1[m][i] = g_i[m] s_i^m, c2 = -w_i c1, w1[m] = -w_end g[m], w0 = -w_0 (the sign: <q, dA r> = -<r, dA q>)
        for m in range(L + 1):
            for i in range(2):
                assert abs(c1[m, i] - gi[i][m] * s[i + 1] ** m) < 1e-15 and abs(c2[m, i] + wq[i + 1] * c1[m, i]) < 1e-15
            assert abs(w1[m] + wq[3] * g[m]) < 1e-15
        assert abs(w0 + wq[0]) < 1e-15
        # difference basis: same q_m for any Taylor terms
        v0, v1, d1, d2 = out[1]
        assert abs(v0 - w0) < 1e-16 and np.all(d1[0] == 0.0) and d1[1, 1] == 0.0
        z = [(rng.standard_normal(5) + 1j * rng.standard_normal(5)) * theta ** m / math.factorial(m) for m in range(L + 1)]
        lam_t = sum(g[m] * z[m] for m in range(L + 1))
        nd = [sum(c1[m, i] * z[m] for m in range(L + 1)) for i in range(2)]
        uu = [sum(d1[m, i] * z[m] for m in range(L + 1)) for i in range(2)]
        assert np.max(np.abs(uu[0])) < 1.5 * theta * 3 and np.max(np.abs(uu[1])) < 1.5 * theta ** 2 * 3
        for m in range(L):
            q_nodes = w1[m] * lam_t + c2[m, 0] * nd[0] + c2[m, 1] * nd[1]
            q_diff = v1[m] * lam_t + d2[m, 0] * uu[0] + d2[m, 1] * uu[1]
            assert np.max(np.abs(q_nodes - q_diff)) < 4e-15, (theta, m)       # (rounding: the vectors are O(1))
    n, w0 = C.c_int(), C.c_double()
    w1, c1, c2 = (C.c_double * 25)(), (C.c_double * 50)(), (C.c_double * 50)()
    assert lib.dq_quadrature_rule(0.2, 2.0 ** -40, 9, 1, 0, C.byref(n), C.byref(w0), w1, c1, c2) == 0 and n.value == 0   # beyond the 4-point rule's bound
