"""GPU parity of the real-symmetric fast path: when H0 and all H_k are real, the step propagator
exp(-i dt H) (diffqc.cc:198 `dH.exp()`, sim_plain.py:145 `scipy.linalg.expm(dH)`) is evaluated as cos X + i sin X
with real symmetric-output MFMA products (rgemm.hip) and differentiated in the same real algebra.
Checked against the oracle (the reference's expm + matmul loop), against the engine's own complex tape on the same
inputs, and against numpy for the real GEMM kernel itself.

Tolerance (BASELINE.json north_star): 1e-10 on final amplitudes and gradients in fp64, 1e-4 in fp32.
"""
import numpy as np
import pytest

from diffquantum_amd import problems

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _engine(H0, Hs, M=None, dtype="c128"):
    from diffquantum_amd.engine import Engine
    e = Engine(len(H0), len(Hs), dtype=dtype)
    e.set_hamiltonians(H0, Hs)
    if M is not None:
        e.set_observable(M)
    e.set_option("real_min_dim", 1)   # auto mode only takes the real path from dim 128 up (launch-bound below)
    e.set_option("small_dim", 0)      # state form for batch < dim (the default sends dim <= 512 to the operator form)
    return e


# ---- the real GEMM kernel -------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 64, 32])
@pytest.mark.parametrize("M,N,K,nb", [(1, 1, 1, 1), (16, 16, 4, 2), (33, 17, 9, 3), (128, 128, 128, 1), (130, 257, 77, 2),
                                      (256, 256, 256, 1), (300, 140, 513, 1)])
def test_rgemm_matches_numpy(cfg, M, N, K, nb):
    from diffquantum_amd.engine import Engine
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    with Engine(4, 0) as e:
        e.set_option("rgemm_cfg", cfg)
        for opA in (0, 1):
            for opB in (0, 1):
                A = rng.standard_normal((nb, K, M) if opA else (nb, M, K))
                B = rng.standard_normal((nb, N, K) if opB else (nb, K, N))
                C0 = rng.standard_normal((nb, M, N))
                got = e.rgemm(A, B, opA, opB, alpha=0.7, beta=-0.3, C_in=C0)
                ref = 0.7 * np.matmul(A.transpose(0, 2, 1) if opA else A, B.transpose(0, 2, 1) if opB else B) - 0.3 * C0
                scale = np.abs(A).max() * np.abs(B).max() * K + 1.0
                assert np.max(np.abs(got - ref)) < 1e-14 * scale, (opA, opB)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 64, 32])
@pytest.mark.parametrize("d,nb", [(5, 2), (64, 3), (128, 1), (200, 2), (384, 1), (1024, 1)])
def test_rgemm_symmetric_output_form(cfg, d, nb):
    """Products of commuting symmetric matrices (two polynomials of one symmetric X): upper tiles + mirror."""
    from diffquantum_amd.engine import Engine
    rng = np.random.default_rng(d)
    X = rng.standard_normal((nb, d, d))
    X = (X + X.transpose(0, 2, 1)) / (2 * np.sqrt(d))
    A = X @ X + 0.3 * X
    B = 2.0 * np.eye(d) - X + 0.5 * (X @ X)
    A = (A + A.transpose(0, 2, 1)) / 2
    B = (B + B.transpose(0, 2, 1)) / 2
    with Engine(4, 0) as e:
        e.set_option("rgemm_cfg", cfg)
        got = e.rgemm(A, B, sym=True)
        full = e.rgemm(A, B, 1, 0)
    ref = A @ B
    assert np.max(np.abs(got - ref)) < 1e-13 * max(1.0, np.abs(ref).max())
    assert np.array_equal(got, got.transpose(0, 2, 1))          # exactly symmetric by construction
    iu = np.triu_indices(d)
    # the tiles that are computed (not mirrored) carry the same bits as the full product
    assert np.max(np.abs(got[:, iu[0], iu[1]] - full[:, iu[0], iu[1]])) < 1e-13 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("d", [4096, 3970])
@pytest.mark.parametrize("cfg", [1, 2, 3])
def test_rgemm_symmetric_form_tail_split_at_full_dimension(cfg, d):
    """d = 4096: 528 triangle tiles on 512 workgroup slots -- the 16 tail tiles are split along K and summed by the
    fix-up kernel (fixed order). Same result as numpy, exactly symmetric, and reproducible bit for bit."""
    from diffquantum_amd.engine import Engine
    rng = np.random.default_rng(d)
    X = rng.standard_normal((d, d))
    X = (X + X.T) / (2 * np.sqrt(d))
    A = X @ X
    A = (A + A.T) / 2
    with Engine(4, 0) as e:
        e.set_option("rgemm_cfg", cfg)
        got = e.rgemm(A, X, sym=True)
        again = e.rgemm(A, X, sym=True)
    ref = A @ X
    assert np.max(np.abs(got - ref)) < 1e-13 * max(1.0, np.abs(ref).max())
    assert np.array_equal(got, got.T) and np.array_equal(got, again)


@pytest.mark.parametrize("cfg", [0, 1, 2, 3, 64, 32])
@pytest.mark.parametrize("d,nb", [(7, 2), (64, 2), (130, 1), (256, 2), (512, 1)])
def test_rgemm_two_product_form(cfg, d, nb):
    """C = alpha (A B + B A) + beta C for symmetric, NON-commuting A, B: the reverse mode's sym(Wbar V) products."""
    from diffquantum_amd.engine import Engine
    rng = np.random.default_rng(d + cfg)
    sym = lambda Z: (Z + Z.transpose(0, 2, 1)) / 2
    A, B, C0 = (sym(rng.standard_normal((nb, d, d))) for _ in range(3))
    with Engine(4, 0) as e:
        e.set_option("rgemm_cfg", cfg)
        got = e.rgemm(A, B, alpha=0.5, beta=-0.7, C_in=C0, sym=2)
    ref = 0.5 * (A @ B + B @ A) - 0.7 * C0
    assert np.max(np.abs(got - ref)) < 1e-13 * d
    assert np.array_equal(got, got.transpose(0, 2, 1))


@pytest.mark.parametrize("glds", [1, 0])
def test_rgemm_two_product_form_at_full_dimension(glds):
    """d = 4096 (528 tiles of double K on 512 slots: K-split tail + fix-up), LDS-DMA staging on and off."""
    from diffquantum_amd.engine import Engine
    d = 4096
    rng = np.random.default_rng(3)
    A = rng.standard_normal((d, d)); A = (A + A.T) / 2
    B = rng.standard_normal((d, d)); B = (B + B.T) / 2
    with Engine(4, 0) as e:
        e.set_option("rgemm_glds", glds)
        got = e.rgemm(A, B, alpha=0.5, sym=2)
        full = e.rgemm(A, B, 1, 0)
    ref = 0.5 * (full + full.T)
    assert np.max(np.abs(got - ref)) < 1e-13 * d
    assert np.max(np.abs(full - A @ B)) < 1e-13 * d
    assert np.array_equal(got, got.T)


def test_rgemm_fp32_and_errors():
    from diffquantum_amd.engine import Engine
    rng = np.random.default_rng(0)
    A = rng.standard_normal((150, 90)).astype(np.float32)
    B = rng.standard_normal((90, 200)).astype(np.float32)
    with Engine(4, 0, dtype="c64") as e:
        got = e.rgemm(A, B)
        assert got.dtype == np.float32
        assert np.max(np.abs(got - A.astype(np.float64) @ B.astype(np.float64))) < 2e-4
        with pytest.raises(ValueError):
            e.rgemm(A, B[:50])
        with pytest.raises(ValueError):
            e.rgemm(A, B, sym=True)          # not square
        with pytest.raises(ValueError):
            e.set_option("rgemm_cfg", 7)
        with pytest.raises(ValueError):
            e.set_option("real_path", 9)


def _dt(H0, Hs, umax, theta):
    """Step size that puts the engine's generator-norm bound dt (|H0|_1 + sum_k |u_k| |H_k|_1) at `theta`: the cos/sin
    polynomials (size <= 6, degree <= 13, no squarings) cover theta <~ 0.4 at the fp64 tolerance; larger steps take the
    complex tape with scaling and squaring (test_complex_or_large_step_hamiltonians_take_the_complex_tape)."""
    n1 = lambda A: float(np.max(np.sum(np.abs(A), axis=0)))
    return theta / (n1(H0) + umax * sum(n1(h) for h in Hs))


# ---- the chain on the real path -------------------------------------------------------------------------------
@pytest.mark.parametrize("n,K,B,S,theta", [(1, 1, 1, 10, 0.05), (2, 2, 5, 7, 0.4), (3, 3, 64, 30, 0.02),
                                           (4, 4, 13, 12, -0.2), (5, 2, 33, 9, 0.3), (2, 0, 3, 4, 0.3), (6, 3, 20, 5, 0.01)])
def test_real_path_matches_oracle_and_complex_tape(oracle, n, K, B, S, theta):
    H0, Hs, M = problems.random_real_symmetric_problem(n, K, seed=n * 10 + K)
    u = np.random.default_rng(S).uniform(-2, 2, (S, K))
    dt = _dt(H0, Hs, 2.0, theta)
    psi0 = problems.random_states(1 << n, B, seed=5)
    with _engine(H0, Hs, M) as e:
        assert e.info("real_hamiltonians") == 1
        fwd = e.propagate(u, dt, psi0)
        assert e.info("real_path") >= 1 and e.info("degree") == 2 * e.info("real_path") + 1
        loss, grad, fin, en = e.energy_grad(u, dt, psi0)
        assert e.info("real_path") >= 1
        e.set_option("real_path", 0)
        fwd_c = e.propagate(u, dt, psi0)
        loss_c, grad_c, fin_c, en_c = e.energy_grad(u, dt, psi0)
        assert e.info("real_path") == 0
    ref = oracle.propagate(H0, list(Hs), u, dt, psi0)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    assert np.max(np.abs(fwd - ref)) < TOL and np.max(np.abs(fin - rf)) < TOL
    assert abs(loss - rl) < TOL * B and np.max(np.abs(grad - rg), initial=0.0) < TOL * B
    assert np.max(np.abs(en - oracle.energies(M, rf))) < TOL
    # the two evaluation schemes of the same exponential agree far below the parity tolerance
    assert np.max(np.abs(fwd - fwd_c)) < 1e-12 and np.max(np.abs(grad - grad_c), initial=0.0) < 1e-12 * max(B, 1)
    assert np.max(np.abs(np.linalg.norm(fwd, axis=0) - 1.0)) < 1e-12


@pytest.mark.parametrize("q", [1, 2, 3, 4, 5, 6])
def test_forced_polynomial_sizes(oracle, q):
    """Every cos/sin polynomial size: forward accuracy follows the Taylor remainder, and the gradient is the EXACT
    derivative of the polynomial actually used (checked by central differences of the same forward)."""
    n, K, B, S = 3, 2, 6, 4
    H0, Hs, M = problems.random_real_symmetric_problem(n, K, seed=q)
    u = np.random.default_rng(q).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(1 << n, B, seed=1)
    dt = _dt(H0, Hs, 1.0, 0.5)
    with _engine(H0, Hs, M) as e:
        e.set_option("real_path", q)
        loss, grad, fin, _ = e.energy_grad(u, dt, psi0)
        assert e.info("real_path") == q and e.info("products") == q + 1
        theta = e.info("theta")
        v = np.random.default_rng(0).standard_normal(u.shape)
        eps = 1e-5
        lp = e.energy(e.propagate(u + eps * v, dt, psi0))[1]
        lm = e.energy(e.propagate(u - eps * v, dt, psi0))[1]
    fd, an = (lp - lm) / (2 * eps), float(np.sum(grad * v))
    assert abs(fd - an) < 2e-9 * max(1.0, abs(an)), (fd, an)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    rem = S * theta ** (2 * q + 2) / np.prod(np.arange(1, 2 * q + 3, dtype=float))
    assert np.max(np.abs(fin - rf)) < 4 * rem + 1e-12


def test_real_path_time_chunks_and_checkpoints(oracle):
    n, K, B, S = 3, 3, 9, 23
    H0, Hs, M = problems.random_real_symmetric_problem(n, K, seed=2)
    u = np.random.default_rng(1).uniform(-2, 2, (S, K))
    dt = _dt(H0, Hs, 2.0, 0.25)
    psi0 = problems.random_states(8, B, seed=3)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M) as e:
        for chunk_steps, ck_states in ((S, 0), (5, 0), (1, 0), (7, 3), (4, 1)):
            e.set_option("chunk_bytes", 24 * 8 * 8 * 16 * chunk_steps)
            e.set_option("checkpoint_bytes", 8 * B * 16 * ck_states)
            loss, grad, fin, _ = e.energy_grad(u, dt, psi0)
            assert e.info("real_path") >= 1
            assert abs(loss - rl) < TOL * B and np.max(np.abs(grad - rg)) < TOL * B, (chunk_steps, ck_states)
            assert np.max(np.abs(fin - rf)) < TOL


def test_real_path_wide_batch_operator_form(oracle):
    """B > d: the propagators (built on the real path) are composed / swept in operator form."""
    n, K, B, S = 3, 2, 40, 11
    H0, Hs, M = problems.random_real_symmetric_problem(n, K, seed=4)
    u = np.random.default_rng(2).uniform(-1.5, 1.5, (S, K))
    dt = _dt(H0, Hs, 1.5, 0.3)
    psi0 = problems.random_states(8, B, seed=3)
    with _engine(H0, Hs, M) as e:
        fwd = e.propagate(u, dt, psi0)
        loss, grad, fin, _ = e.energy_grad(u, dt, psi0)
        assert e.info("composed") == 1 and e.info("real_path") >= 1
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    assert np.max(np.abs(fwd - rf)) < TOL and np.max(np.abs(fin - rf)) < TOL
    assert abs(loss - rl) < TOL * B and np.max(np.abs(grad - rg)) < TOL * B


def test_real_path_fidelity_objective(oracle):
    n, K, B, S = 3, 2, 5, 9
    H0, Hs, _ = problems.random_real_symmetric_problem(n, K, seed=6)
    u = np.random.default_rng(3).uniform(-1, 1, (S, K))
    dt = _dt(H0, Hs, 1.0, 0.3)
    psi0 = problems.random_states(8, B, seed=1)
    tgt = problems.random_states(8, B, seed=2)
    with _engine(H0, Hs) as e:
        l1, g1, f1, fid1 = e.fidelity_grad(u, dt, psi0, tgt)
        assert e.info("real_path") >= 1
        e.set_option("real_path", 0)
        l0, g0, f0, fid0 = e.fidelity_grad(u, dt, psi0, tgt)
    rl, rg, rf, rfid = oracle.fidelity_grad(H0, list(Hs), u, dt, psi0, tgt)
    assert abs(l1 - rl) < TOL * B and np.max(np.abs(g1 - rg)) < TOL * B and np.max(np.abs(fid1 - rfid)) < TOL
    assert np.max(np.abs(g1 - g0)) < 1e-12 * B


def test_complex_or_large_step_hamiltonians_take_the_complex_tape(oracle):
    H0, Hs, M = problems.random_hermitian_problem(3, 2, seed=3)          # complex Hermitian
    u = np.ones((3, 2))
    psi0 = problems.random_states(8, 2, seed=1)
    with _engine(H0, Hs, M) as e:
        assert e.info("real_hamiltonians") == 0
        e.energy_grad(u, 0.1, psi0)
        assert e.info("real_path") == 0
        e.set_option("real_path", 3)                                        # cannot be forced on complex H
        e.energy_grad(u, 0.1, psi0)
        assert e.info("real_path") == 0
    H0, Hs, M = problems.random_real_symmetric_problem(3, 2, seed=3)
    with _engine(H0, Hs, M) as e:
        e.set_option("force_degree", 9)                                     # addresses the complex tape
        e.propagate(u, 0.1, psi0)
        assert e.info("real_path") == 0 and e.info("degree") == 9
        e.set_option("force_degree", 0)
        e.set_option("force_squarings", 2)                                  # so does this
        e.propagate(u, 0.1, psi0)
        assert e.info("real_path") == 0 and e.info("squarings") == 2


@pytest.mark.parametrize("n,K,B,S,theta", [(3, 2, 2, 3, 30.0), (3, 2, 9, 5, 2.0), (4, 3, 5, 4, 0.9), (2, 1, 3, 6, 200.0),
                                           (5, 2, 40, 3, 7.0)])
def test_large_steps_square_in_the_real_algebra(oracle, n, K, B, S, theta):
    """theta beyond the polynomials' range: cos/sin at X / 2^s, then s squarings (C, S) <- (C^2 - S^2, 2 C S) with real
    symmetric-output products, and their reverse mode -- against the oracle and the complex tape."""
    H0, Hs, M = problems.random_real_symmetric_problem(n, K, seed=n + K)
    u = np.random.default_rng(S).uniform(-2, 2, (S, K))
    dt = _dt(H0, Hs, 2.0, theta)
    psi0 = problems.random_states(1 << n, B, seed=1)
    with _engine(H0, Hs, M) as e:
        fwd = e.propagate(u, dt, psi0)
        assert e.info("real_path") >= 1 and e.info("squarings") >= 1
        assert e.info("products") == e.info("real_path") + 1 + 3 * e.info("squarings")
        loss, grad, fin, _ = e.energy_grad(u, dt, psi0)
        e.set_option("real_path", 0)
        loss_c, grad_c, fin_c, _ = e.energy_grad(u, dt, psi0)
        assert e.info("real_path") == 0
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    scale = max(1.0, theta)   # d exp / dA grows with the step: same relative budget
    assert np.max(np.abs(fwd - rf)) < TOL * scale and np.max(np.abs(fin - rf)) < TOL * scale
    assert abs(loss - rl) < TOL * B * scale and np.max(np.abs(grad - rg)) < TOL * B * scale
    assert np.max(np.abs(grad - grad_c)) < TOL * B * scale


def test_real_path_d256_through_the_128_tile(oracle):
    n, K, B, S = 8, 3, 24, 3
    H0, Hs, M = problems.random_real_symmetric_problem(n, K, seed=8)
    u = np.random.default_rng(5).uniform(-2, 2, (S, K))
    dt = _dt(H0, Hs, 2.0, 0.2)
    psi0 = problems.random_states(256, B, seed=2)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M) as e:
        for cfg in (0, 1, 2, 3, 64):
            e.set_option("rgemm_cfg", cfg)
            loss, grad, fin, _ = e.energy_grad(u, dt, psi0)
            assert e.info("real_path") >= 1
            assert np.max(np.abs(fin - rf)) < TOL and abs(loss - rl) < TOL * B and np.max(np.abs(grad - rg)) < TOL * B, cfg


def test_real_path_fp32(oracle):
    n, K, B, S = 4, 3, 10, 12
    H0, Hs, M = problems.random_real_symmetric_problem(n, K, seed=9)
    u = np.random.default_rng(6).uniform(-2, 2, (S, K))
    dt = _dt(H0, Hs, 2.0, 0.3)
    psi0 = problems.random_states(16, B, seed=2)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, dtype="c64") as e:
        loss, grad, fin, _ = e.energy_grad(u, dt, psi0)
        assert e.info("real_path") >= 1
    assert np.max(np.abs(fin - rf)) < 1e-4 and abs(loss - rl) < 1e-4 * B and np.max(np.abs(grad - rg)) < 1e-4 * B


def test_reference_demo_hamiltonians(oracle):
    """The MaxCut problem of the reference demo (ZZ drift, ZZ + X controls, demo_maxcut.py:36-79) is real. With the
    demo's own step (per_step = 10: dt = 1/15, theta ~ 2) the real path squares three times; with a finer grid it needs
    no squaring. Both match the oracle. (In auto mode d = 16 stays on the complex tape: launch-bound, real_min_dim.)"""
    pr = problems.maxcut_problem()
    H0, Hs, M = pr["H0"].astype(complex), pr["Hs"].astype(complex), pr["H_cost"].astype(complex)
    T = pr["T"]
    psi0 = pr["psi0"].astype(complex).reshape(-1, 1)
    with _engine(H0, Hs, M) as e:
        assert e.info("real_hamiltonians") == 1
        for S, squared in ((30, True), (400, False)):
            u = problems.pulse_table(len(Hs), S, T, seed=7)
            loss, grad, fin, _ = e.energy_grad(u, T / S, psi0)
            assert e.info("real_path") >= 1 and (e.info("squarings") > 0) == squared
            rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, T / S, psi0)
            assert np.max(np.abs(fin - rf)) < TOL and abs(loss - rl) < TOL and np.max(np.abs(grad - rg)) < TOL


def test_auto_mode_keeps_small_dimensions_on_the_complex_tape():
    from diffquantum_amd.engine import Engine
    H0, Hs, M = problems.random_real_symmetric_problem(3, 2, seed=1)
    psi0 = problems.random_states(8, 2, seed=1)
    with Engine(8, 2) as e:
        e.set_hamiltonians(H0, Hs)
        e.propagate(0.01 * np.ones((3, 2)), 0.01, psi0)
        assert e.info("real_hamiltonians") == 1 and e.info("real_path") == 0


@pytest.mark.parametrize("B", [3, 8, 40])
def test_propagator_cache_reproduces_the_rebuild_bit_for_bit(oracle, B):
    """The backward sweep takes the real path's intermediates (X, Y_j, P, U) of the cached steps from HBM instead of
    rebuilding them: same bits, for a full, a partial and no cache, in state form (B = 3) and operator form."""
    n, K, S = 3, 2, 21
    H0, Hs, M = problems.random_real_symmetric_problem(n, K, seed=B)
    u = np.random.default_rng(2).uniform(-2, 2, (S, K))
    dt = _dt(H0, Hs, 2.0, 0.2)
    psi0 = problems.random_states(8, B, seed=4)
    per_step = None   # (q + 3) real 8 x 8 matrices per step; q is read back from the engine below
    out = {}
    with _engine(H0, Hs, M) as e:
        e.set_option("chunk_bytes", 24 * 64 * 16 * 4)       # chunks of 4 steps
        e.propagate(u, dt, psi0)
        per_step = int(e.info("real_path") + 3) * 64 * 8
        for name, cb, expect in (("auto", 0, S), ("off", -1, 0), ("partial", per_step * 10, 8), ("tiny", 1, 0)):
            e.set_option("cache_bytes", cb)
            out[name] = e.energy_grad(u, dt, psi0)
            assert e.info("real_path") >= 1 and e.info("cached_steps") == expect, (name, e.info("cached_steps"))
        e.set_option("cache_bytes", 0)
        e.set_option("plan_steps", 10 * S)                  # a slice of a 10x longer chain caches its share only
        e.set_option("cache_bytes", per_step * 10 * S // 2)
        out["slice"] = e.energy_grad(u, dt, psi0)
        assert e.info("cached_steps") == 8                  # share 1/2 -> 10 steps -> whole chunks: 8
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    for name, (l, g, f, en) in out.items():
        assert np.array_equal(g, out["off"][1]) and l == out["off"][0], name
    assert abs(out["auto"][0] - rl) < TOL * B and np.max(np.abs(out["auto"][1] - rg)) < TOL * B
