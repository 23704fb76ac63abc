"""GPU parity of the HIP building blocks (ZGEMM, exp, exp-VJP) against numpy/scipy.

Tolerances: fp64, north_star says 1e-10 on amplitudes/gradients; the kernels themselves are
held to ~1e-13 relative here so that the chain (thousands of steps) stays inside 1e-10.
"""
import numpy as np
import pytest
import scipy.linalg

pytestmark = pytest.mark.gpu


def _rand(rng, *shape):
    return rng.standard_normal(shape) + 1j * rng.standard_normal(shape)


@pytest.fixture(scope="module")
def eng16():
    from diffquantum_amd.engine import Engine
    e = Engine(16, 2)
    yield e
    e.close()


@pytest.mark.parametrize("tile", [0, 16, 32, 64, 128])   # (16: 32 x 32 tiles on four waves of 16 x 16, what problems of few tiles take by default)
@pytest.mark.parametrize("opA,opB", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(16, 1, 16), (16, 16, 16), (33, 17, 9), (64, 64, 64), (130, 257, 70),
                                   (256, 96, 256), (1, 1, 1), (128, 128, 4), (5, 300, 3)])
def test_zgemm_matches_numpy(eng16, opA, opB, M, N, K, tile):
    rng = np.random.default_rng(M * 1000 + N * 10 + K + opA * 2 + opB)
    A = _rand(rng, *((K, M) if opA else (M, K)))
    B = _rand(rng, *((N, K) if opB else (K, N)))
    C0 = _rand(rng, M, N)
    eng16.set_option("force_tile", tile)
    try:
        got = eng16.zgemm(A, B, opA=opA, opB=opB, alpha=0.75, beta=-0.5, C_in=C0)
    finally:
        eng16.set_option("force_tile", 0)
    oa = A.conj().T if opA else A
    ob = B.conj().T if opB else B
    ref = 0.75 * (oa @ ob) - 0.5 * C0
    scale = np.abs(oa) @ np.abs(ob) + np.abs(C0)
    assert np.max(np.abs(got - ref) / scale) < 5e-15


def test_zgemm_asymmetric_identity(eng16):
    # A = I with an asymmetric B catches a transposed C write (cdna guide section 3)
    B = np.arange(64 * 48, dtype=np.float64).reshape(64, 48) * (1 + 0.5j)
    got = eng16.zgemm(np.eye(64), B)
    assert np.array_equal(got, B.astype(np.complex128))


@pytest.mark.parametrize("opA,opB", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 260, 130), (512, 256, 257)])
def test_zgemm_3m_variant(eng16, opA, opB, M, N, K):
    """The 3M form (three real MFMAs per complex tile step) is normwise-stable, not componentwise:
    error bound ~ eps * K * max|A| * max|B| instead of eps * (|A||B|)_ij."""
    rng = np.random.default_rng(M + N + K + opA * 2 + opB)
    A = _rand(rng, *((K, M) if opA else (M, K)))
    B = _rand(rng, *((N, K) if opB else (K, N)))
    C0 = _rand(rng, M, N)
    eng16.set_option("force_tile", 128)
    eng16.set_option("zgemm_alg", 1)
    try:
        got = eng16.zgemm(A, B, opA=opA, opB=opB, alpha=1.25, beta=0.5, C_in=C0)
    finally:
        eng16.set_option("force_tile", 0)
        eng16.set_option("zgemm_alg", 0)
    oa = A.conj().T if opA else A
    ob = B.conj().T if opB else B
    ref = 1.25 * (oa @ ob) + 0.5 * C0
    assert np.max(np.abs(got - ref)) < 4e-16 * K * np.max(np.abs(A)) * np.max(np.abs(B)) * 4


@pytest.mark.parametrize("tile", [0, 32, 64, 128])
@pytest.mark.parametrize("alg", [0, 1])
@pytest.mark.parametrize("herm", [1, -1])
@pytest.mark.parametrize("d,nb", [(5, 2), (64, 2), (130, 1), (384, 1)])
def test_zgemm_hermitian_output_forms(eng16, tile, alg, herm, d, nb):
    """X^H W X for (anti-)Hermitian W is (anti-)Hermitian for ANY X: upper tiles + (+-conj) mirror, exact diagonal."""
    rng = np.random.default_rng(d + herm)
    X = _rand(rng, nb, d, d)
    W = _rand(rng, nb, d, d)
    W = W + herm * W.conj().transpose(0, 2, 1)
    G = W @ X
    C0 = _rand(rng, nb, d, d)
    C0 = C0 + herm * C0.conj().transpose(0, 2, 1)
    eng16.set_option("force_tile", tile)
    eng16.set_option("zgemm_alg", alg)
    try:
        got = eng16.zgemm(X, G, opA=1, opB=0, alpha=0.5, beta=0.25, C_in=C0, herm=herm)
    finally:
        eng16.set_option("force_tile", 0)
        eng16.set_option("zgemm_alg", 0)
    ref = 0.5 * (X.conj().transpose(0, 2, 1) @ G) + 0.25 * C0
    assert np.max(np.abs(got - ref)) < 1e-13 * d * max(1.0, np.abs(ref).max())
    assert np.array_equal(got, herm * got.conj().transpose(0, 2, 1))     # exactly (anti-)Hermitian, diagonal included


def test_zgemm_hermitian_form_tail_split():
    """d = 3072: 300 triangle tiles on 256 workgroup slots -- the 44 tail tiles are split along K and summed by the fix-up
    kernel in a fixed order (reproducible bits)."""
    from diffquantum_amd.engine import Engine
    d = 3072
    rng = np.random.default_rng(1)
    X = (_rand(rng, d, d) / np.sqrt(d)).astype(np.complex128)
    W = _rand(rng, d, d)
    W = W - W.conj().T
    G = W @ X
    with Engine(4, 0) as e:
        e.set_option("zgemm_alg", 1)
        got = e.zgemm(X, G, opA=1, opB=0, herm=-1)
        again = e.zgemm(X, G, opA=1, opB=0, herm=-1)
        full = e.zgemm(X, G, opA=1, opB=0)
    assert np.max(np.abs(got - full)) < 1e-12 * np.abs(full).max()
    assert np.max(np.abs(full - X.conj().T @ G)) < 1e-12 * np.abs(full).max()
    assert np.array_equal(got, -got.conj().T) and np.array_equal(got, again)


def test_zgemm_batched(eng16):
    rng = np.random.default_rng(3)
    A = _rand(rng, 5, 40, 24)
    B = _rand(rng, 5, 24, 56)
    got = eng16.zgemm(A, B)
    assert np.allclose(got, A @ B, rtol=0, atol=1e-12)


def test_zgemm_beta_zero_ignores_nan_c(eng16):
    rng = np.random.default_rng(4)
    A, B = _rand(rng, 20, 12), _rand(rng, 12, 8)
    C0 = np.full((20, 8), np.nan + 0j)
    got = eng16.zgemm(A, B, beta=0.0, C_in=C0)
    assert np.allclose(got, A @ B, atol=1e-12)


@pytest.mark.parametrize("d", [2, 4, 16, 64, 96])
@pytest.mark.parametrize("norm", [1e-6, 1e-3, 0.1, 1.0, 10.0, 40.0])
def test_expm_matches_scipy(d, norm):
    from diffquantum_amd.engine import Engine
    rng = np.random.default_rng(d + int(norm * 1000))
    H = _rand(rng, 3, d, d)
    H = H + np.conj(np.transpose(H, (0, 2, 1)))
    A = -1j * H
    A *= norm / np.max([np.linalg.norm(a, 1) for a in A])
    with Engine(d, 0) as e:
        U = e.expm(A)
        deg, sq = e.info("degree"), e.info("squarings")
    for z in range(3):
        ref = scipy.linalg.expm(A[z])
        err = np.max(np.abs(U[z] - ref))
        assert err < 2e-13 * max(1.0, norm), (d, norm, deg, sq, err)
        assert np.max(np.abs(U[z].conj().T @ U[z] - np.eye(d))) < 5e-13 * max(1.0, norm)


@pytest.mark.parametrize("degree", [2, 4, 6, 8, 9, 12, 16, 20])
def test_expm_every_polynomial_degree(degree):
    from diffquantum_amd.engine import Engine
    rng = np.random.default_rng(degree)
    d = 24
    H = _rand(rng, d, d)
    A = -1j * (H + H.conj().T)
    A *= 0.5 / np.linalg.norm(A, 1)
    with Engine(d, 0) as e:
        e.set_option("force_degree", degree)
        e.set_option("force_squarings", 6)  # theta/64 ~ 8e-3: every degree >= 4 is exact to rounding
        U = e.expm(A)
        assert e.info("degree") == degree and e.info("squarings") == 6
    tol = 1e-12 if degree >= 4 else 1e-5
    assert np.max(np.abs(U - scipy.linalg.expm(A))) < tol


@pytest.mark.parametrize("d,norm", [(4, 0.05), (8, 0.5), (16, 3.0), (48, 0.15), (8, 20.0)])
def test_expm_vjp_matches_frechet(d, norm, oracle):
    from diffquantum_amd.engine import Engine
    rng = np.random.default_rng(d * 7 + int(norm * 100))
    H = _rand(rng, 2, d, d)
    A = -1j * (H + np.conj(np.transpose(H, (0, 2, 1))))
    A *= norm / np.max([np.linalg.norm(a, 1) for a in A])
    G = _rand(rng, 2, d, d)
    with Engine(d, 0) as e:
        Ab = e.expm_vjp(A, G)
    for z in range(2):
        ref = oracle.expm_vjp(A[z], G[z])
        assert np.max(np.abs(Ab[z] - ref)) < 1e-12 * max(1.0, norm) * np.max(np.abs(G[z])) * d
        # defining property against a random direction
        E = _rand(rng, d, d)
        lhs = np.real(np.vdot(G[z], scipy.linalg.expm_frechet(A[z], E, compute_expm=False)))
        rhs = np.real(np.vdot(Ab[z], E))
        assert abs(lhs - rhs) < 1e-10 * max(1.0, abs(lhs))


# ---- linear-form complex GEMM (csrc/zgemm_lin.hip): 16 waves, LDS-DMA staging, op(A) read k-major -----------------

def _structured(rng, n, kind):
    a = _rand(rng, n, n)
    return {1: a + a.T, 2: a + a.conj().T, 3: a - a.conj().T}[kind]


@pytest.mark.parametrize("case", ["AH_B", "symmetric", "hermitian", "antihermitian", "AH_B_wide", "AH_B_deep_k"])
def test_zgemm_linear_form_matches_numpy(eng16, case):
    """Whole 128-tiles at >= 64 tiles take the linear-form kernel when op(A) = A^H or A carries a symmetry hint
    (opA 2 / 3 / 4 = as stored and complex symmetric / Hermitian / anti-Hermitian); same product as numpy, and the same
    BITS as the general 3M kernel (zgemm_alg = 1), whose arithmetic order it keeps."""
    rng = np.random.default_rng(len(case))
    M, N, K = {"AH_B_wide": (1024, 2048, 1024), "AH_B_deep_k": (1024, 1024, 1040)}.get(case, (1024, 1024, 1024))
    if case.startswith("AH_B"):
        A, opA, op = _rand(rng, K, M), 1, lambda a: a.conj().T
    else:
        hint = {"symmetric": 1, "hermitian": 2, "antihermitian": 3}[case]
        A, opA, op = _structured(rng, 1024, hint), hint + 1, lambda a: a
    B = _rand(rng, K, N)
    C0 = _rand(rng, M, N)
    got = eng16.zgemm(A, B, opA=opA, alpha=0.75, beta=-0.5, C_in=C0)
    ref = 0.75 * (op(A) @ B) - 0.5 * C0
    bound = np.finfo(np.float64).eps * K * np.max(np.abs(A)) * np.max(np.abs(B)) * 8
    assert np.max(np.abs(got - ref)) < bound
    eng16.set_option("zgemm_alg", 1)
    try:
        old = eng16.zgemm(A, B, opA=1 if opA == 1 else 0, alpha=0.75, beta=-0.5, C_in=C0)
    finally:
        eng16.set_option("zgemm_alg", 2)
    if case in ("AH_B", "AH_B_wide", "AH_B_deep_k", "hermitian"):
        assert np.max(np.abs(got - old)) < bound
    assert np.max(np.abs(old - ref)) < bound


@pytest.mark.parametrize("herm", [1, -1])
def test_zgemm_linear_form_hermitian_output(eng16, herm):
    """X^H W X for (anti-)Hermitian W at d = 1024 / 2048: triangle tiles + mirror on the linear-form kernel; 2048 has 136
    triangle tiles (no K-split tail on 256 slots), the engine's d = 4096 chain products exercise the tail + fix-up pass
    (tests/test_gpu_baseline_configs.py)."""
    for d in (1024, 2048):
        rng = np.random.default_rng(d + herm)
        X = _rand(rng, d, d) / np.sqrt(d)
        W = _structured(rng, d, 2 if herm > 0 else 3)
        Z = W @ X
        got = eng16.zgemm(X, Z, opA=1, herm=herm)
        ref = X.conj().T @ Z
        assert np.max(np.abs(got - ref)) < 1e-11 * np.max(np.abs(ref))
        assert np.array_equal(got, herm * got.conj().T)          # exactly (anti-)Hermitian as stored
        if d == 1024:   # accumulate into an (anti-)Hermitian C (the Paterson-Stockmeyer tape's R_i = P_q R_{i+1} + B_i)
            C0 = _structured(rng, d, 2 if herm > 0 else 3)
            got = eng16.zgemm(X, Z, opA=1, herm=herm, alpha=-0.5, beta=2.0, C_in=C0)
            ref2 = -0.5 * ref + 2.0 * C0
            assert np.max(np.abs(got - ref2)) < 1e-11 * np.max(np.abs(ref2))
            assert np.array_equal(got, herm * got.conj().T)


@pytest.mark.parametrize("opA,opB", [(0, 0), (0, 1), (1, 1)])
def test_zgemm_transposing_prepass_for_operands_without_a_k_major_form(eng16, opA, opB):
    """Big single products whose operands have no k-major form (A as stored without a symmetry hint; B^H) reach the linear-form
    kernel through a transposing pre-pass; same product as numpy and as the general kernel (transpose_fallback = 0)."""
    rng = np.random.default_rng(opA * 2 + opB)
    M, N, K = 1024, 1152, 1040
    A = _rand(rng, *((K, M) if opA else (M, K)))
    B = _rand(rng, *((N, K) if opB else (K, N)))
    C0 = _rand(rng, M, N)
    got = eng16.zgemm(A, B, opA=opA, opB=opB, alpha=-0.5, beta=0.25, C_in=C0)
    eng16.set_option("transpose_fallback", 0)
    try:
        old = eng16.zgemm(A, B, opA=opA, opB=opB, alpha=-0.5, beta=0.25, C_in=C0)
    finally:
        eng16.set_option("transpose_fallback", 1)
    ref = -0.5 * ((A.conj().T if opA else A) @ (B.conj().T if opB else B)) + 0.25 * C0
    bound = np.finfo(np.float64).eps * K * np.max(np.abs(A)) * np.max(np.abs(B)) * 8
    assert np.max(np.abs(got - ref)) < bound and np.max(np.abs(old - ref)) < bound
    assert np.array_equal(got, old)          # same 3M arithmetic in the same order on both kernels
