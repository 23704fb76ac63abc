"""GPU parity of the sparse-control fast path of the real-symmetric scheme: when the control Hamiltonians H_k are sparse
(the reference's own problems are: ZZ and X terms, demo_maxcut.py:36-79; BASELINE's X_k controls), the dH assembly
(diffqc.cc:192-197, sim_plain.py:137-143) scatters their nonzeros and the gradient's last reverse-mode level becomes row
dot products over nnz(H_k) (realpath.hip). Checked against the oracle and against the dense passes of the same engine.

Tolerance (BASELINE.json north_star): 1e-10 on amplitudes and gradients in fp64, 1e-4 in fp32.
"""
import numpy as np
import pytest

from diffquantum_amd import problems

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _engine(H0, Hs, M, dtype="c128", **opts):
    from diffquantum_amd.engine import Engine
    e = Engine(len(H0), len(Hs), dtype=dtype)
    e.set_hamiltonians(H0, Hs)
    e.set_observable(M)
    e.set_option("real_min_dim", 1)
    for k, v in opts.items():
        e.set_option(k, v)
    return e


def _sparse_problem(d, K, seed, per_row=2, heavy_row=False):
    """Real symmetric H0 (dense), K sparse real symmetric controls with random patterns: diagonal entries, entries shared
    by two controls, empty rows, and optionally one row/column with many nonzeros (more than one register group)."""
    rng = np.random.default_rng(seed)
    H0 = rng.standard_normal((d, d))
    H0 = (H0 + H0.T) / 2
    Hs = np.zeros((K, d, d))
    for k in range(K):
        for _ in range(per_row * d // 2):
            i, j = rng.integers(0, d, 2)
            v = rng.standard_normal()
            Hs[k, i, j] += v
            if i != j:
                Hs[k, j, i] += v
    if K >= 2:
        Hs[1] += 0.5 * (Hs[0] != 0)          # same pattern in two controls
    if heavy_row:
        v = rng.standard_normal(d)
        Hs[0, 0, :] += v
        Hs[0, :, 0] += v
    Hs[:, d // 2, :] = 0                     # an empty row/column
    Hs[:, :, d // 2] = 0
    M = rng.standard_normal((d, d))
    M = (M + M.T) / 2
    return H0.astype(complex), Hs.astype(complex), M.astype(complex)


def _states(d, B, seed):
    rng = np.random.default_rng(seed)
    psi = rng.standard_normal((d, B)) + 1j * rng.standard_normal((d, B))
    return psi / np.linalg.norm(psi, axis=0)


@pytest.mark.parametrize("n,B,S", [(3, 4, 6), (4, 20, 5), (5, 3, 4), (6, 70, 3)])
def test_ising_x_controls_match_oracle_and_dense(oracle, n, B, S):
    """BASELINE's problem family (random Ising drift, X_k controls): state form (B < d) and operator form (B >= d)."""
    dt = 0.03
    H0, Hs, M = problems.random_ising(n, seed=n)
    u = np.random.default_rng(S).uniform(-1, 1, (S, n))
    psi0 = problems.random_states(1 << n, B, seed=2)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, sparse_generator=0) as e:   # (the drift is sparse too: see the sparse-generator tests below)
        assert e.info("control_nonzeros") == n * (1 << n)
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("real_path") >= 1 and e.info("sparse_controls") == 1
        p1 = e.propagate(u, dt, psi0)
        e.set_option("sparse_controls", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_controls") == 0
    assert abs(l1 - rl) < TOL and np.max(np.abs(g1 - rg)) < TOL and np.max(np.abs(f1 - rf)) < TOL
    assert np.max(np.abs(p1 - rf)) < TOL
    assert abs(l1 - l0) < 1e-12 * max(1, B) and np.max(np.abs(g1 - g0)) < 1e-12 and np.max(np.abs(f1 - f0)) < 1e-13


@pytest.mark.parametrize("d,K,B,heavy", [(8, 1, 3, False), (24, 3, 5, True), (64, 2, 64, True), (100, 4, 7, False),
                                         (130, 2, 131, True)])
def test_random_sparse_controls(oracle, d, K, B, heavy):
    """Arbitrary sparse symmetric controls and dimensions that are not powers of two (bounds of the row strips)."""
    S, dt = 4, 0.02
    H0, Hs, M = _sparse_problem(d, K, seed=d + K, heavy_row=heavy)
    u = np.random.default_rng(d).uniform(-1, 1, (S, K))
    psi0 = _states(d, B, seed=1)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M) as e:
        nnz = int(np.count_nonzero(Hs))
        assert e.info("control_nonzeros") == nnz
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_controls") == 1 and e.info("real_path") >= 1
    assert abs(l1 - rl) < TOL * max(1, B) and np.max(np.abs(g1 - rg)) < TOL and np.max(np.abs(f1 - rf)) < TOL


def test_dense_controls_keep_the_dense_passes():
    H0, Hs, M = problems.random_real_symmetric_problem(4, 3, seed=1)
    with _engine(H0, Hs, M) as e:
        assert e.info("control_nonzeros") == -1 and e.info("sparse_controls") == 0
    H0, Hs, M = problems.random_hermitian_problem(3, 2, seed=1)   # complex: no real path at all
    with _engine(H0, Hs, M) as e:
        assert e.info("control_nonzeros") == -1


def test_sparse_controls_with_squarings(oracle):
    """A step too large for the polynomial alone: scaled by 2^-s and squared in the real algebra; the sparse contraction
    then pairs the cotangents with X / 2^s."""
    n, B, S, dt = 4, 6, 3, 0.9
    H0, Hs, M = problems.random_ising(n, seed=3)
    u = np.random.default_rng(1).uniform(-1, 1, (S, n))
    psi0 = problems.random_states(1 << n, B, seed=2)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M) as e:
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("squarings") >= 1 and e.info("sparse_controls") == 1
    assert abs(l1 - rl) < 1e-9 and np.max(np.abs(g1 - rg)) < 1e-9 and np.max(np.abs(f1 - rf)) < 1e-10


def test_sparse_controls_fp32(oracle):
    n, B, S, dt = 5, 40, 5, 0.03
    H0, Hs, M = problems.random_ising(n, seed=5)
    u = np.random.default_rng(2).uniform(-1, 1, (S, n))
    psi0 = problems.random_states(1 << n, B, seed=2)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, dtype="c64") as e:
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0.astype(np.complex64))
        assert e.info("sparse_controls") == 1 and e.info("real_path") >= 1
    assert abs(l1 - rl) < 1e-4 * B and np.max(np.abs(g1 - rg)) < 1e-4 and np.max(np.abs(f1 - rf)) < 1e-4


def test_sparse_controls_mid_dimension_cached_and_sliced():
    """d = 256 with the propagator cache live and the chain split over three virtual ranks: sparse and dense passes agree."""
    from diffquantum_amd.distributed import shard_range, virtual_time_sliced_energy_grad
    n, B, S, dt = 8, 256, 12, 2e-3
    p = problems.qaoa_ring(n)
    H0, Hs = p["H0"].astype(complex), p["Hs"].astype(complex)
    M = H0
    u = np.random.default_rng(0).uniform(-1, 1, (S, len(Hs)))
    psi0 = problems.random_states(1 << n, B, seed=3)
    with _engine(H0, Hs, M, real_min_dim=128, sparse_generator=0) as e:   # controls-only scheme (it is the one that caches)
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_controls") == 1 and e.info("cached_steps") > 0
        e.set_option("sparse_controls", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
    assert abs(l1 - l0) < 1e-11 * B and np.max(np.abs(g1 - g0)) < 1e-12 and np.max(np.abs(f1 - f0)) < 1e-13
    engines = [_engine(H0, Hs, M, real_min_dim=128, sparse_generator=0) for _ in range(3)]
    try:
        shards = [np.ascontiguousarray(psi0[:, f:f + c]) for f, c in (shard_range(B, r, 3) for r in range(3))]
        res = virtual_time_sliced_energy_grad(engines, u, dt, shards)
    finally:
        for e in engines:
            e.close()
    for l, g, f, en in res:
        assert abs(l - l0) < 1e-10 * B and np.max(np.abs(g - g0)) < TOL


def test_zero_and_diagonal_controls(oracle):
    """A control that is identically zero (no nonzeros at all), a purely diagonal one, and all controls zero."""
    d, B, S, dt = 16, 4, 5, 0.05
    rng = np.random.default_rng(7)
    H0 = rng.standard_normal((d, d))
    H0 = ((H0 + H0.T) / 2).astype(complex)
    M = np.diag(rng.standard_normal(d)).astype(complex)
    Hs = np.zeros((3, d, d), dtype=complex)
    Hs[1] = np.diag(rng.standard_normal(d))
    Hs[2, 0, 5] = Hs[2, 5, 0] = 0.7
    u = rng.uniform(-1, 1, (S, 3))
    psi0 = _states(d, B, seed=3)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M) as e:
        assert e.info("control_nonzeros") == d + 2
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_controls") == 1
    assert np.all(g1[:, 0] == 0.0)
    assert abs(l1 - rl) < TOL and np.max(np.abs(g1 - rg)) < TOL and np.max(np.abs(f1 - rf)) < TOL
    with _engine(H0, np.zeros((2, d, d), dtype=complex), M) as e:
        assert e.info("control_nonzeros") == 0
        l2, g2, f2, _ = e.energy_grad(u[:, :2], dt, psi0)
        assert e.info("sparse_controls") == 1 and np.all(g2 == 0.0)
    rl2, rg2, rf2 = oracle.energy_grad(H0, [np.zeros((d, d))] * 2, M, u[:, :2], dt, psi0)
    assert abs(l2 - rl2) < TOL and np.max(np.abs(f2 - rf2)) < TOL


# ---- sparse generator: drift AND controls sparse -> the polynomial runs on sparse x dense passes ---------------------

def _sparse_drift_problem(d, K, seed, per_row=2):
    """Sparse real symmetric drift (diagonal + a few off-diagonal pairs) and sparse controls, random patterns."""
    H0, Hs, M = _sparse_problem(d, K, seed, per_row=per_row)
    rng = np.random.default_rng(seed + 1000)
    D = np.diag(rng.standard_normal(d))
    for _ in range(d):
        i, j = rng.integers(0, d, 2)
        v = rng.standard_normal()
        D[i, j] += v
        if i != j:
            D[j, i] += v
    D[d // 2, :] = 0
    D[:, d // 2] = 0
    return D.astype(complex), Hs, M


@pytest.mark.parametrize("n,B,S", [(3, 4, 6), (4, 20, 5), (5, 3, 4), (6, 70, 3)])
def test_sparse_generator_ising_matches_oracle_and_other_schemes(oracle, n, B, S):
    """BASELINE's problem family: diagonal Ising drift + X_k controls -> X_t has n + 1 nonzeros per row."""
    dt = 0.03
    H0, Hs, M = problems.random_ising(n, seed=n)
    u = np.random.default_rng(S).uniform(-1, 1, (S, n))
    psi0 = problems.random_states(1 << n, B, seed=2)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, sparse_generator=1) as e:   # (auto takes the scheme from dim >= 64 nonzeros-per-row)
        assert e.info("sparse_drift") == 1
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("real_path") >= 1 and e.info("sparse_generator") == 1 and e.info("squarings") == 0
        p1 = e.propagate(u, dt, psi0)
        e.set_option("sparse_generator", 0)      # controls-only sparse scheme (dense symmetric products)
        l2, g2, f2, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_generator") == 0 and e.info("sparse_controls") == 1
        e.set_option("sparse_controls", 0)       # dense passes throughout
        l3, g3, f3, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_controls") == 0
    assert abs(l1 - rl) < TOL and np.max(np.abs(g1 - rg)) < TOL and np.max(np.abs(f1 - rf)) < TOL
    assert np.max(np.abs(p1 - rf)) < TOL
    for l, g, f in ((l2, g2, f2), (l3, g3, f3)):
        assert abs(l1 - l) < 1e-12 * max(1, B) and np.max(np.abs(g1 - g)) < 1e-12 and np.max(np.abs(f1 - f)) < 1e-13


@pytest.mark.parametrize("q", [1, 2, 3, 4, 5, 6])
def test_sparse_generator_every_polynomial_size(oracle, q):
    """Forced q = 1..6 (powers up to X^13: two level groups in the gradient contraction from q = 4); the step is chosen so
    that each q resolves it, and the result is compared with the dense symmetric scheme at the SAME q (same polynomial)."""
    n, B, S = 4, 9, 4
    dt = {1: 1e-4, 2: 2e-3, 3: 0.01, 4: 0.03, 5: 0.06, 6: 0.1}[q]
    H0, Hs, M = problems.random_ising(n, seed=q)
    u = np.random.default_rng(q).uniform(-1, 1, (S, n))
    psi0 = problems.random_states(1 << n, B, seed=q)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, real_path=q, sparse_generator=1) as e:
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_generator") == 1 and e.info("real_path") == q
        e.set_option("sparse_generator", 0)
        l2, g2, f2, _ = e.energy_grad(u, dt, psi0)
    assert abs(l1 - l2) < 1e-12 and np.max(np.abs(g1 - g2)) < 1e-12 and np.max(np.abs(f1 - f2)) < 1e-13
    assert abs(l1 - rl) < 1e-9 and np.max(np.abs(g1 - rg)) < 1e-9 and np.max(np.abs(f1 - rf)) < 1e-9


@pytest.mark.parametrize("d,K,B", [(16, 1, 3), (24, 3, 5), (64, 2, 64), (100, 4, 7), (130, 2, 131)])
def test_sparse_generator_random_patterns(oracle, d, K, B):
    """Random sparse drift and controls: shared entries, empty rows, off-diagonal drift, dimensions that are not powers of two."""
    S, dt = 4, (0.02 if d < 100 else 0.004)   # (small enough for a polynomial without squarings: the scheme's regime)
    H0, Hs, M = _sparse_drift_problem(d, K, seed=d + K)
    u = np.random.default_rng(d).uniform(-1, 1, (S, K))
    psi0 = _states(d, B, seed=1)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, sparse_generator=1) as e:
        assert e.info("sparse_drift") == 1 and e.info("control_nonzeros") == int(np.count_nonzero(Hs))
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_generator") == 1
    assert abs(l1 - rl) < TOL * max(1, B) and np.max(np.abs(g1 - rg)) < TOL and np.max(np.abs(f1 - rf)) < TOL


def test_sparse_generator_large_step_falls_back_to_squarings(oracle):
    n, B, S, dt = 4, 6, 3, 0.9
    H0, Hs, M = problems.random_ising(n, seed=3)
    u = np.random.default_rng(1).uniform(-1, 1, (S, n))
    psi0 = problems.random_states(1 << n, B, seed=2)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, sparse_generator=1) as e:
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("squarings") >= 1 and e.info("sparse_generator") == 0 and e.info("sparse_controls") == 1
    assert abs(l1 - rl) < 1e-9 and np.max(np.abs(g1 - rg)) < 1e-9 and np.max(np.abs(f1 - rf)) < 1e-10


def test_sparse_generator_fp32_and_sliced(oracle):
    from diffquantum_amd.distributed import shard_range, virtual_time_sliced_energy_grad
    n, B, S, dt = 5, 40, 6, 0.03
    H0, Hs, M = problems.random_ising(n, seed=5)
    u = np.random.default_rng(2).uniform(-1, 1, (S, n))
    psi0 = problems.random_states(1 << n, B, seed=2)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, dtype="c64", sparse_generator=1) as e:
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0.astype(np.complex64))
        assert e.info("sparse_generator") == 1
    assert abs(l1 - rl) < 1e-4 * B and np.max(np.abs(g1 - rg)) < 1e-4 and np.max(np.abs(f1 - rf)) < 1e-4
    engines = [_engine(H0, Hs, M, sparse_generator=1) for _ in range(3)]
    try:
        shards = [np.ascontiguousarray(psi0[:, f:f + c]) for f, c in (shard_range(B, r, 3) for r in range(3))]
        res = virtual_time_sliced_energy_grad(engines, u, dt, shards)
        assert all(e.info("sparse_generator") == 1 for e in engines)
    finally:
        for e in engines:
            e.close()
    for l, g, f, en in res:
        assert abs(l - rl) < TOL * B and np.max(np.abs(g - rg)) < TOL


def test_sparse_generator_with_the_propagator_cache():
    """A chain that is mostly cacheable (explicit budget, as a rank's slice of a time-sliced chain has): the forward pass
    leaves X_1 .. X_2q and U in the cache, the backward pass skips its 2q rebuild passes; same results as without."""
    n, B, S, dt = 8, 256, 12, 2e-3
    p = problems.qaoa_ring(n)
    H0, Hs = p["H0"].astype(complex), p["Hs"].astype(complex)
    u = np.random.default_rng(0).uniform(-1, 1, (S, len(Hs)))
    psi0 = problems.random_states(1 << n, B, seed=3)
    with _engine(H0, Hs, H0, real_min_dim=128, cache_bytes=2e9, sparse_generator=1, state_columns=0) as e:
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_generator") == 1 and e.info("cached_steps") == S
        e.set_option("cache_bytes", -1)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_generator") == 1 and e.info("cached_steps") == 0
        e.set_option("cache_bytes", 0)      # auto: a 12-step chain fits whole
        e.energy_grad(u, dt, psi0)
        assert e.info("cached_steps") == S
        e.set_option("plan_steps", 10 ** 7)  # ... a chain of which only a sliver would fit does not take the cache
        e.energy_grad(u, dt, psi0)
        assert e.info("cached_steps") == 0
    assert l1 == l0 and np.array_equal(g1, g0) and np.array_equal(f1, f0)   # same kernels on the same data


def test_sparse_generator_auto_rule():
    """auto: the scheme is taken where a sparse x dense pass beats an MFMA product (dim >= 64 nonzeros-per-row)."""
    n = 8
    p = problems.qaoa_ring(n)                 # d = 256, 17 nonzeros per row: the MFMA products win
    H0, Hs = p["H0"].astype(complex), p["Hs"].astype(complex)
    u = np.random.default_rng(0).uniform(-1, 1, (3, len(Hs)))
    psi0 = problems.random_states(1 << n, 4, seed=3)
    with _engine(H0, Hs, H0, real_min_dim=128, xor_regular=0) as e:
        assert e.info("sparse_drift") == 1
        e.energy_grad(u, 2e-3, psi0)
        assert e.info("sparse_generator") == 0 and e.info("sparse_controls") == 1 and e.info("real_path") >= 1
        e.set_option("xor_regular", 1)        # ... unless the generator is a sum of Pauli strings: the column form needs no pass at all
        e.energy_grad(u, 2e-3, psi0)
        assert e.info("sparse_generator") == 1 and e.info("xor_regular") == 1
    H0, Hs, M = problems.random_ising(10, seed=1)   # d = 1024, 11 nonzeros per row: the sparse passes win
    u = np.random.default_rng(0).uniform(-1, 1, (2, 10))
    psi0 = problems.random_states(1024, 4, seed=3)
    with _engine(H0, Hs, M, real_min_dim=128) as e:
        e.energy_grad(u, 1e-3, psi0)
        assert e.info("sparse_generator") == 1


def test_sparse_generator_row_with_more_than_64_entries(oracle):
    """One row/column of a control with ~100 nonzeros (d = 128, still within the sparsity bound overall): the kernels'
    64-entry staging rounds (sparse x dense passes and the level contraction) are taken more than once for that row."""
    d, K, B, S, dt = 128, 2, 5, 3, 0.002
    rng = np.random.default_rng(11)
    H0 = np.diag(rng.standard_normal(d)).astype(complex)
    Hs = np.zeros((K, d, d), dtype=complex)
    v = rng.standard_normal(d) * (rng.random(d) < 0.8)
    Hs[0, 3, :] += v
    Hs[0, :, 3] += v
    for i in range(0, d - 1, 2):
        Hs[1, i, i + 1] = Hs[1, i + 1, i] = rng.standard_normal()
    M = np.diag(rng.standard_normal(d)).astype(complex)
    u = rng.uniform(-1, 1, (S, K))
    psi0 = _states(d, B, seed=2)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, sparse_generator=1) as e:
        assert e.info("sparse_drift") == 1 and np.count_nonzero(Hs[0, 3]) + 1 > 64
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_generator") == 1
    assert abs(l1 - rl) < TOL and np.max(np.abs(g1 - rg)) < TOL and np.max(np.abs(f1 - rf)) < TOL


@pytest.mark.parametrize("q", [2, 3, 5])
def test_sparse_generator_fused_levels_equal_separate_contraction(q):
    """The gradient's level dot products inside the reverse passes (default) against the separate level-contraction kernel."""
    d, K, B, S = 64, 3, 16, 4
    dt = {2: 2e-4, 3: 1e-3, 5: 0.01}[q]
    H0, Hs, M = _sparse_drift_problem(d, K, seed=q)
    u = np.random.default_rng(q).uniform(-1, 1, (S, K))
    psi0 = _states(d, B, seed=4)
    with _engine(H0, Hs, M, sparse_generator=1, real_path=q) as e:
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_generator") == 1
        e.set_option("fuse_levels", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
    assert l1 == l0 and np.array_equal(f1, f0) and np.max(np.abs(g1 - g0)) < 1e-13 * max(1.0, np.abs(g0).max())


# ---- XCD-slab form of the sparse x dense passes (dimension a multiple of 1024 in fp64, 2048 in fp32) ----------------

def _slab_problem(d, seed):
    """Sparse drift (diagonal + off-diagonal pairs) and two sparse controls at a slab-eligible dimension: one row/column with
    ~100 nonzeros (more than the 64-entry lane staging of the product pass and than the 16-entry rounds of the fused dot
    pass), an empty row, rows with 1-3 entries."""
    rng = np.random.default_rng(seed)
    H0 = np.diag(rng.standard_normal(d))
    Hs = np.zeros((2, d, d))
    for _ in range(d):
        i, j = rng.integers(0, d, 2)
        v = rng.standard_normal()
        H0[i, j] += v
        if i != j:
            H0[j, i] += v
    v = rng.standard_normal(d) * (rng.random(d) < 0.1)
    Hs[0, 5, :] += v
    Hs[0, :, 5] += v
    for i in range(0, d - 1, 2):
        Hs[1, i, i + 1] = Hs[1, i + 1, i] = rng.standard_normal()
    for X in (H0, Hs[0], Hs[1]):
        X[d // 2, :] = 0
        X[:, d // 2] = 0
    M = np.diag(rng.standard_normal(d))
    return H0.astype(complex), Hs.astype(complex), M.astype(complex)


def test_sparse_generator_slab_form_matches_row_form_and_oracle(oracle):
    d, B, S, dt = 1024, 6, 2, 0.002
    H0, Hs, M = _slab_problem(d, 3)
    assert np.count_nonzero(Hs[0, 5]) + np.count_nonzero(H0[5]) > 64
    u = np.random.default_rng(1).uniform(-1, 1, (S, 2))
    psi0 = _states(d, B, seed=4)
    rl, rg, rf = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, dt, psi0)
    res = {}
    for compose in (1, 0):
        for slab in (1, 0):
            with _engine(H0, Hs, M, sparse_generator=1, slab_spmm=slab, compose=compose) as e:
                l, g, f, _ = e.energy_grad(u, dt, psi0)
                assert e.info("sparse_generator") == 1 and e.info("operator_backward") == compose
            assert abs(l - rl) < TOL * B and np.max(np.abs(g - rg)) < TOL and np.max(np.abs(f - rf)) < TOL, (compose, slab)
            res[(compose, slab)] = (l, g, f)
        (l1, g1, f1), (l0, g0, f0) = res[(compose, 1)], res[(compose, 0)]
        assert abs(l1 - l0) < 1e-12 * B and np.max(np.abs(g1 - g0)) < 1e-13 and np.max(np.abs(f1 - f0)) < 1e-14


def test_sparse_generator_slab_form_fp32(oracle):
    d, B, S, dt = 2048, 4, 2, 0.002
    H0, Hs, M = _slab_problem(d, 9)
    u = np.random.default_rng(2).uniform(-1, 1, (S, 2))
    psi0 = _states(d, B, seed=5)
    rl, rg, rf = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, dtype="c64", sparse_generator=1) as e:
        l, g, f, _ = e.energy_grad(u, dt, psi0.astype(np.complex64))
        assert e.info("sparse_generator") == 1
    assert abs(l - rl) < 1e-4 * B and np.max(np.abs(g - rg)) < 1e-4 and np.max(np.abs(f - rf)) < 1e-4
