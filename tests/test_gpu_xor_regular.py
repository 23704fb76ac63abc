"""GPU parity of the column-form kernels for XOR-regular sparse generators (diffquantum_amd/csrc/xorexp.hip): the same step
U = expm(-1j dt H(u)) (sim_plain.py:145-146, diffqc.cc:192-198) and its adjoint gradient as every other scheme of the engine,
evaluated column by column in LDS when H is a sum of Pauli strings.  Checked against the oracle (d = 512) and against the
row-wise sparse kernels of the same engine (option xor_regular = 0) up to the headline dimension.

Tolerance (BASELINE.json north_star): 1e-10 on amplitudes and gradients in fp64.
"""
import numpy as np
import pytest

from diffquantum_amd import problems

pytestmark = pytest.mark.gpu
TOL = 1e-10

PX = np.array([[0, 1], [1, 0]], dtype=float)
PZ = np.array([[1, 0], [0, -1]], dtype=float)
PIY = np.array([[0, 1], [-1, 0]], dtype=float)   # i Y (real): an even number of them makes a real symmetric string


def _string(n, ops):
    """Kronecker product of one-qubit operators; ops = {qubit: 'X' | 'Z' | 'Y'}; pairs of Y become -(iY)(iY)."""
    m = np.ones((1, 1))
    ny = sum(1 for v in ops.values() if v == "Y")
    assert ny % 2 == 0
    for q in range(n):
        o = ops.get(q)
        m = np.kron(m, np.eye(2) if o is None else PX if o == "X" else PZ if o == "Z" else PIY)
    return m * ((-1.0) ** (ny // 2))


def _pauli_problem(n, seed):
    """Drift: ZZ couplings + Z fields (diagonal) + a transverse term h X_0 (drift entries off the diagonal).  Controls: X_k;
    X_a X_b (two-bit offset); Y_a Y_b (negative entries); X_a Z_b (sign pattern); Z_a (a control that is diagonal);
    X_a + 0.5 Z_b (one control with both an XOR-diagonal and diagonal entries)."""
    rng = np.random.default_rng(seed)
    d = 1 << n
    H0 = np.zeros((d, d))
    for a in range(n):
        H0 += rng.standard_normal() * _string(n, {a: "Z"})
        H0 += rng.standard_normal() * _string(n, {a: "Z", (a + 1) % n: "Z"})
    H0 += 0.7 * _string(n, {0: "X"})
    Hs = [_string(n, {k: "X"}) for k in range(1, 4)]
    Hs.append(_string(n, {1: "X", 4: "X"}))
    Hs.append(_string(n, {2: "Y", 5: "Y"}))
    Hs.append(1.5 * _string(n, {3: "X", 6: "Z"}))
    Hs.append(_string(n, {4: "Z"}))
    Hs.append(_string(n, {7: "X"}) + 0.5 * _string(n, {2: "Z"}))
    M = rng.standard_normal((d, d))
    M = (M + M.T) / 2
    return H0.astype(complex), np.array(Hs).astype(complex), M.astype(complex)


def _engine(H0, Hs, M, dtype="c128", **opts):
    """(state_columns = 0: this file is about the column form of the PROPAGATORS, xorexp.hip; the state-column form, which the
    engine would otherwise take for these generators at narrow batches, has tests/test_gpu_state_columns.py)"""
    from diffquantum_amd.engine import Engine
    e = Engine(len(H0), len(Hs), dtype=dtype)
    e.set_hamiltonians(H0, Hs)
    e.set_observable(M)
    opts.setdefault("state_columns", 0)
    for k, v in opts.items():
        e.set_option(k, v)
    return e


@pytest.mark.parametrize("B", [6, 512])
def test_ising_column_form_matches_oracle(oracle, B):
    """d = 512 (the smallest dimension the kernels take), BASELINE's family; state form (B < d) and operator form (B = d)."""
    n, S, dt = 9, 3, 1e-3
    H0, Hs, M = problems.random_ising(n, seed=4)
    u = np.random.default_rng(1).uniform(-1, 1, (S, n))
    psi0 = problems.random_states(1 << n, B, seed=2)
    rl, rg, rf = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, sparse_generator=1) as e:
        assert e.info("xor_regular_found") == n
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_generator") == 1 and e.info("xor_regular") == 1
        p1 = e.propagate(u, dt, psi0)
        e.set_option("xor_regular", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_generator") == 1 and e.info("xor_regular") == 0
    assert abs(l1 - rl) < TOL * max(1, B / 8) and np.max(np.abs(g1 - rg)) < TOL and np.max(np.abs(f1 - rf)) < TOL
    assert np.max(np.abs(p1 - rf)) < TOL
    assert abs(l1 - l0) < 1e-12 * B and np.max(np.abs(g1 - g0)) < 1e-12 and np.max(np.abs(f1 - f0)) < 1e-13


@pytest.mark.parametrize("q", [1, 2, 3, 4, 6])
def test_pauli_strings_with_signs_and_diagonal_controls(oracle, q):
    """Every entry class of the table: drift entries off the diagonal, two-bit offsets, negative entries, a diagonal control,
    a control with both kinds of entries -- at several polynomial sizes (q is a runtime parameter of the kernels)."""
    n, S, B = 9, 3, 5
    dt = {1: 2e-5, 2: 4e-4, 3: 2e-3, 4: 5e-3, 6: 1.5e-2}[q]
    H0, Hs, M = _pauli_problem(n, seed=q)
    u = np.random.default_rng(q).uniform(-1, 1, (S, len(Hs)))
    psi0 = problems.random_states(1 << n, B, seed=3)
    rl, rg, rf = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, sparse_generator=1, real_path=q) as e:
        assert e.info("xor_regular_found") == 8          # X_0 (drift), X_1..X_3, XX, YY, XZ, X_7
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("xor_regular") == 1 and e.info("real_path") == q
        e.set_option("xor_regular", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("xor_regular") == 0
    assert abs(l1 - l0) < 1e-12 and np.max(np.abs(g1 - g0)) < 1e-12 and np.max(np.abs(f1 - f0)) < 1e-13
    assert abs(l1 - rl) < 1e-9 and np.max(np.abs(g1 - rg)) < 1e-9 and np.max(np.abs(f1 - rf)) < 1e-9


def test_irregular_patterns_keep_the_row_kernels():
    """A diagonal of varying magnitude, or a missing entry, is not XOR-regular: the row-wise kernels stay."""
    n = 9
    d = 1 << n
    H0, Hs, M = problems.random_ising(n, seed=1)
    Hs = Hs.copy()
    Hs[0, 0, 1 << (n - 1)] = Hs[0, 1 << (n - 1), 0] = 2.0        # one pair of another magnitude
    u = np.random.default_rng(0).uniform(-1, 1, (2, n))
    psi0 = problems.random_states(d, 4, seed=1)
    with _engine(H0, Hs, M, sparse_generator=1) as e:
        assert e.info("xor_regular_found") == -1
        e.energy_grad(u, 0.01, psi0)
        assert e.info("sparse_generator") == 1 and e.info("xor_regular") == 0
    Hs[0, 0, 1 << (n - 1)] = Hs[0, 1 << (n - 1), 0] = 0.0        # ... and a hole
    with _engine(H0, Hs, M, sparse_generator=1) as e:
        assert e.info("xor_regular_found") == -1


@pytest.mark.parametrize("n,B,S", [(10, 1024, 4), (11, 64, 3), (12, 4096, 2), (12, 32, 3)])
def test_column_form_against_row_kernels_up_to_the_headline_dimension(n, B, S):
    """d = 1024 .. 4096 (all workgroup shapes), automatic scheme choice; B >= d is the operator form of the headline."""
    from diffquantum_amd.engine import Engine
    dt = 5e-4
    H0, Hs, M = problems.random_ising(n, seed=n)
    u = problems.pulse_table(n, S, S * dt, seed=2)
    psi0 = problems.random_states(1 << n, B, seed=3)
    with Engine(1 << n, n) as e:
        e.set_hamiltonians(H0, Hs)
        e.set_observable(M)
        e.set_option("state_columns", 0)
        assert e.info("xor_regular_found") == n
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_generator") == 1 and e.info("xor_regular") == 1
        e.set_option("xor_regular", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("xor_regular") == 0
    assert abs(l1 - l0) < 1e-11 * B and np.max(np.abs(g1 - g0)) < 1e-12 and np.max(np.abs(f1 - f0)) < 1e-13


def test_explicit_propagators_and_their_vjp_in_column_form():
    """dq_build_propagators / dq_propagator_vjp (the propagator-sharded multi-GPU form's two calls): U_t and the pull-back of
    arbitrary cotangents G_t, column form against the row-wise kernels."""
    n, S, dt = 9, 3, 1e-3
    d = 1 << n
    H0, Hs, M = problems.random_ising(n, seed=7)
    u = np.random.default_rng(3).uniform(-1, 1, (S, n))
    rng = np.random.default_rng(4)
    G = rng.standard_normal((S, d, d)) + 1j * rng.standard_normal((S, d, d))
    with _engine(H0, Hs, M, sparse_generator=1) as e:
        U1 = e.build_propagators(u, dt)
        assert e.info("xor_regular") == 1
        g1 = e.propagator_vjp(u, dt, G)
        e.set_option("xor_regular", 0)
        U0 = e.build_propagators(u, dt)
        assert e.info("xor_regular") == 0
        g0 = e.propagator_vjp(u, dt, G)
    assert np.max(np.abs(U1 - U0)) < 1e-14 and np.max(np.abs(g1 - g0)) < 1e-11 * np.max(np.abs(g0))
    assert np.max(np.abs(U1 - np.transpose(U1, (0, 2, 1)))) < 1e-15       # complex symmetric (rows written as columns)


def test_column_form_with_the_propagator_cache_and_time_slices():
    """A cached column-form chunk keeps U alone (no powers): the backward sweep reuses it and the reverse kernel needs nothing else.
    Same numbers with and without the cache, and through the time-sliced form on three virtual ranks."""
    from diffquantum_amd.distributed import shard_range, virtual_time_sliced_energy_grad
    n, B, S, dt = 9, 512, 12, 1e-3
    H0, Hs, M = problems.random_ising(n, seed=2)
    u = np.random.default_rng(0).uniform(-1, 1, (S, n))
    psi0 = problems.random_states(1 << n, B, seed=3)
    with _engine(H0, Hs, M, sparse_generator=1, cache_bytes=2e9, state_columns=0) as e:
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("xor_regular") == 1 and e.info("cached_steps") == S
        e.set_option("cache_bytes", -1)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("xor_regular") == 1 and e.info("cached_steps") == 0
        e.set_option("xor_regular", 0)
        l2, g2, f2, _ = e.energy_grad(u, dt, psi0)
    assert l1 == l0 and np.array_equal(g1, g0) and np.array_equal(f1, f0)      # same kernels on the same data
    assert abs(l1 - l2) < 1e-12 * B and np.max(np.abs(g1 - g2)) < 1e-12
    engines = [_engine(H0, Hs, M, sparse_generator=1) for _ in range(3)]
    try:
        shards = [np.ascontiguousarray(psi0[:, f:f + c]) for f, c in (shard_range(B, r, 3) for r in range(3))]
        res = virtual_time_sliced_energy_grad(engines, u, dt, shards)
        assert all(e.info("xor_regular") == 1 for e in engines)
    finally:
        for e in engines:
            e.close()
    for l, g, f, en in res:
        assert abs(l - l2) < 1e-10 * B and np.max(np.abs(g - g2)) < TOL


@pytest.mark.parametrize("B,S", [(7, 5), (256, 20)])
def test_qaoa_ring_d256_column_form_matches_oracle(oracle, B, S):
    """BASELINE configs[1]'s problem (8-qubit ring QAOA: X_k and Z_k Z_k+1 controls, d = 256 -- the smallest dimension of the
    kernels, one wave per workgroup, many steps stacked per launch): taken automatically, against the oracle and the MFMA scheme."""
    n, dt = 8, 2e-3
    p = problems.qaoa_ring(n)
    H0, Hs, M = p["H0"].astype(complex), p["Hs"].astype(complex), p["H_cost"].astype(complex)
    u = np.random.default_rng(S).uniform(-1, 1, (S, len(Hs)))
    psi0 = problems.random_states(1 << n, B, seed=5)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M) as e:
        assert e.info("xor_regular_found") >= n
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("sparse_generator") == 1 and e.info("xor_regular") == 1
        p1 = e.propagate(u, dt, psi0)
        e.set_option("xor_regular", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("xor_regular") == 0 and e.info("sparse_generator") == 0
    assert abs(l1 - rl) < TOL * max(1, B / 8) and np.max(np.abs(g1 - rg)) < TOL and np.max(np.abs(f1 - rf)) < TOL
    assert np.max(np.abs(p1 - rf)) < TOL
    assert abs(l1 - l0) < 1e-12 * max(1, B) and np.max(np.abs(g1 - g0)) < 1e-12 and np.max(np.abs(f1 - f0)) < 1e-13


@pytest.mark.parametrize("signed", [False, True])
def test_column_form_fp32(oracle, signed):
    """complex64 engines take the same kernels in float (1e-4, BASELINE's fp32 tolerance); gradients accumulate in double."""
    n, S, B, dt = 9, 4, 6, 1e-3
    if signed:
        H0, Hs, M = _pauli_problem(n, seed=11)
    else:
        H0, Hs, M = problems.random_ising(n, seed=6)
    u = np.random.default_rng(8).uniform(-1, 1, (S, len(Hs)))
    psi0 = problems.random_states(1 << n, B, seed=9)
    rl, rg, rf = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, dtype="c64", sparse_generator=1) as e:
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0.astype(np.complex64))
        assert e.info("xor_regular") == 1
        e.set_option("xor_regular", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0.astype(np.complex64))
        assert e.info("xor_regular") == 0
    assert abs(l1 - rl) < 1e-4 * B and np.max(np.abs(g1 - rg)) < 1e-4 and np.max(np.abs(f1 - rf)) < 1e-4
    assert abs(l1 - l0) < 1e-4 * B and np.max(np.abs(g1 - g0)) < 1e-4 and np.max(np.abs(f1 - f0)) < 1e-5


# ---- complex Hermitian generators: X and Y drives (zxr_* kernels) ---------------------------------------------------------

PY = np.array([[0, -1j], [1j, 0]])


def _cstring(n, ops):
    m = np.ones((1, 1), dtype=complex)
    for q in range(n):
        o = ops.get(q)
        m = np.kron(m, np.eye(2) if o is None else PX if o == "X" else PZ if o == "Z" else PY)
    return m


def _xy_problem(n, seed, extra=True):
    """Ising drift (ZZ + Z, diagonal) with a transverse X_0 and a Y_1 term in the drift; controls X_k and Y_k on several qubits
    (X_a and Y_a share an XOR offset), and with `extra` a X_a Y_b string (imaginary entries, two-bit offset) and a Y_a Z_b string."""
    rng = np.random.default_rng(seed)
    d = 1 << n
    H0 = np.zeros((d, d), dtype=complex)
    for a in range(n):
        H0 += rng.standard_normal() * _cstring(n, {a: "Z"}) + rng.standard_normal() * _cstring(n, {a: "Z", (a + 1) % n: "Z"})
    H0 += 0.6 * _cstring(n, {0: "X"}) - 0.4 * _cstring(n, {1: "Y"})
    Hs = [_cstring(n, {k: "X"}) for k in range(2, 5)] + [0.5 * _cstring(n, {k: "Y"}) for k in range(3, 7)]
    if extra:
        Hs.append(_cstring(n, {2: "X", 7: "Y"}))
        Hs.append(1.25 * _cstring(n, {5: "Y", 8 % n: "Z"}))
    M = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
    M = (M + M.conj().T) / 2
    return H0, np.array(Hs), M


@pytest.mark.parametrize("B", [5, 512])
def test_complex_pauli_generator_matches_oracle_and_the_tape(oracle, B):
    """X and Y drives: H complex Hermitian, sparse, XOR-regular with phases +-1 / +-i.  The complex column form (rows of
    U = exp(-i dt H) built in LDS, gradients from the rows of Ubar) against the oracle and against the complex tape."""
    n, S, dt = 9, 3, 2e-3
    H0, Hs, M = _xy_problem(n, seed=3)
    u = np.random.default_rng(2).uniform(-1, 1, (S, len(Hs)))
    psi0 = problems.random_states(1 << n, B, seed=4)
    rl, rg, rf = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M) as e:
        assert e.info("real_hamiltonians") == 0 and e.info("xor_regular_found") == 11   # X_0, Y_1 (drift), 3 X, 4 Y, XY, YZ
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("xor_regular") == 2 and e.info("real_path") == 0
        p1 = e.propagate(u, dt, psi0)
        e.set_option("xor_regular", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("xor_regular") == 0
    assert abs(l1 - rl) < TOL * max(1, B / 8) and np.max(np.abs(g1 - rg)) < TOL and np.max(np.abs(f1 - rf)) < TOL
    assert np.max(np.abs(p1 - rf)) < TOL
    assert abs(l1 - l0) < 1e-11 * max(1, B) and np.max(np.abs(g1 - g0)) < 1e-11 and np.max(np.abs(f1 - f0)) < 1e-12


@pytest.mark.parametrize("n,B,S", [(8, 256, 6), (10, 64, 3), (12, 32, 2), (12, 4096, 2)])
def test_complex_column_form_against_the_tape_up_to_the_headline_dimension(n, B, S):
    """d = 256 .. 4096 (every workgroup shape), state form and operator form; the tape is the reference here."""
    from diffquantum_amd.engine import Engine
    dt = 5e-4
    H0, Hs, M = _xy_problem(n, seed=n, extra=n >= 9)
    u = np.random.default_rng(1).uniform(-1, 1, (S, len(Hs)))
    psi0 = problems.random_states(1 << n, B, seed=3)
    with Engine(1 << n, len(Hs)) as e:
        e.set_hamiltonians(H0, Hs)
        e.set_observable(M)
        e.set_option("state_columns", 0)
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("xor_regular") == 2
        e.set_option("xor_regular", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("xor_regular") == 0
    assert abs(l1 - l0) < 1e-11 * B and np.max(np.abs(g1 - g0)) < 1e-11 and np.max(np.abs(f1 - f0)) < 1e-12
    assert np.max(np.abs(g0)) > 1e-6


def test_complex_generators_outside_the_class_keep_the_tape():
    """An entry with both a real and an imaginary part (X_a + 0.5 Y_a in ONE control) is not in the complex table."""
    n = 8
    H0, Hs, M = _xy_problem(n, seed=1, extra=False)
    Hm = Hs.copy()
    Hm[0] = Hm[0] + 0.5 * _cstring(n, {2: "Y"})            # X_2 + 0.5 Y_2: entries 1 -+ 0.5 i
    u = np.random.default_rng(0).uniform(-1, 1, (2, len(Hs)))
    psi0 = problems.random_states(1 << n, 3, seed=1)
    with _engine(H0, Hm, M) as e:
        assert e.info("xor_regular_found") == -1
        e.energy_grad(u, 1e-3, psi0)
        assert e.info("xor_regular") == 0


@pytest.mark.parametrize("B", [4, 256])
def test_complex_generator_with_z_controls(oracle, B):
    """X, Y and Z drives together: controls with diagonal entries ride in the offset-0 part of the complex table (their gradients
    come from a per-row accumulator), one control mixes an XOR-diagonal with diagonal entries (Y_a + 0.5 Z_b)."""
    n, S, dt = 8, 4, 2e-3
    H0, Hs, M = _xy_problem(n, seed=5, extra=False)
    Hs = np.concatenate([Hs, _cstring(n, {2: "Z"})[None], (_cstring(n, {6: "Y"}) + 0.5 * _cstring(n, {1: "Z", 3: "Z"}))[None]])
    u = np.random.default_rng(3).uniform(-1, 1, (S, len(Hs)))
    psi0 = problems.random_states(1 << n, B, seed=2)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M) as e:
        assert e.info("xor_regular_found") >= 9
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("xor_regular") == 2
        e.set_option("xor_regular", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
    assert abs(l1 - rl) < TOL * max(1, B / 8) and np.max(np.abs(g1 - rg)) < TOL and np.max(np.abs(f1 - rf)) < TOL
    assert abs(l1 - l0) < 1e-11 * max(1, B) and np.max(np.abs(g1 - g0)) < 1e-11
    assert np.max(np.abs(rg[:, -2:])) > 1e-5           # (the Z-type controls' gradients are not zeros)


def test_more_than_sixteen_xor_diagonals(oracle):
    """17 .. 32 XOR-diagonals take the reverse kernels with 32 accumulators (8 rows per thread): real generator with X_k and
    X_k X_k+1 controls (18 diagonals), complex generator with X_k and Y_k on every qubit (18 diagonals: full I/Q control)."""
    n, S, B, dt = 9, 3, 6, 1e-3
    rng = np.random.default_rng(12)
    d = 1 << n
    H0 = np.zeros((d, d))
    for a in range(n):
        H0 += rng.standard_normal() * _string(n, {a: "Z"}) + rng.standard_normal() * _string(n, {a: "Z", (a + 1) % n: "Z"})
    Hr = [_string(n, {k: "X"}) for k in range(n)] + [0.5 * _string(n, {k: "X", (k + 1) % n: "X"}) for k in range(n)]
    Hc = [_cstring(n, {k: "X"}) for k in range(n)] + [_cstring(n, {k: "Y"}) for k in range(n)]
    M = rng.standard_normal((d, d))
    M = ((M + M.T) / 2).astype(complex)
    psi0 = problems.random_states(d, B, seed=1)
    for Hs, want in ((np.array(Hr).astype(complex), 1), (np.array(Hc), 2)):
        u = rng.uniform(-1, 1, (S, len(Hs)))
        rl, rg, rf = oracle.energy_grad_one_frechet(H0.astype(complex), list(Hs), M, u, dt, psi0)
        with _engine(H0.astype(complex), Hs, M, **({"sparse_generator": 1} if want == 1 else {})) as e:
            assert e.info("xor_regular_found") == 18
            l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
            assert e.info("xor_regular") == want
        assert abs(l1 - rl) < TOL and np.max(np.abs(g1 - rg)) < TOL and np.max(np.abs(f1 - rf)) < TOL
