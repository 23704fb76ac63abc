"""GPU parity of the state-column form (diffquantum_amd/csrc/statecol.hip): the chain `psi <- expm(-1j dt H(t)) psi`
(sim_plain.py:135-150, diffqc.cc:190-200) and its adjoint gradient with every state column held in LDS for the whole chain,
for generators that are sums of Pauli strings.  Against the oracle (d = 256 .. 1024 here, d = 4096 in
test_gpu_baseline_configs.py) and against the engine's other schemes (option state_columns = 0) up to d = 4096.

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
    for k, v in opts.items():
        e.set_option(k, v)
    return e


def _xy(n, seed, n_x, n_y):
    import torch
    H0, Hs, M = problems.xy_drives_torch(n, seed, torch.device("cuda:0"), n_x=n_x, n_y=n_y)
    return H0.cpu().numpy(), Hs.cpu().numpy(), M.cpu().numpy()


def _qaoa(n):
    p = problems.qaoa_ring(n)
    return p["H0"].astype(np.complex128), p["Hs"].astype(np.complex128), p["H_cost"].astype(np.complex128)


def _mixed(n, seed):
    """Drift with an off-diagonal term, two-body X X / X Z / Y Z strings (signs, shared offsets), a diagonal control and a
    control with both kinds of entries: every entry class of the table, real and imaginary phases."""
    rng = np.random.default_rng(seed)
    X = np.array([[0, 1], [1, 0]], dtype=complex); Y = np.array([[0, -1j], [1j, 0]]); Z = np.diag([1.0 + 0j, -1.0])

    def s(ops):
        return problems.kron_chain([ops.get(q, np.eye(2)) for q in range(n)])

    H0 = np.diag(problems.ising_diagonal(n, seed)).astype(complex) + 0.7 * s({0: X})
    Hs = [s({1: X}), s({2: X}), s({1: X, 4: X}) + 0.5 * s({1: X, 5: Z}), s({3: Y, 6: Z}), 1.5 * s({3: X, 6: Z}), s({4: Z}),
          s({7: X}) + 0.5 * s({2: Z}), s({n - 1: Y})]
    M = rng.standard_normal((1 << n, 1 << n))
    return H0, np.array(Hs), ((M + M.T) / 2).astype(complex)


_ORACLE_CACHE = {}


def oracle_answer(oracle, key, H0, Hs, M, u, dt, psi0):
    """oracle.energy_grad, computed once per input set of this session (the ZZ-control cases cost 20 s of host time each, and several
    tests -- workgroup shapes, the matrix-free route -- check the same inputs)."""
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    return _ORACLE_CACHE[key]


CASES = {
    "ising_x": lambda n: problems.random_ising(n, 4),
    "xy_drives": lambda n: _xy(n, 5, n // 2, n - n // 2),
    "mixed_two_body": lambda n: _mixed(n, 6),
}


@pytest.mark.parametrize("workgroups", [0, 2], ids=["one_column_per_workgroup", "two_workgroups"])
@pytest.mark.parametrize("case", list(CASES) + ["qaoa_ring_zz_controls"])
@pytest.mark.parametrize("n", [8, 9])
def test_state_columns_match_oracle(oracle, case, n, workgroups):
    """d = 256 (one wave per column, 4 rows per lane) and 512 (8 rows per lane): forward amplitudes, loss, dL/du vs the oracle.
    `two_workgroups`: the 5 columns on 2 workgroups (3 + 2: the several-columns-per-workgroup loop of every kernel family -- sums
    accumulated across a workgroup's columns, images reused, batch not a multiple of the workgroup count)."""
    H0, Hs, M = _qaoa(n) if case == "qaoa_ring_zz_controls" else CASES[case](n)
    K, d = len(Hs), 1 << n
    S, B, dt = 4, 5, 2e-3
    u = np.random.default_rng(n).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(d, B, seed=2)
    rl, rg, rf = oracle_answer(oracle, (case, n, S, B, dt), H0, Hs, M, u, dt, psi0)
    with _engine(H0, Hs, M, state_columns=1 if n == 8 else -1, state_columns_workgroups=workgroups) as e:     # (auto takes the form from dim 512 up)
        loss, grad, fin, en = e.energy_grad(u, dt, psi0)
        assert e.info("state_columns") == 1, "a Pauli-string generator did not take the state-column form"
        # every table class -- unsigned, parity-signed / complex (X / Y drives, two-body strings), diagonal controls (ZZ-edge QAOA) -- runs
        # the quadrature form of the adjoint sweep (d = 512: the difference-basis kernels with u_2 in LDS; d = 256: the node basis)
        # (the 4-point rule serves steps up to theta = 0.053 at its 2^-40 bound: the ZZ-control ring at n = 9 has theta = 0.07 here)
        assert e.info("state_columns_quad_nodes") == 2 or e.info("theta") > 0.053, "the quadrature form of the adjoint sweep did not run"
        if case in ("xy_drives", "mixed_two_body") or n == 8:
            assert e.info("state_columns_quad_nodes") == 2
        fwd = e.propagate(u, dt, psi0)
        assert e.info("state_columns") == 1
        e.set_option("state_columns", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("state_columns") == 0
    assert np.max(np.abs(fin - rf)) < TOL and np.max(np.abs(fwd - rf)) < TOL
    assert abs(loss - rl) < TOL and abs(float(en.sum()) - rl) < TOL
    assert np.max(np.abs(grad - rg)) < TOL, np.max(np.abs(grad - rg))
    assert np.max(np.abs(grad - g0)) < TOL and abs(loss - l0) < TOL and np.max(np.abs(fin - f0)) < TOL
    assert np.max(np.abs(rg)) > 1e-5


@pytest.mark.parametrize("n,B", [(10, 7), (11, 3), (12, 300)])
def test_state_columns_match_the_other_schemes_at_every_dimension(n, B):
    """d = 1024, 2048, 4096 (several columns per workgroup at d = 4096: 300 columns on 256 workgroups) against the engine's
    explicit-propagator schemes, and a negative dt."""
    H0, Hs, M = problems.random_ising(n, seed=n)
    S = 3
    u = np.random.default_rng(n).uniform(-1, 1, (S, n))
    psi0 = problems.random_states(1 << n, B, seed=4)
    for dt in (1.5e-3, -1e-3):
        with _engine(H0, Hs, M) as e:
            l1, g1, f1, e1 = e.energy_grad(u, dt, psi0)
            assert e.info("state_columns") == 1 and e.info("xor_regular") == 1
            e.set_option("state_columns", 0)
            l0, g0, f0, e0 = e.energy_grad(u, dt, psi0)
            assert e.info("state_columns") == 0
        assert np.max(np.abs(f1 - f0)) < 1e-12 and abs(l1 - l0) < 1e-11 * B
        assert np.max(np.abs(e1 - e0)) < 1e-12
        assert np.max(np.abs(g1 - g0)) < 1e-11 * max(1, B / 16), np.max(np.abs(g1 - g0))


def test_state_columns_dimension_1024_matches_oracle(oracle):
    n, S, B, dt = 10, 3, 4, 2e-3
    H0, Hs, M = _xy(n, 9, 5, 5)
    u = np.random.default_rng(3).uniform(-1, 1, (S, n))
    psi0 = problems.random_states(1 << n, B, seed=6)
    rl, rg, rf = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M) as e:
        loss, grad, fin, _ = e.energy_grad(u, dt, psi0)
        assert e.info("state_columns") == 1 and e.info("xor_regular") == 2
    assert np.max(np.abs(fin - rf)) < TOL and abs(loss - rl) < TOL and np.max(np.abs(grad - rg)) < TOL


def test_long_chain_reversal_drift(oracle):
    """120 steps at d = 256: the backward sweep re-derives psi_t = U_t^H psi_{t+1} all the way down; gradient vs the oracle
    (one expm_frechet per step: the oracle's cost is what bounds the chain length here; bench.py's full 2000-step chain carries a
    directional-derivative check at full size)."""
    n, S, B = 8, 120, 3
    H0, Hs, M = problems.random_ising(n, seed=2)
    u = problems.pulse_table(n, S, 0.4, seed=3)
    psi0 = problems.random_states(1 << n, B, seed=7)
    dt = 0.4 / S
    rl, rg, rf = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, state_columns=1) as e:
        loss, grad, fin, _ = e.energy_grad(u, dt, psi0)
        assert e.info("state_columns") == 1
    assert np.max(np.abs(fin - rf)) < TOL and abs(loss - rl) < TOL and np.max(np.abs(grad - rg)) < TOL


def test_fidelity_objective_and_fp32(oracle):
    n, S, B, dt = 8, 5, 6, 3e-3
    H0, Hs, M = _xy(n, 3, 4, 4)
    u = np.random.default_rng(5).uniform(-1, 1, (S, n))
    psi0 = problems.random_states(1 << n, B, seed=8)
    tg = problems.random_states(1 << n, B, seed=9)
    rl, rg = oracle.fidelity_grad(H0, list(Hs), u, dt, psi0, tg)[:2]
    with _engine(H0, Hs, M, state_columns=1) as e:
        out = e.fidelity_grad(u, dt, psi0, tg)
        assert e.info("state_columns") == 1
    assert abs(out[0] - rl) < TOL and np.max(np.abs(out[1] - rg)) < TOL
    l64, g64, f64 = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0.astype(np.complex64), Hs.astype(np.complex64), M.astype(np.complex64), dtype="c64", state_columns=1) as e:
        loss, grad, fin, _ = e.energy_grad(u, dt, psi0.astype(np.complex64))
        assert e.info("state_columns") == 1
    assert np.max(np.abs(fin - f64)) < 1e-4 and abs(loss - l64) < 1e-4 * B and np.max(np.abs(grad - g64)) < 1e-4


def test_options_that_pin_another_scheme_keep_it_and_wide_batches_leave():
    n = 9
    H0, Hs, M = problems.random_ising(n, seed=1)
    u = np.random.default_rng(0).uniform(-1, 1, (2, n))
    psi0 = problems.random_states(1 << n, 4, seed=1)
    for opt, val in (("compose", 1), ("compose", 0), ("real_path", 0), ("real_path", 2), ("xor_regular", 0), ("sparse_generator", 0),
                     ("force_degree", 9)):
        with _engine(H0, Hs, M, **{opt: val}) as e:
            e.energy_grad(u, 1e-3, psi0)
            assert e.info("state_columns") == 0, opt
    with _engine(H0, Hs, M) as e:
        e.energy_grad(u, 1e-3, psi0)
        assert e.info("state_columns") == 1               # auto: dim >= 512, batch <= 8 x dim
    H8, Hs8, M8 = problems.random_ising(8, seed=1)
    with _engine(H8, Hs8, M8) as e:
        e.energy_grad(u[:, :8], 1e-3, problems.random_states(256, 4, seed=1))
        assert e.info("state_columns") == 0               # dim 256: one wave per column; the operator form is faster there
    wide = problems.random_states(1 << n, 8 * 512 + 1, seed=1)
    with _engine(H0, Hs, M) as e:
        e.energy_grad(u, 1e-3, wide)
        assert e.info("state_columns") == 0               # batch > 8 x dim: the operator form's cost does not grow with the batch
        e.set_option("state_columns", 1)
        e.energy_grad(u, 1e-3, wide)
        assert e.info("state_columns") == 1
    Hd, Hsd, Md = problems.random_real_symmetric_problem(8, 2, seed=0)
    with _engine(Hd, Hsd, Md, state_columns=1) as e:
        e.energy_grad(u[:, :2], 1e-3, problems.random_states(256, 4, seed=1))
        assert e.info("state_columns") == 0               # dense H: no table, the GEMM chain stays


@pytest.mark.parametrize("n", [8, 9, 10])
def test_wide_workgroup_variant_matches_oracle(oracle, n):
    """`state_columns_variant` bit 0: 256 threads per column with 1 / 2 / 4 rows per thread at d = 256 / 512 / 1024 (instead of one
    or two waves with 4-8 rows): the same recurrences on another workgroup shape (X/Y drives and a Z control: signs, both phases,
    a diagonal control), against the oracle."""
    H0, Hs, M = _mixed(n, seed=n)
    K, d = len(Hs), 1 << n
    S, B, dt = 3, 5, 2e-3
    u = np.random.default_rng(n).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(d, B, seed=2)
    rl, rg, rf = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, state_columns=1, state_columns_variant=1) as e:
        loss, grad, fin, en = e.energy_grad(u, dt, psi0)
        fwd = e.propagate(u, dt, psi0)
        assert e.info("state_columns") == 1
    assert np.max(np.abs(fin - rf)) < TOL and np.max(np.abs(fwd - rf)) < TOL and abs(loss - rl) < TOL
    assert np.max(np.abs(grad - rg)) < TOL, np.max(np.abs(grad - rg))


def test_more_than_sixteen_xor_diagonals_on_state_columns(oracle):
    """17 .. 32 XOR-diagonals: the adjoint kernel with two accumulator vectors.  A real generator with X_k and X_k X_k+1 controls
    (18 diagonals, shared low / high offsets) and full I/Q control (X_k and Y_k on every qubit: 18 diagonals, both phases, sign
    words), at d = 512, against the oracle."""
    n, S, B, dt = 9, 3, 6, 1e-3
    rng = np.random.default_rng(12)
    d = 1 << n
    X = np.array([[0, 1], [1, 0]], dtype=complex); Y = np.array([[0, -1j], [1j, 0]]); Z = np.diag([1.0 + 0j, -1.0])

    def s(ops):
        return problems.kron_chain([ops.get(q, np.eye(2)) for q in range(n)])

    H0 = np.diag(problems.ising_diagonal(n, 3)).astype(complex)
    Hr = [s({k: X}) for k in range(n)] + [0.5 * s({k: X, (k + 1) % n: X}) for k in range(n)]
    Hc = [s({k: X}) for k in range(n)] + [s({k: Y}) for k in range(n)]
    M = rng.standard_normal((d, d))
    M = ((M + M.T) / 2).astype(complex)
    psi0 = problems.random_states(d, B, seed=1)
    for Hs, want in ((np.array(Hr), 1), (np.array(Hc), 2)):
        u = rng.uniform(-1, 1, (S, len(Hs)))
        rl, rg, rf = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, dt, psi0)
        with _engine(H0, Hs, M) as e:
            assert e.info("xor_regular_found") == 18
            loss, grad, fin, _ = e.energy_grad(u, dt, psi0)
            assert e.info("state_columns") == 1 and e.info("xor_regular") == want
        assert abs(loss - rl) < TOL and np.max(np.abs(grad - rg)) < TOL and np.max(np.abs(fin - rf)) < TOL
        assert np.max(np.abs(rg)) > 1e-5


def test_thirteen_qubits_in_complex64():
    """d = 8192 exists in complex64 only (a column is 64 KB there; two images fit the LDS): n = 13 Ising drift + X_k, against the
    engine's explicit-propagator scheme in the same precision (the oracle's expm at d = 8192 is out of reach of a test), 1e-4."""
    import torch
    from diffquantum_amd.engine import Engine
    n, S, B, dt = 13, 2, 24, 1e-3
    d = 1 << n
    dev = torch.device("cuda:0")
    diag = torch.from_numpy(problems.ising_diagonal(n, 2)).to(dev)
    H0 = torch.zeros((d, d), dtype=torch.complex64, device=dev)
    H0.diagonal().copy_(diag.to(torch.complex64))
    Hs = torch.zeros((n, d, d), dtype=torch.complex64, device=dev)
    idx = torch.arange(d, device=dev)
    for q in range(n):
        Hs[q, idx, idx ^ (1 << (n - 1 - q))] = 1.0
    u = np.random.default_rng(4).uniform(-1, 1, (S, n))
    psi0 = torch.from_numpy(problems.random_states(d, B, seed=5).astype(np.complex64)).to(dev)
    with Engine(d, n, dtype="c64") as e:
        e.set_hamiltonians(H0, Hs)
        e.set_observable(H0)
        del Hs
        assert e.info("xor_regular_found") == n
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("state_columns") == 1
        e.set_option("state_columns", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("state_columns") == 0
    gs = float(np.max(np.abs(g0)))
    assert gs > 1e-4                                                      # (gradients of this problem are ~5e-4)
    assert float((f1 - f0).abs().max()) < 1e-5 and abs(l1 - l0) < 1e-4 * B
    assert np.max(np.abs(g1 - g0)) < 1e-3 * gs, (np.max(np.abs(g1 - g0)), gs)   # fp32: both schemes carry ~1e-6 relative rounding


# ---- the chain length BASELINE states: S = 2000 (tests/s2000_cases.py; fixture tests/golden/oracle_s2000.npz) --------------------
@pytest.mark.parametrize("adjoint", [1, 0], ids=["quadrature", "exact_polynomial"])
@pytest.mark.parametrize("case,variant", [("ising8", 0), ("ising8", 1), ("ising9", 0), ("ising9", 1), ("xyz9", 0), ("xyz9", 1)])
def test_full_2000_step_chain_matches_the_oracle_fixture(case, variant, adjoint):
    """The whole 2000-step chain (configs[2]'s step count; reference loop diffqc.cc:190-200) on state columns against the oracle's
    answer (scipy expm + expm_frechet per step, committed fixture): final amplitudes, energies, loss and dL/du at 1e-10 -- both
    workgroup shapes (variant 1: 256 threads per column) and both forms of the adjoint sweep.  The sweep re-derives psi_t from
    psi_{t+1} over all 2000 steps without a checkpoint: this is the check that the drift stays inside the tolerance."""
    import s2000_cases
    P, a = s2000_cases.answer(case)
    with _engine(P["H0"], P["Hs"], P["M"], state_columns=1, state_columns_variant=variant, state_columns_adjoint=adjoint) as e:
        loss, grad, fin, en = e.energy_grad(P["u"], P["dt"], P["psi0"])
        assert e.info("state_columns") == 1
        nodes = e.info("state_columns_quad_nodes")
        # (round 5: signed / complex tables and diagonal controls -- xyz9 -- run the quadrature form too)
        assert nodes == (0 if adjoint == 0 else 2), "the requested form of the adjoint sweep did not run"
        fwd = e.propagate(P["u"], P["dt"], P["psi0"])
    assert np.max(np.abs(fin - a["fin"])) < TOL and np.max(np.abs(fwd - a["fin"])) < TOL
    assert np.max(np.abs(en - a["energies"])) < TOL and abs(loss - a["loss"]) < TOL
    assert np.max(np.abs(grad - a["grad"])) < TOL, np.max(np.abs(grad - a["grad"]))
    assert np.max(np.abs(a["grad"])) > 1e-4


@pytest.mark.parametrize("adjoint", [1, 0], ids=["quadrature", "exact_polynomial"])
@pytest.mark.parametrize("dtype,variant,tol", [("c128", 0, 1e-12), ("c128", 1, 1e-12), ("c64", 0, 2e-5)])
def test_several_columns_per_workgroup_and_segmented_sweep(dtype, variant, tol, adjoint):
    """d = 1024 / 512: 7 columns on 3 workgroups (3 + 2 + 2) and the adjoint sweep split into segments of a few steps (scratch budget of
    its partial sums) -- against the same engine with one column per workgroup and the chain in one launch: same arithmetic per column
    and step, so the states are identical and the sums differ by their order only."""
    n, S, B, dt = (10 if variant == 0 else 9), 23, 7, 2e-3
    H0, Hs, M = problems.random_ising(n, seed=12)
    u = np.random.default_rng(4).uniform(-1, 1, (S, n))
    psi0 = problems.random_states(1 << n, B, seed=5)
    if dtype == "c64":
        psi0 = psi0.astype(np.complex64)
    with _engine(H0, Hs, M, dtype=dtype, state_columns=1, state_columns_variant=variant, state_columns_adjoint=adjoint) as e:
        l0, g0, f0, e0 = e.energy_grad(u, dt, psi0)
        assert e.info("state_columns") == 1 and e.info("state_columns_segments") == 1
        e.set_option("state_columns_workgroups", 3)
        waves = (1 << n) // 8 // 64 if variant == 0 else 4
        e.set_option("state_columns_part_bytes", 8 * 3 * max(1, waves) * 16 * 5)     # room for 5 steps of 3 workgroups' sums
        l1, g1, f1, e1 = e.energy_grad(u, dt, psi0)
        assert e.info("state_columns_segments") == 5
    assert np.array_equal(f0, f1) and np.max(np.abs(e0 - e1)) <= tol
    assert abs(l0 - l1) <= tol * B and np.max(np.abs(g0 - g1)) <= tol * B, np.max(np.abs(g0 - g1))
    assert np.max(np.abs(g0)) > 1e-5


@pytest.mark.parametrize("adjoint", [1, 0], ids=["quadrature", "exact_polynomial"])
@pytest.mark.parametrize("case,n,dt", [("ising_x", 9, 2e-3), ("ising_x", 9, 2e-2), ("xy_drives", 9, 3e-3), ("mixed_two_body", 8, 4e-3),
                                       ("qaoa_ring_zz_controls", 8, 2e-2)])
def test_economized_step_polynomial_against_taylor_and_the_oracle(oracle, case, n, dt, adjoint):
    """The default step polynomial is the Chebyshev-economized one (dq_step_polynomial; csrc/statecol.hip xs_poly_make): at these step
    sizes one to three sweeps fewer per step than the Taylor polynomial of the same tolerance (option state_columns_polynomial = 0),
    and both chains and both adjoint forms -- the exact form's Horner levels and the quadrature form's node sums carry the level
    factors -- match the oracle (scipy expm / expm_frechet) to the same 1e-10."""
    H0, Hs, M = _qaoa(n) if case == "qaoa_ring_zz_controls" else CASES[case](n)
    K, d = len(Hs), 1 << n
    S, B = 5, 6
    u = np.random.default_rng(40 + n).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(d, B, seed=41)
    rl, rg, rf = oracle_answer(oracle, ("econ", case, n, dt), H0, Hs, M, u, dt, psi0)
    out = {}
    with _engine(H0, Hs, M, state_columns=1, state_columns_adjoint=adjoint) as e:
        for poly in (1, 0):
            e.set_option("state_columns_polynomial", poly)
            loss, grad, fin, en = e.energy_grad(u, dt, psi0)
            assert e.info("state_columns") == 1
            out[poly] = (loss, grad, fin, e.info("degree"), e.info("state_columns_taylor_degree"), e.propagate(u, dt, psi0), e.info("degree"))
    assert out[0][3] == out[0][4] == out[1][4], "Taylor: the degree of the remainder bound"
    # the host-only entry of the C-ABI names the same polynomials for the step bound the engine reports
    import ctypes as C
    from diffquantum_amd import _lib
    with _engine(H0, Hs, M, state_columns=1) as e2:
        e2.propagate(u, dt, psi0)
        theta = e2.info("theta")
    L1, L0, g = C.c_int(), C.c_int(), (C.c_double * 25)()
    assert _lib.load().dq_step_polynomial(theta, 2.0 ** -53, 1, C.byref(L1), g) == 0 and L1.value == out[1][3]
    assert _lib.load().dq_step_polynomial(theta, 2.0 ** -53, 0, C.byref(L0), g) == 0 and L0.value == out[0][3]
    assert out[1][3] <= out[0][3] and out[1][6] == out[1][3], (out[1][3], out[0][3])
    if dt >= 1e-2:
        assert out[1][3] <= out[0][3] - 1 and e is not None, (out[1][3], out[0][3])     # (theta ~ 0.2 .. 0.4: two degrees fewer)
    for poly in (1, 0):
        loss, grad, fin, _, _, fwd, _ = out[poly]
        assert np.max(np.abs(fin - rf)) < TOL and np.max(np.abs(fwd - rf)) < TOL and abs(loss - rl) < TOL
        assert np.max(np.abs(grad - rg)) < TOL, (poly, np.max(np.abs(grad - rg)))
    assert np.max(np.abs(out[1][1] - out[0][1])) < 1e-12 * max(1.0, np.max(np.abs(rg))) and np.max(np.abs(out[1][2] - out[0][2])) < 1e-13
    assert np.max(np.abs(rg)) > 1e-5


def test_economized_step_polynomial_in_complex64():
    """fp32 tolerance (6e-8 per step): the step size is set from the engine's own bound so that theta = 0.12 -- Taylor takes degree 5
    there (theta^6 / 720), the economized polynomial 4 (2 J_5(theta)); against the complex128 state-column chain.  A long chain
    tightens the per-step bound (state_columns_chain_tol / steps): the same step over 2000 steps takes one degree more (and Taylor's degree 5 already meets it)."""
    n, S, B = 10, 40, 9
    H0, Hs, M = problems.random_ising(n, seed=44)
    u = problems.pulse_table(n, S, 1.0, seed=45)
    psi0 = problems.random_states(1 << n, B, seed=46)
    with _engine(H0, Hs, M) as e:
        e.propagate(u, 1e-3, psi0)
        dt = 1e-3 * 0.12 / e.info("theta")
        rl, rg, rf, _ = e.energy_grad(u, dt, psi0)
    with _engine(H0, Hs, M, dtype="c64") as e:
        loss, grad, fin, _ = e.energy_grad(u, dt, psi0.astype(np.complex64))
        assert e.info("state_columns") == 1 and abs(e.info("theta") - 0.12) < 1e-6
        assert (e.info("degree"), e.info("state_columns_taylor_degree")) == (4, 5)
        e.propagate(np.tile(u, (50, 1)), dt, psi0.astype(np.complex64))
        assert (e.info("degree"), e.info("state_columns_taylor_degree")) == (5, 5)     # (1e-8 per step: 2 J_5 = 2.6e-8 is too much, theta^6 / 720 = 4e-9 is not)
    assert np.max(np.abs(fin - rf)) < 1e-4 and abs(loss - rl) < 1e-4 * B
    assert np.max(np.abs(grad - rg)) < 1e-4 * max(1.0, np.max(np.abs(rg))) and np.max(np.abs(rg)) > 1e-4


@pytest.mark.parametrize("case,n,S,B", [("qaoa_ring_zz_controls", 8, 203, 600), ("ising_x", 9, 41, 1100), ("xy_drives", 9, 7, 700),
                                        ("mixed_two_body", 8, 1, 300)])
def test_composed_forward_pass_from_state_column_segments(oracle, case, n, S, B):
    """Batch wider than the dimension: the forward pass applies the PRODUCT of the chain's propagators once (sim_plain.py:135-150 /
    diffqc.cc:190-200 chained).  For a Pauli-sum generator that product comes from the state-column forward kernel run on the
    identity's columns, cut into time segments that are multiplied afterwards (engine.hip xs_compose_segments; BASELINE configs[1]'s
    shape) -- against the explicit-propagator tree of the same engine and, for four columns, against the oracle's chain."""
    H0, Hs, M = _qaoa(n) if case == "qaoa_ring_zz_controls" else CASES[case](n)
    K, d = len(Hs), 1 << n
    dt = 1.5e-3
    u = np.random.default_rng(50 + S).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(d, B, seed=51)
    with _engine(H0, Hs, M) as e:
        if d > 256:
            e.set_option("compose", 1)       # (d = 256 composes by itself; above that the engine weighs it against sweeping the B states)
        f1 = e.propagate(u, dt, psi0)
        segs = e.info("state_columns_compose_segments")
        assert e.info("composed") == 1 and segs >= 1 and segs <= S, "the composed forward pass did not take the state-column product"
        assert e.info("state_columns") == 0                  # (the STATES are applied by one product, not swept)
        e.set_option("state_columns_compose", 0)
        f0 = e.propagate(u, dt, psi0)
        assert e.info("state_columns_compose_segments") == 0 and e.info("composed") == 1
    assert np.max(np.abs(f1 - f0)) < 1e-11, np.max(np.abs(f1 - f0))
    ref = oracle.propagate(H0, list(Hs), u, dt, psi0[:, :4])
    assert np.max(np.abs(f1[:, :4] - ref)) < TOL
    assert np.max(np.abs(np.linalg.norm(f1, axis=0) - 1.0)) < 1e-12


def test_wide_batches_take_the_composed_form_when_it_is_cheaper():
    """d = 512: 1100 states over 300 steps cost more swept one by one than the product of the chain from the identity's 512 columns + one
    application; over 10 steps they do not.  Same numbers either way."""
    n, B, dt = 9, 1100, 1.5e-3
    H0, Hs, M = problems.random_ising(n, 4)
    psi0 = problems.random_states(1 << n, B, seed=61)
    with _engine(H0, Hs, M) as e:
        for S, composed in ((300, 1), (10, 0)):
            u = np.random.default_rng(S).uniform(-1, 1, (S, n))
            f1 = e.propagate(u, dt, psi0)
            assert e.info("composed") == composed and (e.info("state_columns_compose_segments") > 0) == bool(composed)
            assert e.info("state_columns") == 1 - composed
            e.set_option("state_columns_compose", 0)
            f0 = e.propagate(u, dt, psi0)
            assert e.info("state_columns") == 1 and e.info("composed") == 0        # (without the composed form: the states are swept)
            e.set_option("state_columns_compose", -1)
            assert np.max(np.abs(f1 - f0)) < 1e-11


@pytest.mark.parametrize("case,n,S,B,dtype,tol", [("ising_x", 9, 40, 1100, "c128", 1e-10), ("qaoa_ring_zz_controls", 8, 25, 700, "c128", 1e-10),
                                                   ("xy_drives", 9, 12, 600, "c128", 1e-10), ("mixed_two_body", 8, 6, 300, "c128", 1e-10),
                                                   ("ising_x", 9, 40, 1100, "c64", 2e-4)])
def test_wide_batch_energy_gradient_on_dim_columns(oracle, case, n, S, B, dtype, tol):
    """Batch wider than the dimension, energy objective (option state_columns_wide_adjoint): L = tr(M P R P^H) with R = Psi0 Psi0^H and P
    the chain's product, so the adjoint sweep runs on the dim columns of (P R, 2 M P) instead of the B states (engine.hip
    do_energy_grad_xs_wide) -- loss, dL/du, per-state energies and final states against the sweep over the states themselves, and
    dL/du against the oracle on a slice of the batch where the batch is additive."""
    H0, Hs, M = _qaoa(n) if case == "qaoa_ring_zz_controls" else CASES[case](n)
    K, d = len(Hs), 1 << n
    dt = 2e-3
    u = np.random.default_rng(70 + S).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(d, B, seed=71)
    if dtype == "c64":
        psi0 = psi0.astype(np.complex64)
    with _engine(H0, Hs, M, dtype=dtype, state_columns=1, state_columns_wide_adjoint=0) as e:
        l0, g0, f0, e0 = e.energy_grad(u, dt, psi0)
        assert e.info("state_columns") == 1 and e.info("state_columns_wide_adjoint") == 0
        e.set_option("state_columns", -1)
        e.set_option("state_columns_wide_adjoint", 1)
        l1, g1, f1, e1 = e.energy_grad(u, dt, psi0)
        assert e.info("state_columns_wide_adjoint") == 1 and e.info("state_columns_compose_segments") >= 1
        l2, g2, _, _ = e.energy_grad(u, dt, psi0, want_final=False, want_energies=False)     # (no application of P to the batch)
        assert e.info("state_columns_wide_adjoint") == 1
        # the oracle on a slice: loss and gradient are sums over the states, so the wide form on the batch WITHOUT three of its states
        # (still wider than the dimension) must differ from the full batch's by exactly the oracle's answer for those three
        NS = 3
        l3, g3, _, _ = e.energy_grad(u, dt, np.ascontiguousarray(psi0[:, NS:]), want_final=False, want_energies=False)
        assert e.info("state_columns_wide_adjoint") == 1
    ls, gs, fs = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0[:, :NS].astype(np.complex128))
    otol = tol * max(1.0, float(np.max(np.abs(g2))))   # (a difference of two sums over B states: rounding of the sums, not of the slice)
    assert abs((l2 - l3) - ls) < otol * 10 and np.max(np.abs((g2 - g3) - gs)) < otol, (abs((l2 - l3) - ls), np.max(np.abs((g2 - g3) - gs)))
    assert np.max(np.abs(f1[:, :NS] - fs)) < tol * 10
    scale = max(1.0, float(np.max(np.abs(g0))))
    assert abs(l1 - l0) < tol * B and abs(l2 - l0) < tol * B
    assert np.max(np.abs(g1 - g0)) < tol * scale and np.max(np.abs(g2 - g0)) < tol * scale, np.max(np.abs(g1 - g0))
    assert np.max(np.abs(f1 - f0)) < tol * 10 and np.max(np.abs(e1 - e0)) < tol * 100
    assert np.max(np.abs(g0)) > 1e-5 and np.max(np.abs(gs)) > 1e-7


@pytest.mark.parametrize("case,n", [("qaoa_ring_zz_controls", 9), ("qaoa_ring_zz_controls", 8), ("xy_drives", 9), ("ising_x", 10)])
def test_centred_diagonal_matches_the_uncentred_kernels_and_the_oracle(oracle, case, n):
    """Round 5: the kernels run on the diagonal minus the midpoint of its range (exp(A) = e^{-ic} exp(A + ic): a lower step bound,
    for the reference demo's drift sum_e (I - Z_i Z_j) >= 0 half of it) and the launches restore the scalar.  Same amplitudes, loss
    and gradient as with state_columns_center = 0 and as the oracle, over a chain long enough for the scalar to wind around (the
    adjoint sweep in 3 launches: every launch applies its own segment's scalar), forward pass and composed product included."""
    H0, Hs, M = _qaoa(n) if case == "qaoa_ring_zz_controls" else CASES[case](n)
    K, d = len(Hs), 1 << n
    S, B, dt = 30, 5, 2.5e-2 if case != "ising_x" else 4e-3
    dt = dt / (1.0 + 0.25 * K)
    u = np.random.default_rng(n + 7).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(d, B, seed=9)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    out = {}
    with _engine(H0, Hs, M, state_columns=1) as e:
        e.set_option("state_columns_part_bytes", 14000.0)   # (ten steps of partial sums per launch: 1280 B per step for 5 columns)
        for center in (1, 0):
            e.set_option("state_columns_center", center)
            loss, grad, fin, en = e.energy_grad(u, dt, psi0)
            assert e.info("state_columns") == 1 and e.info("state_columns_centered") == center
            th, deg, segs = e.info("theta"), e.info("degree"), e.info("state_columns_segments")
            fwd = e.propagate(u, dt, psi0)
            out[center] = (loss, grad, fin, fwd, th, deg, segs)
        e.set_option("state_columns_center", 1)
        e.set_option("state_columns", -1)
        e.set_option("compose", 1)
        comp = e.propagate(u, dt, psi0)          # the product of the chain from state-column segments, each with its own scalar
        assert e.info("state_columns_compose_segments") >= 1 and e.info("state_columns_centered") == 1
    (l1, g1, f1, w1, th1, L1, seg1), (l0, g0, f0, w0, th0, L0, _) = out[1], out[0]
    assert th1 <= th0 * (1 + 1e-12) and L1 <= L0 and seg1 >= 2      # (the adjoint sweep ran in several launches, each with its own scalar)
    if case == "qaoa_ring_zz_controls":
        assert th1 < 0.8 * th0, (th1, th0)       # the drift's range [0, 2 n] centred: half its share of the bound
    for (l, g, f, w) in ((l1, g1, f1, w1), (l0, g0, f0, w0)):
        assert abs(l - rl) < TOL * B and np.max(np.abs(g - rg)) < TOL and np.max(np.abs(f - rf)) < TOL and np.max(np.abs(w - rf)) < TOL
    assert np.max(np.abs(comp - rf)) < TOL
    assert np.max(np.abs(rg)) > 1e-5


@pytest.mark.parametrize("case,n", [("ising_x", 9), ("xy_drives", 9), ("qaoa_ring_zz_controls", 8)])
def test_complex64_takes_the_quadrature_form_at_moderate_steps(oracle, case, n):
    """the rule's bound defaults to fp32 rounding in complex64 (1e-7 instead of 2^-40), so the 2L-sweep form serves steps up
    to theta ~ 0.3 there instead of 0.05 -- here theta ~ 0.15: the quadrature form runs (node basis: fp32 vectors fit the registers for
    every table class) and agrees with the oracle at the fp32 tolerance."""
    H0, Hs, M = _qaoa(n) if case == "qaoa_ring_zz_controls" else CASES[case](n)
    K, d = len(Hs), 1 << n
    S, B = 6, 5
    u = np.random.default_rng(n + 3).uniform(-1, 1, (S, K))
    n1 = lambda A: np.abs(A).sum(axis=0).max()
    dt = 0.15 / (n1(H0) + sum(n1(h) for h in Hs))
    psi0 = problems.random_states(d, B, seed=2)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, dtype="c64", state_columns=1) as e:
        loss, grad, fin, en = e.energy_grad(u, dt, psi0.astype(np.complex64))
        assert e.info("state_columns") == 1 and e.info("state_columns_quad_nodes") == 2, (e.info("theta"), e.info("state_columns_quad_nodes"))
        e.set_option("state_columns_adjoint", 0)
        l0, g0, _, _ = e.energy_grad(u, dt, psi0.astype(np.complex64))
        assert e.info("state_columns_quad_nodes") == 0
    scale = max(1.0, float(np.max(np.abs(rg))))
    assert np.max(np.abs(fin - rf)) < 2e-5 and abs(loss - rl) < 1e-4 * B
    assert np.max(np.abs(grad - rg)) < 1e-4 * scale and np.max(np.abs(g0 - rg)) < 1e-4 * scale
