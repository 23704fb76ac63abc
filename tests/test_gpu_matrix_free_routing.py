"""GPU parity of the matrix-free routing behind the dense API: `dq_set_hamiltonians` accepts any dimension (diffqc.cc:43-67 puts no
bound on it), and a generator that is a sum of Pauli strings whose dimension the state-column kernels do not take (complex128 above
4096, complex64 above 8192) runs on the matrix-free passes (csrc/structured.hip) INSIDE the same context -- strings read off the
XOR-regular table, no second API for the caller.  Checked against the oracle at d = 512 (the route forced), against the oracle's
committed answer at d = 8192 (tests/d8192_case.py), and against the explicit-propagator chain of the same engine at d = 8192
(complex128) and d = 16384 (complex64).

Tolerance (BASELINE.json north_star): 1e-10 on amplitudes and gradients in fp64, 1e-4 in fp32."""
import numpy as np
import pytest

from diffquantum_amd import problems
from test_gpu_state_columns import CASES, _engine, _qaoa, oracle_answer

pytestmark = pytest.mark.gpu
TOL = 1e-10


@pytest.mark.parametrize("case", ["ising_x", "xy_drives", "qaoa_ring_zz_controls", "mixed_two_body"])
def test_forced_route_matches_oracle_at_dimension_512(oracle, case):
    """n = 9 with option matrix_free = 1: every entry class of the table as strings -- X_k, Y_k (imaginary diagonals), ZZ controls
    (offset-0 diagonals by their Walsh-Hadamard coefficients), two-body strings with row signs, a control with entries of both kinds."""
    n = 9
    H0, Hs, M = _qaoa(n) if case == "qaoa_ring_zz_controls" else CASES[case](n)
    K, d = len(Hs), 1 << n
    S, B, dt = 4, 5, 2e-3
    u = np.random.default_rng(n).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(d, B, seed=2)
    diag_m = not np.any(M - np.diag(np.diag(M)))
    rl, rg, rf = oracle_answer(oracle, (case, n, S, B, dt), H0, Hs, M, u, dt, psi0)      # (the inputs of test_state_columns_match_oracle at n = 9)
    with _engine(H0, Hs, M, matrix_free=1) as e:
        fwd = e.propagate(u, dt, psi0)
        assert e.info("matrix_free") == 1, "a Pauli-string generator was not routed to the matrix-free passes"
        assert e.info("matrix_free_passes") > 0
        loss, grad, fin, en = e.energy_grad(u, dt, psi0)
        # (a general -- non-diagonal -- observable is not a string table here: the gradient call keeps the dense engine's schemes)
        assert e.info("matrix_free") == (1 if diag_m else 0)
    assert np.max(np.abs(fwd - rf)) < TOL and np.max(np.abs(fin - rf)) < TOL
    assert abs(loss - rl) < TOL and abs(float(en.sum()) - rl) < TOL
    assert np.max(np.abs(grad - rg)) < TOL, np.max(np.abs(grad - rg))
    assert np.max(np.abs(rg)) > 1e-5


def test_forced_route_complex64_and_a_long_chain():
    """n = 10, complex64, 300 steps (two checkpoint levels under a small budget): against the complex128 state-column form."""
    n, S, B, dt = 10, 300, 16, 2e-3
    H0, Hs, M = problems.random_ising(n, seed=31)
    u = problems.pulse_table(n, S, S * dt, seed=32)
    psi0 = problems.random_states(1 << n, B, seed=33)
    with _engine(H0, Hs, M) as e:
        rl, rg, rf, _ = e.energy_grad(u, dt, psi0)
        assert e.info("state_columns") == 1 and e.info("matrix_free") == 0
    with _engine(H0, Hs, M, dtype="c64", matrix_free=1, matrix_free_checkpoint_bytes=(1 << n) * B * 8 * 60) as e:
        loss, grad, fin, _ = e.energy_grad(u, dt, psi0.astype(np.complex64))
        assert e.info("matrix_free") == 1 and e.info("matrix_free_two_level") == 1
    assert np.max(np.abs(fin - rf)) < 1e-4 and abs(loss - rl) < 1e-4 * B
    assert np.max(np.abs(grad - rg)) < 1e-4 * np.max(np.abs(rg)) * 10 and np.max(np.abs(rg)) > 1e-4


def test_auto_route_at_dimension_8192_matches_the_oracle_fixture():
    """13 qubits, complex128: chosen automatically (no state-column kernel at this size); one step against the oracle's committed answer
    (scipy expm + expm_frechet at d = 8192, tests/golden/oracle_d8192.npz)."""
    import torch
    import d8192_case as case
    a = case.answer()
    d, dev = 1 << case.N, torch.device("cuda:0")
    H0 = torch.zeros((d, d), dtype=torch.complex128, device=dev)
    H0.diagonal().copy_(torch.from_numpy(case.diag()).to(dev).to(torch.complex128))
    Hs = torch.zeros((case.K, d, d), dtype=torch.complex128, device=dev)
    idx = torch.arange(d, device=dev)
    for k, q in enumerate(case.QUBITS):
        Hs[k, idx, idx ^ (1 << (case.N - 1 - q))] = 1.0
    with _engine(H0, Hs, H0) as e:
        del H0, Hs
        loss, grad, fin, en = e.energy_grad(case.pulse(), case.DT, case.states())
        assert e.info("matrix_free") == 1 and e.info("state_columns") == 0
    assert np.max(np.abs(fin - a["fin"])) < TOL and np.max(np.abs(en - a["energies"])) < TOL
    assert abs(loss - a["loss"]) < TOL and np.max(np.abs(grad - a["grad"])) < TOL, np.max(np.abs(grad - a["grad"]))
    assert np.max(np.abs(a["grad"])) > 1e-5


@pytest.mark.parametrize("dtype,n,tol", [("c128", 13, 1e-10), ("c64", 14, 1e-4)])
def test_auto_route_matches_the_explicit_propagator_chain(dtype, n, tol):
    """d = 8192 in complex128 / d = 16384 in complex64, full Ising + X_k (every qubit driven), 3 steps, 16 states: the routed passes
    against the SAME context with the route switched off (dense step exponentials + complex GEMMs: what these shapes ran before)."""
    import torch
    dev = torch.device("cuda:0")
    S, B, dt = 3, 16, 1e-3
    H0, Hs, M = problems.random_ising_torch(n, 9, dev)
    if dtype == "c64":
        H0, Hs, M = H0.to(torch.complex64), Hs.to(torch.complex64), M.to(torch.complex64)
    u = problems.pulse_table(n, S, S * dt, seed=10)
    psi0 = problems.random_states(1 << n, B, seed=11)
    if dtype == "c64":
        psi0 = psi0.astype(np.complex64)
    with _engine(H0, Hs, M, dtype=dtype) as e:
        del H0, Hs, M
        torch.cuda.empty_cache()
        l1, g1, f1, e1 = e.energy_grad(u, dt, psi0)
        assert e.info("matrix_free") == 1
        fwd = e.propagate(u, dt, psi0)
        assert e.info("matrix_free") == 1
        e.set_option("matrix_free", 0)
        l0, g0, f0, e0 = e.energy_grad(u, dt, psi0)
        assert e.info("matrix_free") == 0 and e.info("state_columns") == 0
    scale = float(np.max(np.abs(g0)))
    assert np.max(np.abs(f1 - f0)) < tol and np.max(np.abs(fwd - f0)) < tol and np.max(np.abs(e1 - e0)) < tol * 10
    assert abs(l1 - l0) < tol * B and np.max(np.abs(g1 - g0)) < (tol if dtype == "c128" else tol * scale * 10), np.max(np.abs(g1 - g0))
    assert scale > 1e-5
