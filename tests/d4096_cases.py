"""Inputs of the three d = 4096 oracle comparisons of tests/test_gpu_baseline_configs.py, and their oracle answers.

An oracle evaluation at d = 4096 (scipy expm + expm_frechet per step, complex128) is 35 - 80 s of host time, three of them a third of
the GPU suite.  The answers are therefore ALSO kept as a fixture, tests/golden/oracle_d4096.npz, written by
tests/golden/make_oracle_d4096_goldens.py (which calls oracle.energy_grad_one_frechet on exactly these inputs; --check recomputes
and compares).  `answer(case, oracle)` returns the stored answer when the fixture holds one for identical inputs (same pulse table
and the same checksums of psi0 and of the Hamiltonians) and otherwise computes it with the oracle on the spot.
The fixture keeps loss, gradient, the energy <psi_b|M|psi_b> of EVERY final column and the amplitudes of the FIRST FOUR columns."""
import os

import numpy as np

from diffquantum_amd import problems

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_d4096.npz")
FIN_COLS = 4


def ising12_inputs():
    """BASELINE's workload (random Ising drift stored dense, X_k controls, M = H0): 2 steps, 32 states."""
    n, S, B, dt = 12, 2, 32, 1e-3
    H0, Hs, M = problems.random_ising(n, seed=0)
    u = problems.pulse_table(n, S, 2000 * dt, seed=0)[:S].copy()      # the first steps of the 2000-step pulse
    u[1] += 0.37 * np.random.default_rng(5).uniform(-1, 1, n)        # (bump rows 0..1 are nearly equal: make them differ)
    psi0 = problems.random_states(1 << n, B, seed=1)
    return dict(H0=H0, Hs=Hs, M=M, u=u, psi0=psi0, dt=dt, B=B)


def xyz12_inputs():
    """d = 4096 with X, Y and Z drives (a COMPLEX Hermitian Pauli sum: I/Q control plus a detuning): random Ising drift, X_q on
    five qubits, Y_q on five others, one Z_q control and one two-body X_a Z_b control; 2 steps, 16 states."""
    n, S, B, dt = 12, 2, 16, 1e-3
    d = 1 << n
    H0 = np.diag(problems.ising_diagonal(n, 7)).astype(np.complex128)
    idx = np.arange(d)
    bit = lambda q: 1 << (n - 1 - q)
    sgn = lambda q: 1.0 - 2.0 * ((idx & bit(q)) != 0)
    Hs = np.zeros((12, d, d), dtype=np.complex128)
    for k, q in enumerate(range(0, 5)):
        Hs[k, idx, idx ^ bit(q)] = 1.0                                   # X_q
    for k, q in enumerate(range(5, 10)):
        Hs[5 + k, idx, idx ^ bit(q)] = -1j * sgn(q)                      # Y_q: Y[0,1] = -i, Y[1,0] = +i
    Hs[10, idx, idx] = sgn(10)                                           # Z_10
    Hs[11, idx, idx ^ bit(11)] = sgn(3)                                  # X_11 Z_3
    M = H0.copy()
    u = np.random.default_rng(11).uniform(-1, 1, (S, 12))
    psi0 = problems.random_states(d, B, seed=5)
    return dict(H0=H0, Hs=Hs, M=M, u=u, psi0=psi0, dt=dt, B=B)


def herm4096_inputs():
    """Dense COMPLEX Hermitian drift, controls and observable (spectral norms ~ 1), theta ~ 1: 1 step, 16 states."""
    d, K, B, dt = 4096, 2, 16, 0.02
    rng = np.random.default_rng(2024)

    def herm():
        a = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        return (a + a.conj().T) / (4.0 * np.sqrt(d))                    # spectral norm ~ 1 (semicircle edge), no SVD needed

    H0, Hs, M = herm(), np.stack([herm() for _ in range(K)]), herm()
    u = rng.uniform(-1, 1, (1, K))
    psi0 = problems.random_states(d, B, seed=3)
    return dict(H0=H0, Hs=Hs, M=M, u=u, psi0=psi0, dt=dt, B=B)


CASES = {"ising12": ising12_inputs, "xyz12": xyz12_inputs, "herm4096": herm4096_inputs}


def signature(P):
    """What identifies the inputs: the pulse table, dt and weighted checksums of the states and the operators."""
    def cs(a):
        a = np.asarray(a, dtype=np.complex128).ravel()
        w = np.cos(np.arange(a.size, dtype=np.float64) * 0.7310585786300049)      # fixed weights: order-sensitive
        return np.array([np.dot(w, a.real), np.dot(w, a.imag)])
    return np.concatenate([np.asarray(P["u"], dtype=np.float64).ravel(), [P["dt"]], cs(P["psi0"]), cs(P["H0"]), cs(P["Hs"]), cs(P["M"])])


def compute(P, oracle):
    loss, grad, fin = oracle.energy_grad_one_frechet(P["H0"], list(P["Hs"]), P["M"], P["u"], P["dt"], P["psi0"])
    en = np.asarray(oracle.energies(P["M"], fin), dtype=np.float64)          # <psi_b|M|psi_b> of EVERY column
    return dict(loss=float(loss), grad=np.asarray(grad), fin=np.asarray(fin)[:, :FIN_COLS].copy(), energies=en)


def answer(case, oracle, P=None):
    """(inputs, oracle answer) of a case; the answer from the fixture when it was made for identical inputs."""
    P = P if P is not None else CASES[case]()
    if os.path.exists(GOLDEN):
        z = np.load(GOLDEN)
        if case + "/signature" in z.files:
            sig = signature(P)
            old = z[case + "/signature"]
            if old.shape == sig.shape and np.allclose(old, sig, rtol=1e-13, atol=1e-13):
                return P, dict(loss=float(z[case + "/loss"]), grad=z[case + "/grad"], fin=z[case + "/fin"],
                               energies=z[case + "/energies"], stored=True)
    a = compute(P, oracle)
    a["stored"] = False
    return P, a
