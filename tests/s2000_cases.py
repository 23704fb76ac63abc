"""Inputs of the chain-length parity checks at the step count BASELINE states (S = 2000; configs[2]: "2000 steps"), and their
oracle answers.

The state-column form (diffquantum_amd/csrc/statecol.hip) re-derives psi_t = exp(-A_t) psi_{t+1} all the way down the chain
instead of keeping checkpoints; this fixture pins it over the FULL 2000 steps of the reference loop (diffqc.cc:190-200,
sim_plain.py:135-150) at the dimensions the oracle can afford: d = 256 and d = 512, 3 states, BASELINE's generator family (random
Ising drift + X_k, dt = 1e-3, the bump-basis pulse of problems.pulse_table) and one X / Y / Z case.  An oracle evaluation
(scipy expm + expm_frechet per step) is ~15 minutes of host time per case, so the answers are a committed fixture,
tests/golden/oracle_s2000.npz, written by tests/golden/make_oracle_s2000_goldens.py (--check recomputes and compares).
`answer(case)` returns the stored answer only when it was made for identical inputs (signature check), else fails loudly."""
import os

import numpy as np

from diffquantum_amd import problems

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_s2000.npz")
S, B, DT = 2000, 3, 1e-3


def _ising(n, seed):
    H0, Hs, M = problems.random_ising(n, seed=seed)
    u = problems.pulse_table(n, S, S * DT, seed=seed + 1)
    psi0 = problems.random_states(1 << n, B, seed=seed + 2)
    return dict(H0=H0, Hs=Hs, M=M, u=u, psi0=psi0, dt=DT, B=B)


def ising8_inputs():
    """d = 256 (one wave per column, or the wide 256-thread shape): random Ising drift, X_k controls, M = H0."""
    return _ising(8, 20)


def ising9_inputs():
    """d = 512 (the smallest dimension the form is auto-selected at; 8 rows per lane / 256 x 2)."""
    return _ising(9, 30)


def xyz9_inputs():
    """d = 512, COMPLEX Hermitian Pauli sum: X_q on four qubits, Y_q on three others, a Z_q control (diagonal control: per-step
    diagonal weights) and a two-body X_a Z_b control (row signs); M = a different Ising diagonal."""
    n = 9
    d = 1 << n
    H0 = np.diag(problems.ising_diagonal(n, 41)).astype(np.complex128)
    idx = np.arange(d)
    bit = lambda q: 1 << (n - 1 - q)
    sgn = lambda q: 1.0 - 2.0 * ((idx & bit(q)) != 0)
    Hs = np.zeros((9, d, d), dtype=np.complex128)
    for k, q in enumerate(range(0, 4)):
        Hs[k, idx, idx ^ bit(q)] = 1.0                                   # X_q
    for k, q in enumerate(range(4, 7)):
        Hs[4 + k, idx, idx ^ bit(q)] = -1j * sgn(q)                      # Y_q
    Hs[7, idx, idx] = sgn(7)                                             # Z_7
    Hs[8, idx, idx ^ bit(8)] = sgn(2)                                    # X_8 Z_2
    M = np.diag(problems.ising_diagonal(n, 43)).astype(np.complex128)
    u = problems.pulse_table(9, S, S * DT, seed=44)
    psi0 = problems.random_states(d, B, seed=45)
    return dict(H0=H0, Hs=Hs, M=M, u=u, psi0=psi0, dt=DT, B=B)


CASES = {"ising8": ising8_inputs, "ising9": ising9_inputs, "xyz9": xyz9_inputs}


def signature(P):
    """What identifies the inputs: checksums of the pulse table, the states and the operators, and dt."""
    def cs(a):
        a = np.asarray(a, dtype=np.complex128).ravel()
        w = np.cos(np.arange(a.size, dtype=np.float64) * 0.7310585786300049)      # fixed weights: order-sensitive
        return np.array([np.dot(w, a.real), np.dot(w, a.imag)])
    return np.concatenate([[P["dt"], float(len(P["u"]))], cs(P["u"]), cs(P["psi0"]), cs(P["H0"]), cs(P["Hs"]), cs(P["M"])])


def compute(P, oracle):
    loss, grad, fin = oracle.energy_grad_one_frechet(P["H0"], list(P["Hs"]), P["M"], P["u"], P["dt"], P["psi0"])
    en = np.asarray(oracle.energies(P["M"], fin), dtype=np.float64)
    return dict(loss=float(loss), grad=np.asarray(grad), fin=np.asarray(fin), energies=en)


def answer(case):
    """(inputs, stored oracle answer) of a case; raises when the fixture is missing or was made for other inputs."""
    P = CASES[case]()
    z = np.load(GOLDEN)
    sig, old = signature(P), z[case + "/signature"]
    if old.shape != sig.shape or not np.allclose(old, sig, rtol=1e-13, atol=1e-13):
        raise AssertionError("tests/golden/oracle_s2000.npz was made for other inputs of case {}: regenerate it with "
                             "tests/golden/make_oracle_s2000_goldens.py".format(case))
    return P, dict(loss=float(z[case + "/loss"]), grad=z[case + "/grad"], fin=z[case + "/fin"], energies=z[case + "/energies"])
