"""Inputs of the d = 8192 (13 qubits, complex128) oracle comparison of the matrix-free routing (tests/test_gpu_matrix_free_routing.py)
and its oracle answer.  One oracle step at this size (scipy expm + expm_frechet on 1 GB matrices) is ~15 minutes of host time and
~20 GB of host memory, so the answer is a committed fixture, tests/golden/oracle_d8192.npz, written by
tests/golden/make_oracle_d8192_golden.py (--check recomputes and compares).  The dense API accepts any dimension
(diffqc.cc:43-67); above 4096 in complex128 a Pauli-sum generator runs on the matrix-free passes inside the same context."""
import os

import numpy as np

from diffquantum_amd import problems

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_d8192.npz")
N, K, S, B, DT = 13, 4, 1, 8, 1e-3
QUBITS = (0, 5, 9, 12)          # X_q controls on these qubits (4 dense 1 GB matrices instead of 13 on the oracle's host)


def diag():
    return problems.ising_diagonal(N, 77)


def pulse():
    return np.array([[0.83, -0.41, 0.27, -0.66]])


def states():
    return problems.random_states(1 << N, B, seed=78)


def dense_inputs():
    """(H0, Hs, M) dense complex128 on the host (6.4 GB)."""
    d = 1 << N
    H0 = np.zeros((d, d), dtype=np.complex128)
    H0[np.arange(d), np.arange(d)] = diag()
    Hs = np.zeros((K, d, d), dtype=np.complex128)
    idx = np.arange(d)
    for k, q in enumerate(QUBITS):
        Hs[k, idx, idx ^ (1 << (N - 1 - q))] = 1.0
    return H0, Hs, H0.copy()


def signature():
    w = np.cos(np.arange(1 << N, dtype=np.float64) * 0.7310585786300049)
    p = states()
    return np.concatenate([[DT, float(N), float(K)], np.asarray(QUBITS, dtype=np.float64), pulse().ravel(), [np.dot(w, diag())],
                           [np.dot(w, p[:, 0].real), np.dot(w, p[:, B - 1].imag)]])


def answer():
    z = np.load(GOLDEN)
    sig = signature()
    if z["signature"].shape != sig.shape or not np.allclose(z["signature"], sig, rtol=1e-13, atol=1e-13):
        raise AssertionError("tests/golden/oracle_d8192.npz was made for other inputs: regenerate it with tests/golden/make_oracle_d8192_golden.py")
    return dict(loss=float(z["loss"]), grad=z["grad"], fin=z["fin"], energies=z["energies"])
