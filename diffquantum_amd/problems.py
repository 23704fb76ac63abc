"""Seeded synthetic inputs for the propagator path and the MaxCut
problem of the reference demo (/root/reference/demo_maxcut.py:10-85), as host numpy arrays
(small n) or built directly in HBM with torch (bench sizes).

Host-side setup code only: nothing here is on the timed path.
"""
from __future__ import annotations

import math

import numpy as np

_I2 = np.eye(2)
_X = np.array([[0.0, 1.0], [1.0, 0.0]])
_Z = np.array([[1.0, 0.0], [0.0, -1.0]])


def kron_chain(ops):
    out = np.array([[1.0]])
    for o in ops:
        out = np.kron(out, o)
    return out


def z_signs(n: int):
    """signs[q][i] = <i| Z_q |i> with qubit 0 the most significant bit (kron order of the demo)."""
    d = 1 << n
    idx = np.arange(d)
    return np.stack([1.0 - 2.0 * ((idx >> (n - 1 - q)) & 1) for q in range(n)])


def ising_diagonal(n: int, seed: int = 0):
    """diag of H0 = sum_{i<j} J_ij Z_i Z_j + sum_i h_i Z_i,  J, h ~ U(-1, 1)."""
    rng = np.random.default_rng(seed)
    J = rng.uniform(-1, 1, (n, n))
    h = rng.uniform(-1, 1, n)
    s = z_signs(n)
    diag = h @ s
    for i in range(n):
        for j in range(i + 1, n):
            diag = diag + J[i, j] * s[i] * s[j]
    return diag


def random_ising(n: int, seed: int = 0):
    """(H0, Hs[K=n], M) dense complex128 on the host: H0 random Ising (diagonal), H_k = X_k, M = H0."""
    d = 1 << n
    H0 = np.diag(ising_diagonal(n, seed)).astype(np.complex128)
    Hs = np.zeros((n, d, d), dtype=np.complex128)
    idx = np.arange(d)
    for q in range(n):
        Hs[q, idx, idx ^ (1 << (n - 1 - q))] = 1.0
    return H0, Hs, H0.copy()


def random_ising_torch(n: int, seed: int, device):
    """Same matrices as random_ising, built in device memory (complex128 torch tensors)."""
    import torch
    d = 1 << n
    diag = torch.from_numpy(ising_diagonal(n, seed)).to(device)
    H0 = torch.zeros((d, d), dtype=torch.complex128, device=device)
    H0.diagonal().copy_(diag.to(torch.complex128))
    Hs = torch.zeros((n, d, d), dtype=torch.complex128, device=device)
    idx = torch.arange(d, device=device)
    for q in range(n):
        Hs[q, idx, idx ^ (1 << (n - 1 - q))] = 1.0
    return H0, Hs, H0.clone()


def xy_drives_torch(n: int, seed: int, device, n_x: int = 6, n_y: int = 6):
    """Random Ising drift (as random_ising) with X_k controls on the first n_x qubits and Y_k controls on the last n_y ones: a
    COMPLEX Hermitian, sparse generator (I/Q drives) built in device memory.  K = n_x + n_y controls; M = H0."""
    import torch
    d = 1 << n
    diag = torch.from_numpy(ising_diagonal(n, seed)).to(device)
    H0 = torch.zeros((d, d), dtype=torch.complex128, device=device)
    H0.diagonal().copy_(diag.to(torch.complex128))
    Hs = torch.zeros((n_x + n_y, d, d), dtype=torch.complex128, device=device)
    idx = torch.arange(d, device=device)
    for q in range(n_x):
        Hs[q, idx, idx ^ (1 << (n - 1 - q))] = 1.0
    for j in range(n_y):
        e = 1 << (n - 1 - (n - n_y + j))
        bit = ((idx & e) != 0).to(torch.float64)
        Hs[n_x + j, idx, idx ^ e] = torch.complex(torch.zeros_like(bit), 2.0 * bit - 1.0)   # Y[0,1] = -i, Y[1,0] = +i
    return H0, Hs, H0.clone()


def qaoa_ring_torch(n: int, device):
    """The n-qubit ring analog-QAOA problem of qaoa_ring / the reference demo (demo_maxcut.py:44-79: H0 = sum_e (I - Z_i Z_j) by the
    demo's aliasing, controls = the n edge strings Z_i Z_j and the n vertex strings X_i), built in device memory: K = 2 n controls,
    M = H_cost = -H0 / 2.  (The dense host arrays of qaoa_ring are 6.7 GB at n = 12.)"""
    import torch
    d = 1 << n
    idx = torch.arange(d, device=device)
    zsign = lambda q: 1.0 - 2.0 * ((idx >> (n - 1 - q)) & 1).to(torch.float64)
    Hs = torch.zeros((2 * n, d, d), dtype=torch.complex128, device=device)
    diag0 = torch.zeros(d, dtype=torch.float64, device=device)
    for e in range(n):
        zz = zsign(e) * zsign((e + 1) % n)
        Hs[e, idx, idx] = zz.to(torch.complex128)
        diag0 += 1.0 - zz
    for q in range(n):
        Hs[n + q, idx, idx ^ (1 << (n - 1 - q))] = 1.0
    H0 = torch.zeros((d, d), dtype=torch.complex128, device=device)
    H0.diagonal().copy_(diag0.to(torch.complex128))
    M = torch.zeros((d, d), dtype=torch.complex128, device=device)
    M.diagonal().copy_((-0.5 * diag0).to(torch.complex128))
    return H0, Hs, M


def random_dense_torch(n: int, K: int, seed: int, device, complex_h: bool = False, drift_norm1: float = 12.0,
                       control_norm1: float = 1.0):
    """(H0, Hs[K], M) DENSE random matrices built in device memory: real symmetric (stored complex128, imaginary parts
    exactly zero) or complex Hermitian -- what `set_H` (diffqc.cc:43-67) accepts in general.  Every matrix is scaled to
    a given 1-norm (drift 12, controls 1: the scale of the Ising + X_k workload, so the step exponential needs the same
    polynomial degree) -- the structure of the data differs, not the size of the step."""
    import torch
    d = 1 << n
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed) + 777)

    def one(norm1):
        a = torch.randn((d, d), dtype=torch.float64, device=device, generator=gen)
        if complex_h:
            a = torch.complex(a, torch.randn((d, d), dtype=torch.float64, device=device, generator=gen))
            a = a + a.conj().T
        else:
            a = torch.complex(a + a.T, torch.zeros((d, d), dtype=torch.float64, device=device))
        return a * (norm1 / float(a.abs().sum(dim=0).max()))

    H0 = one(drift_norm1)
    Hs = torch.stack([one(control_norm1) for _ in range(K)]) if K else torch.zeros((0, d, d), dtype=torch.complex128, device=device)
    return H0, Hs, one(drift_norm1)


def random_hermitian_problem(n: int, K: int, seed: int = 0):
    """Dense random complex Hermitian H0, H_k, M (unit-ish norm) for parity tests at small n."""
    rng = np.random.default_rng(seed)
    d = 1 << n

    def herm():
        a = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        a = a + a.conj().T
        return a / np.linalg.norm(a, 2)

    return herm(), np.stack([herm() for _ in range(K)]) if K else np.zeros((0, d, d), complex), herm()


def random_real_symmetric_problem(n: int, K: int, seed: int = 0):
    """Dense random REAL symmetric H0, H_k (stored complex, imaginary parts exactly zero) and a complex Hermitian M:
    the class of Hamiltonians the engine's real-symmetric fast path (cos/sin form of the step exponential) covers --
    the reference's own problems are of this kind (Z/ZZ drifts, X controls: demo_maxcut.py:36-79)."""
    rng = np.random.default_rng(seed)
    d = 1 << n

    def sym():
        a = rng.standard_normal((d, d))
        a = a + a.T
        return (a / np.linalg.norm(a, 2)).astype(np.complex128)

    _, _, M = random_hermitian_problem(n, 0, seed + 1000)
    return sym(), np.stack([sym() for _ in range(K)]) if K else np.zeros((0, d, d), complex), M


def bump(b: int, n_basis: int, x):
    """The reference's "BSpline" bump (sim_plain.py:52-70 / diffqc.cc:82-93), vectorised over x."""
    tau = 1.0 / (n_basis - 2)
    tau_b = tau * (b - 1.5)
    l, r = tau_b - 1.5 * tau, tau_b + 1.5 * tau
    x = np.asarray(x, dtype=np.float64)
    v = (x - l) * (x - r) / (-((1.5 * tau) ** 2))
    return np.where((x < r) & (x > l), v, 0.0)


def pulse_table(K: int, n_steps: int, T: float, seed: int = 0, n_basis: int = 6, omega: float = math.pi,
                sigma: float = 0.5):
    """u[t][k] = omega * tanh(a_k(t)/2), a_k = sum_j c_kj bump_j(t/T), c ~ N(0, sigma); left endpoints."""
    rng = np.random.default_rng(seed + 12345)
    coeff = rng.normal(0.0, sigma, (K, n_basis))
    ts = np.arange(n_steps) * (T / n_steps)
    phi = np.stack([bump(j, n_basis, ts / T) for j in range(n_basis)])  # [n_basis][S]
    a = coeff @ phi  # [K][S]
    return np.ascontiguousarray((omega * np.tanh(a / 2.0)).T)


def random_states(d: int, B: int, seed: int = 0, first: int = 0):
    """Columns first..first+B-1 of the seeded state batch: column b ~ CN(0,1)^d normalised, rng(seed, b)."""
    out = np.empty((d, B), dtype=np.complex128)
    for j in range(B):
        rng = np.random.default_rng([seed, first + j])
        v = rng.standard_normal(d) + 1j * rng.standard_normal(d)
        out[:, j] = v / np.linalg.norm(v)
    return out


def maxcut_problem(graph=((0, 1), (0, 3), (1, 2), (2, 3)), n_qubit: int = 4, omega0=math.pi, omega1=math.pi,
                   n_layers: int = 1):
    """The 4-vertex MaxCut analog-QAOA instance of the reference demo.

    Returns dict(H_cost, H0, Hs, omegas, T, psi0) as numpy arrays.  Reproduces the demo's in-place
    aliasing (demo_maxcut.py:34-38,60-61): the accumulator shared by H0 and H_cost leaves
    H0 = sum_e (I - Z_i Z_j) while H_cost = -0.5 * that.
    """
    d = 1 << n_qubit
    II = np.eye(d)
    acc = np.zeros((d, d))
    for e in graph:
        acc = acc + (II - kron_chain([_Z if q in e else _I2 for q in range(n_qubit)]))
    H0 = acc.copy()
    H_cost = -0.5 * acc
    Hs, omegas = [], []
    for e in graph:
        Hs.append(kron_chain([_Z if q in e else _I2 for q in range(n_qubit)]))
        omegas.append(omega0)
    for i in range(n_qubit):
        Hs.append(kron_chain([_X if q == i else _I2 for q in range(n_qubit)]))
        omegas.append(omega1)
    T = math.pi * (1.0 / omega0 + 1.0 / omega1) * n_layers
    psi0 = np.ones(d) / math.sqrt(float(d))
    return dict(H_cost=H_cost, H0=H0, Hs=np.stack(Hs), omegas=np.array(omegas), T=T, psi0=psi0)


def qaoa_ring(n: int, omega: float = math.pi):
    """n-qubit ring analog QAOA (BASELINE config 2): H0 = sum_e (I - Z_i Z_j), controls = edge ZZ + vertex X."""
    graph = tuple((i, (i + 1) % n) for i in range(n))
    return maxcut_problem(graph, n, omega, omega)
