"""fp32-complex (complex64) engine: the reference's commented-out `typedef float Scalar` (diffqc.cc:13-19).
north_star tolerance for fp32: 1e-4 on final amplitudes and gradients."""
import numpy as np
import pytest
import scipy.linalg

from diffquantum_amd import problems

pytestmark = pytest.mark.gpu
TOL32 = 1e-4


def _rand(rng, *shape):
    return (rng.standard_normal(shape) + 1j * rng.standard_normal(shape)).astype(np.complex64)


@pytest.mark.parametrize("tile", [0, 32, 64, 128])
@pytest.mark.parametrize("opA,opB", [(0, 0), (0, 1), (1, 0), (1, 1)])
@pytest.mark.parametrize("M,N,K", [(16, 1, 16), (33, 17, 9), (64, 64, 64), (130, 257, 70), (256, 128, 256), (1, 1, 1)])
def test_cgemm_matches_numpy(opA, opB, M, N, K, tile):
    from diffquantum_amd.engine import Engine
    rng = np.random.default_rng(M * 1000 + N * 10 + K + opA * 2 + opB)
    A = _rand(rng, *((K, M) if opA else (M, K)))
    B = _rand(rng, *((N, K) if opB else (K, N)))
    C0 = _rand(rng, M, N)
    with Engine(16, 0, dtype="c64") as e:
        e.set_option("force_tile", tile)
        got = e.zgemm(A, B, opA=opA, opB=opB, alpha=0.75, beta=-0.5, C_in=C0)
    assert got.dtype == np.complex64
    oa = (A.conj().T if opA else A).astype(np.complex128)
    ob = (B.conj().T if opB else B).astype(np.complex128)
    ref = 0.75 * (oa @ ob) - 0.5 * C0.astype(np.complex128)
    bound = 6e-8 * (K + 4) * np.max(np.abs(A)) * np.max(np.abs(B)) * 4 + 1e-6
    assert np.max(np.abs(got - ref)) < bound


def test_cgemm_asymmetric_identity():
    from diffquantum_amd.engine import Engine
    B = (np.arange(64 * 48, dtype=np.float64).reshape(64, 48) * (1 + 0.5j)).astype(np.complex64)
    with Engine(4, 0, dtype="c64") as e:
        got = e.zgemm(np.eye(64), B)
    assert np.array_equal(got, B)


@pytest.mark.parametrize("d,norm", [(4, 1e-3), (16, 0.1), (64, 1.0), (32, 12.0)])
def test_expm_c64(d, norm):
    from diffquantum_amd.engine import Engine
    rng = np.random.default_rng(d)
    H = rng.standard_normal((2, d, d)) + 1j * rng.standard_normal((2, d, d))
    A = -1j * (H + np.conj(np.transpose(H, (0, 2, 1))))
    A *= norm / np.max([np.linalg.norm(a, 1) for a in A])
    A32 = A.astype(np.complex64)
    with Engine(d, 0, dtype="c64") as e:
        U = e.expm(A32)
    for z in range(2):
        ref = scipy.linalg.expm(A32[z].astype(np.complex128))
        assert np.max(np.abs(U[z] - ref)) < 2e-6 * max(1.0, norm) * np.sqrt(d)


@pytest.mark.parametrize("n,K,B,S,dt", [(2, 2, 3, 8, 0.1), (4, 3, 40, 30, 0.05), (5, 2, 130, 12, -0.07)])
def test_chain_and_gradient_c64_vs_oracle(oracle, n, K, B, S, dt):
    from diffquantum_amd.engine import Engine
    H0, Hs, M = problems.random_hermitian_problem(n, K, seed=n + K)
    u = np.random.default_rng(S).uniform(-2, 2, (S, K))
    psi0 = problems.random_states(1 << n, B, seed=4)
    with Engine(1 << n, K, dtype="c64") as e:
        e.set_hamiltonians(H0, Hs)
        e.set_observable(M)
        fwd = e.propagate(u, dt, psi0)
        loss, grad, fin, en = e.energy_grad(u, dt, psi0)
        assert fwd.dtype == np.complex64 and fin.dtype == np.complex64 and grad.dtype == np.float64
        assert e.info("dtype") == 1
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    assert np.max(np.abs(fwd - rf)) < TOL32 and np.max(np.abs(fin - rf)) < TOL32
    assert abs(loss - rl) < TOL32 * B
    assert np.max(np.abs(grad - rg)) < TOL32 * B
    assert np.max(np.abs(en - oracle.energies(M, rf))) < TOL32


def test_c64_demo_problem_forward(oracle, golden_dir):
    """G3 (reference golden, fp64) reproduced by the fp32 engine within the fp32 tolerance."""
    import os
    from diffquantum_amd.engine import Engine
    G = dict(np.load(os.path.join(golden_dir, "reference_goldens.npz")))
    with Engine(16, 8, dtype="c64") as e:
        e.set_hamiltonians(G["demo_H0"], G["demo_Hs"])
        out = e.propagate(G["g3_u"], float(G["demo_T"]) / 30, G["demo_psi0"])
    assert np.max(np.abs(out[:, 0] - G["g3_final"])) < TOL32


def test_c64_rejects_wrong_torch_dtype():
    import torch
    from diffquantum_amd.engine import Engine
    with Engine(4, 0, dtype="c64") as e:
        with pytest.raises(ValueError):
            e.set_hamiltonians(torch.zeros((4, 4), dtype=torch.complex128, device="cuda:0"), None)
    with pytest.raises(ValueError):
        Engine(4, 0, dtype="c32")
