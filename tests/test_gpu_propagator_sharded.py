"""SURVEY.md section 8-F2 in its literal form on the GPU engine: dq_build_propagators / dq_propagator_vjp against the oracle
(scipy expm / expm_frechet on the reference's step generator, sim_plain.py:137-146), and the propagator-sharded state-form
rank program (propagators built 1/G per rank and all-gathered, cotangents reduce-scattered to their owners) with virtual ranks
on one GPU against the oracle's gradient at 1e-10."""
import numpy as np
import pytest
import scipy.linalg as sla

from diffquantum_amd import problems

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _engine(H0, Hs, M, **opts):
    from diffquantum_amd.engine import Engine
    e = Engine(len(H0), len(Hs))
    e.set_hamiltonians(H0, Hs)
    e.set_observable(M)
    for k, v in opts.items():
        e.set_option(k, v)
    return e


PROBLEMS = {
    "complex": lambda: problems.random_hermitian_problem(3, 2, seed=11),
    "real_dense": lambda: problems.random_real_symmetric_problem(3, 2, seed=12),
    "ising_sparse": lambda: problems.random_ising(3, seed=13),
}


@pytest.mark.parametrize("kind", list(PROBLEMS))
def test_build_propagators_and_vjp_match_oracle(oracle, kind):
    H0, Hs, M = PROBLEMS[kind]()
    K, d, n, dt = len(Hs), len(H0), 5, 0.07
    rng = np.random.default_rng(3)
    u = rng.uniform(-1, 1, (n, K))
    G = rng.standard_normal((n, d, d)) + 1j * rng.standard_normal((n, d, d))
    opts = {} if kind == "complex" else {"real_min_dim": 1}
    if kind == "ising_sparse":
        opts["sparse_generator"] = 1
    with _engine(H0, Hs, M, **opts) as e:
        U = e.build_propagators(u, dt)
        g = e.propagator_vjp(u, dt, G)
        assert e.build_propagators(u[:0], dt).shape == (0, d, d)
        with pytest.raises(ValueError):
            e.propagator_vjp(u, dt, G[:2])
    for t in range(n):
        A = oracle.step_generator(H0, list(Hs), u[t], dt)
        assert np.max(np.abs(U[t] - sla.expm(A))) < 1e-13
        for k in range(K):
            ref = np.real(np.vdot(G[t], sla.expm_frechet(A, -1j * dt * Hs[k], compute_expm=False)))
            assert abs(g[t, k] - ref) < 1e-12, (t, k)


@pytest.mark.parametrize("kind", list(PROBLEMS))
@pytest.mark.parametrize("world,per_rank", [(1, 2), (2, 1), (3, 2)])
def test_propagator_sharded_virtual_ranks_match_oracle(oracle, kind, world, per_rank):
    from diffquantum_amd.distributed import shard_range, virtual_propagator_sharded_energy_grad
    H0, Hs, M = PROBLEMS[kind]()
    K, d, S, B, dt = len(Hs), len(H0), 7, 7, 0.05
    u = np.random.default_rng(world).uniform(-1, 1, (S, K))
    full = problems.random_states(d, B, seed=5)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, full)
    opts = {} if kind == "complex" else {"real_min_dim": 1}
    engines = [_engine(H0, Hs, M, **opts) for _ in range(world)]
    try:
        shards = [np.ascontiguousarray(full[:, f:f + c]) for f, c in (shard_range(B, r, world) for r in range(world))]
        res = virtual_propagator_sharded_energy_grad(engines, M, u, dt, shards, per_rank=per_rank)
    finally:
        for e in engines:
            e.close()
    for r, (l, g, f, en) in enumerate(res):
        f0, c0 = shard_range(B, r, world)
        assert abs(l - rl) < TOL * B and np.max(np.abs(g - rg)) < TOL
        assert np.max(np.abs(f - rf[:, f0:f0 + c0])) < TOL
        assert np.max(np.abs(en - oracle.energies(M, rf[:, f0:f0 + c0]))) < TOL


def test_propagator_sharded_device_tensors_mid_dimension(oracle):
    """d = 256 with CUDA tensors end to end (propagators, states and cotangents stay on the device), 2 virtual ranks, against
    the single-call engine gradient and the oracle."""
    import torch
    from diffquantum_amd.distributed import shard_range, virtual_propagator_sharded_energy_grad
    from diffquantum_amd.engine import Engine
    dev = torch.device("cuda:0")
    p = problems.qaoa_ring(8)
    H0, Hs, M = p["H0"].astype(complex), p["Hs"].astype(complex), p["H_cost"].astype(complex)
    S, B, dt = 6, 24, 0.01
    u = problems.pulse_table(16, S, S * dt, seed=2)
    full = problems.random_states(256, B, seed=4)
    rl, rg, rf = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, dt, full)
    world = 2
    engines = []
    for _ in range(world):
        e = Engine(256, 16)
        e.use_torch_stream()
        e.set_hamiltonians(H0, Hs)
        e.set_observable(M)
        engines.append(e)
    try:
        shards = [torch.from_numpy(np.ascontiguousarray(full[:, f:f + c])).to(dev) for f, c in
                  (shard_range(B, r, world) for r in range(world))]
        res = virtual_propagator_sharded_energy_grad(engines, torch.from_numpy(M).to(dev), u, dt, shards, per_rank=2)
    finally:
        for e in engines:
            e.close()
    for r, (l, g, f, en) in enumerate(res):
        f0, c0 = shard_range(B, r, world)
        assert abs(l - rl) < TOL * B and np.max(np.abs(g - rg)) < TOL
        assert float((f.cpu() - torch.from_numpy(rf[:, f0:f0 + c0])).abs().max()) < TOL
