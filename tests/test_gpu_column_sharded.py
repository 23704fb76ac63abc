"""GPU parity of the column-sharded form of the wide-batch energy gradient (diffquantum_amd.distributed.column_sharded_energy_grad,
C-ABI dq_gram / dq_energy_grad_columns; BASELINE configs[3]: the 32768-state batch over 8 GPUs): virtual ranks on one GPU -- the
same rank program the multi-process run executes, collectives carried out on the spot -- against dq_energy_grad of the whole batch
on the same device and against the oracle (diffqc.cc:190-200 loop, loss sim_plain.py:281).  The gloo world-2 run of the program is
in tests/test_distributed_cpu.py."""
import numpy as np
import pytest

from diffquantum_amd import problems
from diffquantum_amd.distributed import shard_range, virtual_column_sharded_energy_grad

pytestmark = pytest.mark.gpu
TOL = 1e-10


def _engine(H0, Hs, M):
    from diffquantum_amd.engine import Engine
    e = Engine(len(H0), len(Hs))
    e.set_hamiltonians(H0, Hs)
    e.set_observable(M)
    return e


def _qaoa(n):
    p = problems.qaoa_ring(n)
    return p["H0"].astype(np.complex128), p["Hs"].astype(np.complex128), p["H_cost"].astype(np.complex128)


@pytest.mark.parametrize("case,n,B,S,world", [("ising", 8, 300, 6, 3), ("qaoa", 8, 270, 5, 2), ("ising", 9, 1300, 25, 5)])
def test_virtual_ranks_match_the_single_gpu_gradient_and_the_oracle(oracle, case, n, B, S, world):
    H0, Hs, M = _qaoa(n) if case == "qaoa" else problems.random_ising(n, 3)
    K, d = len(Hs), 1 << n
    dt = 2e-3
    u = np.random.default_rng(n + S).uniform(-1, 1, (S, K))
    full = problems.random_states(d, B, seed=12)
    engines = [_engine(H0, Hs, M) for _ in range(world)]
    try:
        shards = [np.ascontiguousarray(full[:, f:f + c]) for f, c in (shard_range(B, r, world) for r in range(world))]
        res = virtual_column_sharded_energy_grad(engines, u, dt, shards, slice_product=False)   # P built by every rank
        assert all(e.info("state_columns_wide_adjoint") == 1 and e.info("state_columns_compose_segments") >= 1 for e in engines)
        res += virtual_column_sharded_energy_grad(engines, u, dt, shards, slice_product=True)    # P in time slices, one per rank
        assert all(e.info("state_columns_wide_adjoint") == 1 for e in engines)
        engines[0].set_option("state_columns", 1)              # the whole batch on one GPU, swept state by state
        engines[0].set_option("state_columns_wide_adjoint", 0)
        l0, g0, _, _ = engines[0].energy_grad(u, dt, full, want_final=False, want_energies=False)
        assert engines[0].info("state_columns") == 1 and engines[0].info("state_columns_wide_adjoint") == 0
    finally:
        for e in engines:
            e.close()
    scale = max(1.0, float(np.max(np.abs(g0))))
    for l, g, _, _ in res:
        assert abs(l - l0) < TOL * B and np.max(np.abs(g - g0)) < TOL * scale, (abs(l - l0), float(np.max(np.abs(g - g0))))
    if n == 8:
        rl, rg, _ = oracle.energy_grad(H0, list(Hs), M, u, dt, full)
        assert abs(res[0][0] - rl) < TOL * B and np.max(np.abs(res[0][1] - rg)) < TOL * max(1.0, float(np.max(np.abs(rg))))
    assert np.max(np.abs(g0)) > 1e-4


def test_more_ranks_than_columns_and_bad_ranges():
    from diffquantum_amd._lib import DiffqcError
    n, B, S = 8, 260, 3
    H0, Hs, M = problems.random_ising(n, 5)
    u = np.random.default_rng(2).uniform(-1, 1, (S, n))
    full = problems.random_states(1 << n, B, seed=3)
    with _engine(H0, Hs, M) as e:
        R = e.gram(full)
        assert np.max(np.abs(R - full @ full.conj().T)) < 1e-11
        l_all, g_all = e.energy_grad_columns(u, 2e-3, R, 0, 256)
        l_a, g_a = e.energy_grad_columns(u, 2e-3, R, 0, 100)
        l_b, g_b = e.energy_grad_columns(u, 2e-3, R, 100, 156)
        l_0, g_0 = e.energy_grad_columns(u, 2e-3, R, 256, 0)
        assert l_0 == 0.0 and not np.any(g_0)
        assert abs(l_a + l_b - l_all) < 1e-10 * B and np.max(np.abs(g_a + g_b - g_all)) < 1e-10
        with pytest.raises((DiffqcError, ValueError)):
            e.energy_grad_columns(u, 2e-3, R, 200, 100)


def test_config4_shape_four_virtual_ranks_on_one_gpu():
    """d = 4096, 8192 states, 3 steps, 4 virtual ranks on ONE context (the ranks' calls are sequential): the sums equal the single-GPU
    wide-batch gradient of the whole batch; a rank's sweep covers 1024 of the 4096 columns."""
    import torch
    from diffquantum_amd.engine import Engine
    n, B, S, dt, world = 12, 8192, 3, 1e-3, 4
    d = 1 << n
    dev = torch.device("cuda:0")
    H0, Hs, M = problems.random_ising_torch(n, 0, dev)
    u = problems.pulse_table(n, 2000, 2000 * dt, seed=0)[:S].copy()
    u[1] += 0.3 * np.random.default_rng(8).uniform(-1, 1, n)
    gen = torch.Generator(device=dev)
    gen.manual_seed(77)
    psi0 = torch.randn((d, B), dtype=torch.float64, device=dev, generator=gen) + 1j * torch.randn((d, B), dtype=torch.float64, device=dev, generator=gen)
    psi0 = (psi0 / torch.linalg.vector_norm(psi0, dim=0, keepdim=True)).contiguous()
    with Engine(d, n) as e:
        e.use_torch_stream()
        e.set_hamiltonians(H0, Hs)
        e.set_observable(M)
        shards = [psi0[:, f:f + c].contiguous() for f, c in (shard_range(B, r, world) for r in range(world))]
        res = virtual_column_sharded_energy_grad([e] * world, u, dt, shards)
        e.set_option("state_columns_wide_adjoint", 1)
        l1, g1, _, _ = e.energy_grad(u, dt, psi0, want_final=False, want_energies=False)
        assert e.info("state_columns_wide_adjoint") == 1
    scale = max(1.0, float(np.max(np.abs(g1))))
    for l, g, _, _ in res:
        assert abs(l - l1) < TOL * B and np.max(np.abs(g - g1)) < TOL * scale
