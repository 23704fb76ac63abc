"""GPU parity of the time-sliced operator form: the phases dq_slice_forward /
dq_slice_terminal / dq_slice_backward of dq_energy_grad, and the rank program of diffquantum_amd.distributed that
splits the chain of `psi = U_t psi` (diffqc.cc:190-200, sim_plain.py:135-150) over TIME across ranks.

Checked against the oracle (the reference's expm + matmul loop and its exact gradient) and against the engine's own
single-call dq_energy_grad on the whole batch, with G virtual ranks on the one GPU of the test box (one dq_ctx per rank,
collectives carried out in-process; the same rank program runs through gloo in tests/test_distributed_cpu.py and
through RCCL in bench.py).

Tolerance (BASELINE.json north_star): 1e-10 on amplitudes and gradients in fp64, 1e-4 in fp32.
"""
import numpy as np
import pytest

from diffquantum_amd import problems
from diffquantum_amd.distributed import (shard_range, time_sliced_energy_grad, time_sliced_propagate,
                                         virtual_time_sliced_energy_grad)

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


def _problem(kind, n, K, seed):
    if kind == "real":
        return problems.random_real_symmetric_problem(n, K, seed=seed)
    return problems.random_hermitian_problem(n, K, seed=seed)


@pytest.mark.parametrize("kind", ["complex", "real"])
@pytest.mark.parametrize("n,B,S", [(2, 1, 5), (3, 8, 9), (4, 24, 7), (4, 5, 6)])
def test_single_slice_equals_energy_grad_and_oracle(oracle, kind, n, B, S):
    """One slice covering the chain = dq_energy_grad (the world-size-1 path of the rank program)."""
    K, dt = 3, 0.05
    H0, Hs, M = _problem(kind, n, K, seed=n)
    u = np.random.default_rng(S).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(1 << n, B, seed=3)
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    with _engine(H0, Hs, M, real_min_dim=1) as e:
        loss, grad, fin, en = time_sliced_energy_grad(e, u, dt, psi0, want_final=True, want_energies=True)
        assert e.info("real_path") == (0 if kind == "complex" else e.info("real_path"))
        l2, g2, f2, e2 = e.energy_grad(u, dt, psi0)
    assert abs(loss - rl) < TOL and np.max(np.abs(grad - rg)) < TOL and np.max(np.abs(fin - rf)) < TOL
    assert abs(loss - l2) < TOL and np.max(np.abs(grad - g2)) < TOL and np.max(np.abs(fin - f2)) < TOL
    assert np.max(np.abs(en - oracle.energies(M, rf))) < TOL


@pytest.mark.parametrize("kind", ["complex", "real"])
@pytest.mark.parametrize("world", [2, 3, 5])
@pytest.mark.parametrize("skew", [1, 0])
def test_virtual_ranks_equal_oracle(oracle, kind, world, skew):
    """Chain split over time across `world` virtual ranks, batch split over the same ranks; ragged slices, and
    world > S leaves ranks with an empty slice (identity slice product, no backward steps)."""
    n, K, dt, B = 3, 2, 0.07, 11
    H0, Hs, M = _problem(kind, n, K, seed=11)
    full = problems.random_states(1 << n, B, seed=5)
    shards = [np.ascontiguousarray(full[:, f:f + c]) for f, c in (shard_range(B, r, world) for r in range(world))]
    for S in (4, 7):
        u = np.random.default_rng(S + world).uniform(-1, 1, (S, K))
        rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, full)
        engines = [_engine(H0, Hs, M, real_min_dim=1, skew_trick=skew) for _ in range(world)]
        try:
            res = virtual_time_sliced_energy_grad(engines, u, dt, shards)
        finally:
            for e in engines:
                e.close()
        for r, (l, g, f, en) in enumerate(res):
            f0, c0 = shard_range(B, r, world)
            assert abs(l - rl) < TOL, (r, l, rl)
            assert np.max(np.abs(g - rg)) < TOL, r
            assert np.max(np.abs(f - rf[:, f0:f0 + c0])) < TOL, r
            assert np.max(np.abs(en - oracle.energies(M, rf[:, f0:f0 + c0]))) < TOL, r


def test_virtual_ranks_device_tensors_mid_dimension():
    """d = 256 on the real path with torch device tensors end to end (as bench.py runs it): four virtual ranks against
    the single-call gradient of the whole batch; the forward call's propagator cache serves the backward call."""
    import torch
    n, K, dt, B, S, world = 8, 4, 2e-3, 256, 24, 4
    H0, Hs, M = problems.random_real_symmetric_problem(n, K, seed=2)
    full = problems.random_states(1 << n, B, seed=8)
    u = np.random.default_rng(0).uniform(-1, 1, (S, K))
    dev = torch.device("cuda", 0)
    with _engine(H0, Hs, M) as e:
        l0, g0, f0, _ = e.energy_grad(u, dt, full)
        assert e.info("real_path") >= 1 and e.info("operator_backward") == 1
    engines = [_engine(H0, Hs, M) for _ in range(world)]
    shards = [torch.from_numpy(np.ascontiguousarray(full[:, f:f + c])).to(dev)
              for f, c in (shard_range(B, r, world) for r in range(world))]
    ut = torch.from_numpy(u).to(dev)
    try:
        res = virtual_time_sliced_energy_grad(engines, ut, dt, shards)
        cached = [e.info("cached_steps") for e in engines]
    finally:
        for e in engines:
            e.close()
    assert all(c == S // world for c in cached), cached
    for r, (l, g, f, en) in enumerate(res):
        a, c = shard_range(B, r, world)
        assert abs(l - l0) < TOL * B
        assert np.max(np.abs(g - g0)) < TOL
        assert np.max(np.abs(f.cpu().numpy() - f0[:, a:a + c])) < TOL


def test_backward_without_matching_forward_rebuilds():
    """dq_slice_backward after a forward call on OTHER pulse rows (or none) must not read the stale cache."""
    n, K, dt, B, S = 7, 3, 2e-3, 128, 6
    H0, Hs, M = problems.random_real_symmetric_problem(n, K, seed=4)
    psi0 = problems.random_states(1 << n, B, seed=1)
    rng = np.random.default_rng(3)
    u, u_other = rng.uniform(-1, 1, (S, K)), rng.uniform(-1, 1, (S, K))
    with _engine(H0, Hs, M) as e:
        l0, g0, _, _ = e.energy_grad(u, dt, psi0)
        P = e.slice_forward(u, dt)
        loss, W, _, _ = e.slice_terminal(P, psi0)
        e.slice_forward(u_other, dt)          # leaves the cache holding another slice's intermediates
        g1 = e.slice_backward(u, dt, W)
        g2, Ws = e.slice_backward(u, dt, W, want_w_start=True)   # no forward at all in between
    assert abs(loss - l0) < TOL * B and np.max(np.abs(g1 - g0)) < TOL and np.max(np.abs(g2 - g0)) < TOL
    # W at the start of the chain = Lambda_0 Psi_0^H: anti-Hermitian part carried (skew form)
    assert np.max(np.abs(Ws + Ws.conj().T)) < 1e-12 * B


def test_fidelity_objective_sliced(oracle):
    n, K, dt, B, S, world = 3, 2, 0.05, 6, 8, 3
    H0, Hs, M = problems.random_hermitian_problem(n, K, seed=6)
    full = problems.random_states(1 << n, B, seed=2)
    tg = problems.random_states(1 << n, B, seed=12)
    u = np.random.default_rng(5).uniform(-1, 1, (S, K))
    rl, rg, rf, rfid = oracle.fidelity_grad(H0, list(Hs), u, dt, full, tg)
    spans = [shard_range(B, r, world) for r in range(world)]
    engines = [_engine(H0, Hs, M) for _ in range(world)]
    try:
        res = virtual_time_sliced_energy_grad(engines, u, dt, [np.ascontiguousarray(full[:, f:f + c]) for f, c in spans],
                                              targets=[np.ascontiguousarray(tg[:, f:f + c]) for f, c in spans])
    finally:
        for e in engines:
            e.close()
    for r, (l, g, f, fid) in enumerate(res):
        f0, c0 = spans[r]
        assert abs(l - rl) < TOL and np.max(np.abs(g - rg)) < TOL
        assert np.max(np.abs(fid - rfid[f0:f0 + c0])) < TOL


def test_fp32_sliced(oracle):
    n, K, dt, B, S, world = 4, 3, 0.05, 16, 10, 2
    H0, Hs, M = problems.random_real_symmetric_problem(n, K, seed=9)
    full = problems.random_states(1 << n, B, seed=2)
    u = np.random.default_rng(5).uniform(-1, 1, (S, K))
    rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, full)
    for rmin in (1, 1 << 20):  # real path / complex tape
        engines = [_engine(H0, Hs, M, dtype="c64", real_min_dim=rmin) for _ in range(world)]
        shards = [np.ascontiguousarray(full[:, f:f + c]).astype(np.complex64)
                  for f, c in (shard_range(B, r, world) for r in range(world))]
        try:
            res = virtual_time_sliced_energy_grad(engines, u, dt, shards)
        finally:
            for e in engines:
                e.close()
        for l, g, f, en in res:
            assert abs(l - rl) < 1e-4 * B and np.max(np.abs(g - rg)) < 1e-4


def test_slice_argument_errors():
    from diffquantum_amd.engine import Engine
    H0, Hs, M = problems.random_hermitian_problem(2, 2, seed=1)
    psi0 = problems.random_states(4, 3, seed=1)
    u = np.zeros((3, 2))
    with Engine(4, 2) as e:
        with pytest.raises(Exception):      # Hamiltonians not set
            e.slice_forward(u, 0.1)
        e.set_hamiltonians(H0, Hs)
        P = e.slice_forward(u, 0.1)
        with pytest.raises(Exception):      # observable not set
            e.slice_terminal(P, psi0)
        e.set_observable(M)
        with pytest.raises(ValueError):
            e.slice_forward(np.zeros((3, 5)), 0.1)
        with pytest.raises(ValueError):
            e.slice_forward(u, float("nan"))
        with pytest.raises(ValueError):
            e.slice_terminal(np.zeros((3, 3), dtype=complex), psi0)
        loss, W, _, _ = e.slice_terminal(P, psi0)
        with pytest.raises(ValueError):
            e.slice_backward(u, float("inf"), W)
        assert e.slice_backward(np.zeros((0, 2)), 0.1, W).shape == (0, 2)
    A = np.array([[0, 1], [0, 0]], dtype=complex)  # non-Hermitian generator: the operator form is refused
    with Engine(2, 0) as e:
        e.set_hamiltonians(A, np.zeros((0, 2, 2), dtype=complex))
        with pytest.raises(NotImplementedError):
            e.slice_forward(np.zeros((2, 0)), 0.1)


@pytest.mark.parametrize("kind", ["complex", "real"])
def test_sliced_propagate_single_rank_and_composed_slices(oracle, kind):
    """Forward only: the slice products of a 3-way split of the chain, composed on the host, equal dq_propagate."""
    n, K, dt, B, S = 4, 3, 0.04, 9, 10
    H0, Hs, M = _problem(kind, n, K, seed=21)
    u = np.random.default_rng(2).uniform(-1, 1, (S, K))
    psi0 = problems.random_states(1 << n, B, seed=4)
    ref = oracle.propagate(H0, list(Hs), u, dt, psi0)
    with _engine(H0, Hs, M, real_min_dim=1) as e:
        got = time_sliced_propagate(e, u, dt, psi0)
        Ps = [e.slice_forward(u[f:f + c], dt) for f, c in (shard_range(S, r, 3) for r in range(3))]
        Q = e.matmul(e.matmul(Ps[2], Ps[1]), Ps[0])
        got3 = e.apply(Q, psi0)
    assert np.max(np.abs(got - ref)) < TOL and np.max(np.abs(got3 - ref)) < TOL


def test_headline_shape_two_virtual_ranks():
    """BASELINE configs[3] per-rank shape (n = 12, d = 4096, 4096 states per rank, fp64) with the chain of 5 steps split
    over two virtual ranks (3 + 2 steps) and the batch of 8192 over the same two ranks: loss and gradient equal the single
    call on the whole batch (which runs composed: batch > dim), final states equal its columns. One dq_ctx serves both
    virtual ranks here (HBM: a second context would double the 3.5 GB Hamiltonian set and the cache arena), so every
    backward call finds the cache holding the OTHER rank's slice and rebuilds -- the 128-tile kernels, the
    anti-Hermitian-output conjugation R^H W R and the K-split tails at full size."""
    import torch
    from diffquantum_amd.engine import Engine
    n, Bq, S, dt, world = 12, 4096, 5, 1e-3, 2
    d = 1 << n
    dev = torch.device("cuda:0")
    H0, Hs, M = problems.random_ising_torch(n, 0, dev)
    u = torch.from_numpy(problems.pulse_table(n, S, S * dt, seed=0)).to(dev)
    full = torch.from_numpy(problems.random_states(d, world * Bq, seed=1)).to(dev)
    shards = [full[:, r * Bq:(r + 1) * Bq].contiguous() for r in range(world)]
    with Engine(d, n) as e:
        e.use_torch_stream()
        e.set_hamiltonians(H0, Hs)
        e.set_observable(M)
        e.set_option("cache_bytes", 8 * 2 ** 30)
        e.set_option("state_columns", 0)                      # (this test is about the operator form and its time slices)
        l0, g0, f0, _ = e.energy_grad(u, dt, full, want_energies=False)
        assert e.info("real_path") >= 1 and e.info("operator_backward") == 1 and e.info("composed") == 1
        res = virtual_time_sliced_energy_grad([e] * world, u, dt, shards)
    for r, (l, g, f, en) in enumerate(res):
        assert abs(l - l0) < 1e-10 * world * Bq, (r, l, l0)
        assert np.max(np.abs(g - g0)) < 1e-10, r               # gradient entries are ~1e-2
        assert float((f - f0[:, r * Bq:(r + 1) * Bq]).abs().max()) < 1e-12, r
    assert abs(sum(float(x[3].sum()) for x in res) - l0) < 1e-9 * world


def test_headline_dimension_eight_virtual_ranks_narrow_batch():
    """n = 12 (d = 4096) with the chain of 10 steps split over EIGHT virtual ranks (2 or 1 steps each) and 256 states per
    rank: the single call on the 2048-state batch runs in STATE form (batch < 3/4 dim: checkpoints, psi_t rebuilds), the
    sliced program in operator form -- two different algorithms for the same gradient, at the full matrix size, with seven
    slice products composed and every rank's W conjugated by its own suffix product."""
    import torch
    from diffquantum_amd.engine import Engine
    n, Bq, S, dt, world = 12, 256, 10, 1e-3, 8
    d = 1 << n
    dev = torch.device("cuda:0")
    H0, Hs, M = problems.random_ising_torch(n, 0, dev)
    u = torch.from_numpy(problems.pulse_table(n, S, S * dt, seed=3)).to(dev)
    full = torch.from_numpy(problems.random_states(d, world * Bq, seed=4)).to(dev)
    shards = [full[:, r * Bq:(r + 1) * Bq].contiguous() for r in range(world)]
    with Engine(d, n) as e:
        e.use_torch_stream()
        e.set_hamiltonians(H0, Hs)
        e.set_observable(M)
        e.set_option("cache_bytes", 8 * 2 ** 30)
        e.set_option("state_columns", 0)                      # (state form on explicit propagators vs the sliced operator form)
        l0, g0, f0, _ = e.energy_grad(u, dt, full, want_energies=False)
        assert e.info("operator_backward") == 0 and e.info("checkpoint_stride") >= 1
        res = virtual_time_sliced_energy_grad([e] * world, u, dt, shards)
    for r, (l, g, f, en) in enumerate(res):
        assert abs(l - l0) < 1e-10 * world * Bq, (r, l, l0)
        assert np.max(np.abs(g - g0)) < 1e-10, r
        assert float((f - f0[:, r * Bq:(r + 1) * Bq]).abs().max()) < 1e-12, r


def test_new_hamiltonians_between_slice_calls_invalidate_the_cache():
    """dq_set_hamiltonians between dq_slice_forward and dq_slice_backward: the cached intermediates of the old H are dropped."""
    n, K, dt, B, S = 7, 3, 2e-3, 128, 5
    H0, Hs, M = problems.random_real_symmetric_problem(n, K, seed=4)
    H0b, Hsb, _ = problems.random_real_symmetric_problem(n, K, seed=5)
    psi0 = problems.random_states(1 << n, B, seed=1)
    u = np.random.default_rng(3).uniform(-1, 1, (S, K))
    with _engine(H0b, Hsb, M) as e:
        l0, g0, _, _ = e.energy_grad(u, dt, psi0)
        e.set_hamiltonians(H0, Hs)
        e.slice_forward(u, dt)                 # caches the intermediates of (H0, Hs)
        e.set_hamiltonians(H0b, Hsb)
        P = e.slice_forward(u, dt)
        loss, W, _, _ = e.slice_terminal(P, psi0)
        e.set_hamiltonians(H0, Hs)             # ... and back and forth once more before the backward call
        e.set_hamiltonians(H0b, Hsb)
        g1 = e.slice_backward(u, dt, W)
    assert abs(loss - l0) < TOL * B and np.max(np.abs(g1 - g0)) < TOL
