"""world_size-2 gloo test of the batch sharding + single all-reduce (the N>1 path of bench.py).
The per-shard compute is the ORACLE here (tests may use it); on GPUs it is Engine.energy_grad."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_range_partitions():
    from diffquantum_amd.distributed import shard_range
    for n in (0, 1, 7, 4096, 32768):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (f0, c0), (f1, _) in zip(spans, spans[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    with pytest.raises(ValueError):
        shard_range(4, 2, 2)


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from diffquantum_amd import problems
    from diffquantum_amd.distributed import shard_range, sharded_energy_grad
    from oracle import trotter_oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H0, Hs, M = problems.random_hermitian_problem(2, 2, seed=5)
    u = np.random.default_rng(1).uniform(-1, 1, (6, 2))
    B = 5
    first, count = shard_range(B, rank, world)
    psi = problems.random_states(4, count, seed=9, first=first)

    def fn(p):
        loss, grad, _ = orc.energy_grad(H0, list(Hs), M, u, 0.1, p)
        return loss, grad

    loss, grad = sharded_energy_grad(fn, psi)
    q.put((rank, loss, grad))
    dist.destroy_process_group()


def test_two_rank_gloo_allreduce_equals_single_process():
    import torch.multiprocessing as mp
    from diffquantum_amd import problems
    from oracle import trotter_oracle as orc
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    H0, Hs, M = problems.random_hermitian_problem(2, 2, seed=5)
    u = np.random.default_rng(1).uniform(-1, 1, (6, 2))
    full = problems.random_states(4, 5, seed=9)
    loss, grad, _ = orc.energy_grad(H0, list(Hs), M, u, 0.1, full)
    for rank, l, g in res:
        assert abs(l - loss) < 1e-12 and np.max(np.abs(g - grad)) < 1e-12


# ---- time-sliced operator form: the rank program through real gloo collectives -------------------

def test_virtual_ranks_time_sliced_equals_oracle():
    """G virtual ranks in one process (numpy phase engine = oracle arithmetic): chain split over time, batch over ranks."""
    from diffquantum_amd import problems
    from diffquantum_amd.distributed import shard_range, virtual_time_sliced_energy_grad
    from oracle import trotter_oracle as orc
    from tests.np_slice_engine import NumpySliceEngine
    H0, Hs, M = problems.random_hermitian_problem(2, 2, seed=5)
    dt = 0.1
    B = 7
    full = problems.random_states(4, B, seed=9)
    for S in (0, 1, 6, 7):
        u = np.random.default_rng(S).uniform(-1, 1, (S, 2))
        loss, grad, fin = orc.energy_grad(H0, list(Hs), M, u, dt, full)
        for world in (1, 2, 3, 8):  # world > S: ranks with empty slices; world > B would leave ranks without columns
            if world > B:
                continue
            for skew in (True, False):
                engines = [NumpySliceEngine(H0, Hs, M, skew) for _ in range(world)]
                shards = [full[:, f:f + c] for f, c in (shard_range(B, r, world) for r in range(world))]
                res = virtual_time_sliced_energy_grad(engines, u, dt, shards)
                for r, (l, g, f, e) in enumerate(res):
                    assert abs(l - loss) < 1e-12
                    assert g.shape == (S, 2) and (S == 0 or np.max(np.abs(g - grad)) < 1e-12)
                    f0, c0 = shard_range(B, r, world)
                    assert np.max(np.abs(f - fin[:, f0:f0 + c0])) < 1e-12


def _sliced_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from diffquantum_amd import problems
    from diffquantum_amd.distributed import shard_range, time_sliced_energy_grad
    from tests.np_slice_engine import NumpySliceEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H0, Hs, M = problems.random_hermitian_problem(2, 2, seed=5)
    u = np.random.default_rng(1).uniform(-1, 1, (7, 2))
    B = 5
    first, count = shard_range(B, rank, world)
    psi = problems.random_states(4, count, seed=9, first=first)
    out = []
    for _ in range(2):  # twice: the collectives stay matched call after call
        loss, grad, fin, en = time_sliced_energy_grad(NumpySliceEngine(H0, Hs, M), u, 0.1, psi, want_final=True,
                                                      want_energies=True)
        out.append((loss, grad))
    q.put((rank, out, fin, first, count))
    dist.destroy_process_group()


def test_two_rank_gloo_time_sliced_equals_single_process():
    import torch.multiprocessing as mp
    from diffquantum_amd import problems
    from oracle import trotter_oracle as orc
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sliced_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    H0, Hs, M = problems.random_hermitian_problem(2, 2, seed=5)
    u = np.random.default_rng(1).uniform(-1, 1, (7, 2))
    full = problems.random_states(4, 5, seed=9)
    loss, grad, fin = orc.energy_grad(H0, list(Hs), M, u, 0.1, full)
    for rank, out, f, first, count in res:
        for l, g in out:
            assert abs(l - loss) < 1e-12 and np.max(np.abs(g - grad)) < 1e-12
        assert np.max(np.abs(f - fin[:, first:first + count])) < 1e-12


def _propagator_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from diffquantum_amd import problems
    from diffquantum_amd.distributed import propagator_sharded_energy_grad, shard_range
    from tests.np_slice_engine import NumpySliceEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H0, Hs, M = problems.random_hermitian_problem(2, 2, seed=5)
    u = np.random.default_rng(1).uniform(-1, 1, (7, 2))
    B = 5
    first, count = shard_range(B, rank, world)
    psi = problems.random_states(4, count, seed=9, first=first)
    out = []
    for per_rank in (1, 2):  # 7 steps: windows of 2 / 4 steps, the last one ragged (a rank without a step)
        loss, grad, fin, en = propagator_sharded_energy_grad(NumpySliceEngine(H0, Hs, M), M, u, 0.1, psi, per_rank=per_rank,
                                                             want_final=True, want_energies=True)
        out.append((loss, grad))
    q.put((rank, out, fin, first, count))
    dist.destroy_process_group()


def test_two_rank_gloo_propagator_sharded_equals_single_process():
    """SURVEY 8-F2 in its literal form (propagators built 1/G per rank, all-gathered; cotangents reduce-scattered to their
    owners) through real gloo collectives, with the oracle's arithmetic as the per-rank engine."""
    import torch.multiprocessing as mp
    from diffquantum_amd import problems
    from oracle import trotter_oracle as orc
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_propagator_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    H0, Hs, M = problems.random_hermitian_problem(2, 2, seed=5)
    u = np.random.default_rng(1).uniform(-1, 1, (7, 2))
    full = problems.random_states(4, 5, seed=9)
    loss, grad, fin = orc.energy_grad(H0, list(Hs), M, u, 0.1, full)
    for rank, out, f, first, count in res:
        for l, g in out:
            assert abs(l - loss) < 1e-12 and np.max(np.abs(g - grad)) < 1e-12
        assert np.max(np.abs(f - fin[:, first:first + count])) < 1e-12


def test_virtual_ranks_propagator_sharded_numpy():
    from diffquantum_amd import problems
    from diffquantum_amd.distributed import shard_range, virtual_propagator_sharded_energy_grad
    from oracle import trotter_oracle as orc
    from tests.np_slice_engine import NumpySliceEngine
    H0, Hs, M = problems.random_hermitian_problem(3, 2, seed=7)
    u = np.random.default_rng(2).uniform(-1, 1, (5, 2))
    full = problems.random_states(8, 7, seed=3)
    loss, grad, fin = orc.energy_grad(H0, list(Hs), M, u, 0.07, full)
    for world, per_rank in ((1, 1), (3, 1), (3, 2), (4, 3)):
        shards = [full[:, f:f + c] for f, c in (shard_range(7, r, world) for r in range(world))]
        res = virtual_propagator_sharded_energy_grad([NumpySliceEngine(H0, Hs, M) for _ in range(world)], M, u, 0.07, shards,
                                                     per_rank=per_rank)
        for r, (l, g, f, en) in enumerate(res):
            f0, c0 = shard_range(7, r, world)
            assert abs(l - loss) < 1e-12 and np.max(np.abs(g - grad)) < 1e-12
            assert np.max(np.abs(f - fin[:, f0:f0 + c0])) < 1e-12


# ---- column-sharded wide-batch gradient (round 5; BASELINE configs[3]) ---------------------------------------------------------------

def test_virtual_ranks_column_sharded_equals_oracle():
    """G virtual ranks (numpy twin = the oracle's arithmetic): any split of the batch, the dim columns of (P R, 2 M P) split over the
    ranks -- more ranks than columns included -- give the oracle's loss and gradient of the whole batch on every rank."""
    from diffquantum_amd import problems
    from diffquantum_amd.distributed import shard_range, virtual_column_sharded_energy_grad
    from oracle import trotter_oracle as orc
    from tests.np_slice_engine import NumpySliceEngine
    H0, Hs, M = problems.random_hermitian_problem(2, 2, seed=5)
    dt, B = 0.1, 11                                  # batch wider than the dimension (4)
    full = problems.random_states(4, B, seed=9)
    for S in (1, 5):
        u = np.random.default_rng(S).uniform(-1, 1, (S, 2))
        loss, grad, _ = orc.energy_grad(H0, list(Hs), M, u, dt, full)
        for world in (1, 2, 3, 6):                   # 6 ranks, 4 columns: two ranks sweep nothing
            engines = [NumpySliceEngine(H0, Hs, M) for _ in range(world)]
            shards = [full[:, f:f + c] for f, c in (shard_range(B, r, world) for r in range(world))]
            for slice_product in (True, False):      # the chain's product in time slices (one per rank, gathered) / built by every rank
                for l, g, _, _ in virtual_column_sharded_energy_grad(engines, u, dt, shards, slice_product=slice_product):
                    assert abs(l - loss) < 1e-11 and g.shape == (S, 2) and np.max(np.abs(g - grad)) < 1e-11


def _columns_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from diffquantum_amd import problems
    from diffquantum_amd.distributed import column_sharded_energy_grad, shard_range
    from tests.np_slice_engine import NumpySliceEngine
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    H0, Hs, M = problems.random_hermitian_problem(2, 2, seed=5)
    u = np.random.default_rng(1).uniform(-1, 1, (6, 2))
    B = 9
    first, count = shard_range(B, rank, world)
    psi = problems.random_states(4, count, seed=9, first=first)
    out = [column_sharded_energy_grad(NumpySliceEngine(H0, Hs, M), u, 0.1, psi) for _ in range(2)]   # twice: collectives stay matched
    q.put((rank, out))
    dist.destroy_process_group()


def test_two_rank_gloo_column_sharded_equals_single_process():
    import torch.multiprocessing as mp
    from diffquantum_amd import problems
    from oracle import trotter_oracle as orc
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_columns_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    H0, Hs, M = problems.random_hermitian_problem(2, 2, seed=5)
    u = np.random.default_rng(1).uniform(-1, 1, (6, 2))
    loss, grad, _ = orc.energy_grad(H0, list(Hs), M, u, 0.1, problems.random_states(4, 9, seed=9))
    for rank, out in res:
        for l, g in out:
            assert abs(l - loss) < 1e-11 and np.max(np.abs(g - grad)) < 1e-11
