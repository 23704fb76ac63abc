"""Data parallelism over the batch of independent trajectories.

One process per GPU.  The B trajectories (columns of Psi0) are split contiguously over the ranks;
Hamiltonians, pulse table and therefore every propagator U_t are replicated, so the chain runs with
NO communication.  Each rank ends with a partial loss and a partial dL/du[S][K] (its columns' share of
the batch sum); ONE all-reduce(sum, fp64) of S*K+1 doubles -- RCCL over xGMI with the "nccl" backend,
gloo on CPU in the tests -- finishes the gradient.  At <= 200 KB the collective is latency-bound.
The reference has no distributed code; this is the only collective the path needs.

Time-sliced operator form (`time_sliced_energy_grad`).  For batch >= 3/4 dim the engine
runs in operator form, where the work of a gradient evaluation is d x d matrix products per step whatever the
batch: replicating the chain on every rank wastes (G-1)/G of the machine.  Instead the chain is split over TIME:
rank r builds only the propagators of its steps [a_r, b_r), the slice products P_r = U_{b_r-1} ... U_{a_r} are
all-gathered (G x 16 d^2 bytes), every rank forms Q = P_{G-1} ... P_0 and applies it to ITS batch shard, the
d x d matrices W_S = Lambda_S Psi_S^H are all-reduced (the only batch reduction), and rank r sweeps its own steps
backward from W_{b_r} = R_r^H W_S R_r with R_r = P_{G-1} ... P_{r+1}.  Three collectives per gradient evaluation
(all-gather of P, all-reduce of W_S, all-reduce of the gradient rows + loss), all issued in the same order with the
same sizes on every rank; per-rank work = 1/G of the chain + (G + 2) extra d x d products.
"""
from __future__ import annotations

import os

import numpy as np


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous [first, first+count) of `n_total` columns owned by `rank` (remainder to the low ranks)."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError("bad rank/world {}/{}".format(rank, world))
    base, rem = divmod(int(n_total), world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def allreduce_loss_grad(loss: float, grad_u, group=None, device=None):
    """Sum (loss, grad_u) over the ranks with a single all-reduce.  grad_u: numpy [S][K] or a torch tensor."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return loss, grad_u
    is_np = isinstance(grad_u, np.ndarray)
    g = torch.from_numpy(np.ascontiguousarray(grad_u, dtype=np.float64)) if is_np else grad_u
    where = device if device is not None else g.device
    if dist.get_backend(group) != "nccl":
        where = "cpu"  # gloo: stage through the host
    buf = torch.empty(g.numel() + 1, dtype=torch.float64, device=where)
    buf[:-1].copy_(g.reshape(-1))
    buf[-1] = float(loss)
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    out = buf[:-1].reshape(g.shape)
    total = float(buf[-1].item())
    if is_np:
        return total, out.cpu().numpy()
    return total, out.to(g.device)


def sharded_energy_grad(energy_grad_fn, psi0_shard, group=None, device=None):
    """Run `energy_grad_fn(psi0_shard) -> (loss, grad_u)` on this rank's columns and all-reduce the result."""
    loss, grad = energy_grad_fn(psi0_shard)
    return allreduce_loss_grad(loss, grad, group=group, device=device)


# ---- time-sliced operator form -------------------------------------------------------------------------------------

def _sliced_phases(engine, u, dt, psi0_shard, rank, world, targets=None, want_final=False, want_energies=False):
    """The per-rank program as a generator: yields (kind, payload) collective requests -- ("all_gather", P) ->
    list of the G slice products in rank order, ("all_reduce", x) -> the sum over ranks (x: [dim][dim] matrix or a
    numpy vector) -- and returns (loss, grad_u[S][K], psi_final, energies).  Every rank yields the same sequence."""
    S = int(u.shape[0])
    K = engine.n_controls
    first, count = shard_range(S, rank, world)
    u_own = u[first:first + count]
    if hasattr(u_own, "contiguous"):
        u_own = u_own.contiguous()
    P = engine.slice_forward(u_own, dt, like=psi0_shard)
    Ps = yield ("all_gather", P)
    # acc runs through P_{G-1}, P_{G-1} P_{G-2}, ...: R_r when it has absorbed the slices after r, Q at the end
    acc, R = None, None
    for j in range(world - 1, -1, -1):
        if j == rank:
            R = acc  # product of the later slices (None = identity for the last rank)
        acc = Ps[j] if acc is None else engine.matmul(acc, Ps[j])
    loss, W, fin, en = engine.slice_terminal(acc, psi0_shard, targets=targets, want_final=want_final,
                                             want_energies=want_energies)
    W = yield ("all_reduce", W)
    if R is not None:
        Z = engine.matmul(W, R)
        W = engine.matmul(R, Z, opA=1, herm=-1 if engine.info("slice_skew") else 0)
    rows = engine.slice_backward(u_own, dt, W)
    buf = np.zeros(S * K + 1, dtype=np.float64)
    buf[first * K:(first + count) * K] = np.asarray(rows, dtype=np.float64).reshape(-1)
    buf[-1] = loss
    buf = yield ("all_reduce", buf)
    return float(buf[-1]), np.array(buf[:-1]).reshape(S, K), fin, en


def _torch_collective(kind, x, group, device):
    """Run one collective request through torch.distributed (nccl: device tensors in place; gloo: staged on the host)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    is_np = isinstance(x, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(x)) if is_np else x
    on_dev = (not is_np) and t.is_cuda
    if backend == "nccl":
        if not t.is_cuda:
            t = t.to(device if device is not None else torch.device("cuda", torch.cuda.current_device()))
    elif t.is_cuda:
        t = t.cpu()
    if t.is_complex():
        tr = torch.view_as_real(t)  # collectives on the (re, im) pairs
    else:
        tr = t
    if kind == "all_gather":
        out = torch.empty((world,) + tuple(tr.shape), dtype=tr.dtype, device=tr.device)
        if backend == "nccl":
            dist.all_gather_into_tensor(out, tr.contiguous(), group=group)  # one RCCL all-gather into the stacked buffer
        else:
            dist.all_gather(list(out.unbind(0)), tr.contiguous(), group=group)
        if t.is_complex():
            out = torch.view_as_complex(out)
        if is_np:
            return list(out.cpu().numpy())
        return list(out.to(x.device) if on_dev else out)
    if kind == "all_reduce":
        tr = tr.contiguous()
        dist.all_reduce(tr, op=dist.ReduceOp.SUM, group=group)
        res = torch.view_as_complex(tr) if t.is_complex() else tr
        if is_np:
            return res.cpu().numpy()
        return res.to(x.device) if on_dev else res
    raise ValueError("unknown collective {}".format(kind))


def time_sliced_energy_grad(engine, u, dt, psi0_shard, group=None, device=None, targets=None, want_final=False,
                            want_energies=False):
    """Loss and dL/du of the WHOLE batch with the chain split over time across the ranks of `group`.

    engine: this rank's Engine (Hamiltonians / observable set identically on every rank); u: the full pulse table
    [S][K] (numpy or a float64 tensor, identical on every rank); psi0_shard: this rank's columns of Psi_0.
    Returns (loss, grad_u[S][K] numpy, psi_final of this rank's columns or None, their energies or None) -- loss and
    gradient are the sums over all ranks' columns, equal (to rounding) to what the batch-sharded replicated form gives.
    Needs the operator form: Hermitian Hamiltonians, and pays off for batch per rank >= 3/4 dim."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        rank, world = 0, 1
    else:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    # DQ_FORCE_COLLECTIVES=1: issue the collectives even at world size 1 (rehearsal of the RCCL calls on a one-GPU box)
    local_only = world == 1 and not (os.environ.get("DQ_FORCE_COLLECTIVES") and dist.is_available() and dist.is_initialized())
    gen = _sliced_phases(engine, u, dt, psi0_shard, rank, world, targets, want_final, want_energies)
    req = next(gen)
    while True:
        kind, x = req
        if local_only:
            res = [x] if kind == "all_gather" else x
        else:
            res = _torch_collective(kind, x, group, device)
        try:
            req = gen.send(res)
        except StopIteration as done:
            return done.value


def time_sliced_propagate(engine, u, dt, psi_shard, group=None, device=None):
    """Forward chain only (dq_propagate) with the propagators built 1/world per rank: all-gather of the slice products,
    Q = P_{G-1} ... P_0, one apply to this rank's columns.  Returns Psi_S of this rank's columns."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        rank, world = 0, 1
    else:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    first, count = shard_range(int(u.shape[0]), rank, world)
    u_own = u[first:first + count]
    if hasattr(u_own, "contiguous"):
        u_own = u_own.contiguous()
    P = engine.slice_forward(u_own, dt, like=psi_shard)
    Ps = [P] if world == 1 else _torch_collective("all_gather", P, group, device)
    acc = Ps[world - 1]
    for j in range(world - 2, -1, -1):
        acc = engine.matmul(acc, Ps[j])
    return engine.apply(acc, psi_shard)


def virtual_time_sliced_energy_grad(engines, u, dt, psi0_shards, targets=None):
    """The same program for G = len(engines) VIRTUAL ranks inside one process (tests, single-GPU rehearsal): the rank
    programs advance in lockstep and the collectives are carried out on the spot.  engines[r] holds rank r's context
    (all with the same Hamiltonians), psi0_shards[r] its columns.  Returns the list of per-rank results."""
    world = len(engines)
    gens = [_sliced_phases(engines[r], u, dt, psi0_shards[r], r, world, None if targets is None else targets[r],
                           want_final=True, want_energies=True) for r in range(world)]
    reqs = [next(g) for g in gens]
    results = [None] * world
    while any(r is not None for r in reqs):
        kinds = {r[0] for r in reqs if r is not None}
        if len(kinds) != 1 or any(r is None for r in reqs):
            raise RuntimeError("virtual ranks diverged: {}".format([None if r is None else r[0] for r in reqs]))
        kind = kinds.pop()
        xs = [r[1] for r in reqs]
        if kind == "all_gather":
            res = [list(xs) for _ in range(world)]
        else:
            tot = xs[0].copy() if isinstance(xs[0], np.ndarray) else xs[0].clone()
            for x in xs[1:]:
                tot += x
            res = [tot.copy() if isinstance(tot, np.ndarray) else tot.clone() for _ in range(world)]
        nxt = []
        for r, g in enumerate(gens):
            try:
                nxt.append(g.send(res[r]))
            except StopIteration as done:
                results[r] = done.value
                nxt.append(None)
        reqs = nxt
    return results
