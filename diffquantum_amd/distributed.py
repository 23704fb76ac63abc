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

Column-sharded form of the wide-batch energy gradient (`column_sharded_energy_grad`, round 5; BASELINE configs[3]: 32768 states over
8 GPUs).  For a batch wider than the dimension and a Pauli-sum generator one GPU already runs the whole job on dim columns (engine.hip
do_energy_grad_xs_wide): L = tr(M P R P^H) depends on the states through R = sum_b psi0_b psi0_b^H alone.  Sharding the BATCH therefore
shortens nothing -- every rank would sweep all dim columns for its own R.  Instead: rank r forms the Gram matrix of its states, ONE
all-reduce makes R (16 d^2 bytes: 268 MB at d = 4096, ~2 ms per ring step over xGMI), every rank builds P (replicated: 1/3 of a
forward pass over dim columns) and sweeps dim / G of the columns of (P R, 2 M P); one all-reduce of (loss, dL/du).  The adjoint sweep --
2/3 of the single-GPU time -- is split G ways: 1 / (1/3 + 2/(3 G)) = 2.4x at G = 8 with P replicated.  With `slice_product` (default: where it pays, `slice_product_pays`)
P is built in time slices as well: rank r runs the forward kernel over ITS S / G steps (Engine.slice_forward), the G slice products
are all-gathered (G x 16 d^2 bytes) and multiplied on every rank (G - 1 products of 8 d^3 flop: 8 ms each at d = 4096) -- at configs[3]
(2000 steps) 0.46 s of replicated forward pass become 0.06 s + ~0.07 s of products and gather: ~5.8x at G = 8 by the single-GPU
kernel times (1.74 s for the whole job on one GPU).
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


# ---- column-sharded wide-batch gradient --------------------------------------------------------------------------------

def slice_product_pays(n_steps, dim, world):
    """The chain's product in time slices costs world - 1 products of 8 dim^3 flop on every rank (and the all-gather) and saves
    (1 - 1/world) of a forward pass over dim columns: by the single-GPU kernel times (0.23 ms per forward step, 8 ms per product at
    dim = 4096; both ratios scale with dim) it pays beyond ~45 steps per rank at dim = 4096."""
    return world > 1 and n_steps > 45.0 * world * dim / 4096.0


def _columns_phases(engine, u, dt, psi0_shard, rank, world, slice_product=None):
    """The per-rank program of the column-sharded form as a generator of collective requests (as _sliced_phases): the Gram matrix of
    this rank's states, all-reduce; with slice_product the chain's product in TIME SLICES (this rank's steps through
    Engine.slice_forward, all-gather, G - 1 products on every rank) instead of a whole forward pass on every rank; the sweep over
    this rank's columns; all-reduce of (dL/du, loss)."""
    S, K = int(u.shape[0]), engine.n_controls
    R = engine.gram(psi0_shard)
    R = yield ("all_reduce", R)
    P = None
    if slice_product is None:
        slice_product = slice_product_pays(S, engine.dim, world)
    if slice_product and world > 1:
        first, count = shard_range(S, rank, world)
        u_own = u[first:first + count]
        if hasattr(u_own, "contiguous"):
            u_own = u_own.contiguous()
        Ps = yield ("all_gather", engine.slice_forward(u_own, dt, like=psi0_shard))
        P = Ps[0]
        for j in range(1, world):          # P = P_{G-1} ... P_0
            P = engine.matmul(Ps[j], P)
    first, count = shard_range(engine.dim, rank, world)
    loss, grad = engine.energy_grad_columns(u, dt, R, first, count, P=P)
    buf = np.empty(S * K + 1, dtype=np.float64)
    buf[:-1] = np.asarray(grad, dtype=np.float64).reshape(-1)
    buf[-1] = loss
    buf = yield ("all_reduce", buf)
    return float(buf[-1]), np.array(buf[:-1]).reshape(S, K), None, None


def _drive(gen, world, group, device):
    """Run one rank's generator of collective requests through torch.distributed (locally at world size 1)."""
    import torch.distributed as dist
    local_only = world == 1 and not (os.environ.get("DQ_FORCE_COLLECTIVES") and dist.is_available() and dist.is_initialized())
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


def column_sharded_energy_grad(engine, u, dt, psi0_shard, group=None, device=None, slice_product=None):
    """Loss and dL/du of the WHOLE batch (energy objective) with the adjoint sweep split over the dim COLUMNS of (P R, 2 M P) across
    the ranks of `group`; psi0_shard: this rank's states (any split of the batch, empty shards excluded).  Returns
    (loss, grad_u[S][K] numpy) -- the sums over all ranks' states, equal to rounding to engine.energy_grad of the whole batch.
    Needs a Pauli-sum generator at a dimension the state-column kernels hold and dq_set_observable."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        rank, world = 0, 1
    else:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    loss, grad, _, _ = _drive(_columns_phases(engine, u, dt, psi0_shard, rank, world, slice_product), world, group, device)
    return loss, grad


def virtual_column_sharded_energy_grad(engines, u, dt, psi0_shards, slice_product=None):
    """The same program for len(engines) VIRTUAL ranks in one process (tests, single-GPU rehearsal); returns the per-rank results."""
    world = len(engines)
    return _run_virtual([_columns_phases(engines[r], u, dt, psi0_shards[r], r, world, slice_product) for r in range(world)])


# ---- SURVEY.md 8-F2 in its literal form: the propagators of a chain built 1/G per rank and all-gathered ---------------------

def _window_owner(t0, n_win, per_rank, world):
    """Steps [t0, t0 + n_win) of a window: rank r owns the per_rank consecutive steps starting at t0 + r * per_rank
    (fewer, possibly none, in the last window). Returns [(first, count)] per rank."""
    spans = []
    for r in range(world):
        first = min(t0 + r * per_rank, t0 + n_win)
        spans.append((first, max(0, min(per_rank, t0 + n_win - first))))
    return spans


def _propagator_sharded_phases(engine, M, u, dt, psi0_shard, rank, world, per_rank=1, want_final=False, want_energies=False):
    """State-form fwd + adjoint of THIS rank's columns on propagators that are built 1/world per rank (generator of
    collective requests, as _sliced_phases). Per window of world * per_rank steps:
      forward : every rank builds the propagators of its own steps (dq_build_propagators), all-gather, apply in order;
      backward: the window's propagators are built and gathered again, the states of the window recomputed from the window's
                checkpoint, then for t descending  G_t = lambda_{t+1} psi_t^H (this rank's columns), lambda_t = U_t^H lambda_{t+1};
                the G_t are summed over the ranks and scattered to their owners (reduce-scatter), which pull them back to their
                own pulse rows (dq_propagator_vjp).
    One all-reduce of the zero-padded gradient rows + loss at the end. Every rank issues the same sequence."""
    S, K, d = int(u.shape[0]), engine.n_controls, engine.dim
    u = np.ascontiguousarray(u.cpu().numpy() if hasattr(u, "cpu") else u, dtype=np.float64)
    win = world * per_rank
    windows = [(t0, min(win, S - t0)) for t0 in range(0, S, win)]

    def gather_window(t0, n_win):
        spans = _window_owner(t0, n_win, per_rank, world)
        first, count = spans[rank]
        Umine = engine.build_propagators(u[first:first + count], dt, like=psi0_shard)      # [count][d][d]
        pad = _like_zeros(Umine, (per_rank, d, d))
        if count:
            pad[:count] = Umine
        return spans, pad

    psi = psi0_shard
    checkpoints = []
    for t0, n_win in windows:                                  # ---- forward ----
        spans, pad = gather_window(t0, n_win)
        Ugath = yield ("all_gather", pad)                      # list over ranks of [per_rank][d][d]
        checkpoints.append(psi)
        for r in range(world):
            for j in range(spans[r][1]):
                psi = engine.gemm(Ugath[r][j], psi)
    energies, loss = engine.energy(psi)
    lam = engine.gemm(M, psi, alpha=2.0)                       # Lambda_S = 2 M Psi_S
    fin = psi
    grad = np.zeros((S, K), dtype=np.float64)
    for (t0, n_win), ck in zip(reversed(windows), reversed(checkpoints)):   # ---- backward ----
        spans, pad = gather_window(t0, n_win)
        Ugath = yield ("all_gather", pad)
        Us = [Ugath[r][j] for r in range(world) for j in range(spans[r][1])]            # the window's steps in order
        states = [ck]
        for U in Us[:-1]:
            states.append(engine.gemm(U, states[-1]))
        Gs = _like_zeros(pad, (world * per_rank, d, d))
        for i in range(len(Us) - 1, -1, -1):
            r, j = divmod(i, per_rank)
            engine.gemm(lam, states[i], opB=1, out=Gs[r * per_rank + j])               # G_t = lambda_{t+1} psi_t^H
            lam = engine.gemm(Us[i], lam, opA=1)                                       # lambda_t = U_t^H lambda_{t+1}
        Gmine = yield ("reduce_scatter", Gs)                   # [per_rank][d][d]: the sums over all ranks' columns, own steps
        first, count = spans[rank]
        if count:
            grad[first:first + count] = engine.propagator_vjp(u[first:first + count], dt, Gmine[:count])
    buf = np.zeros(S * K + 1, dtype=np.float64)
    buf[:-1] = grad.reshape(-1)
    buf[-1] = loss
    buf = yield ("all_reduce", buf)
    return float(buf[-1]), np.array(buf[:-1]).reshape(S, K), (fin if want_final else None), (energies if want_energies else None)


def _like_zeros(like, shape):
    if isinstance(like, np.ndarray):
        return np.zeros(shape, dtype=like.dtype)
    import torch
    return torch.zeros(shape, dtype=like.dtype, device=like.device)


def propagator_sharded_energy_grad(engine, M, u, dt, psi0_shard, group=None, device=None, per_rank=1, want_final=False,
                                   want_energies=False):
    """SURVEY.md section 8-F2 as written: "each GPU builds S/G propagators and all-gathers them" -- for the STATE-form regime
    (batch per rank < 3/4 dim), where every rank needs every U_t for its own columns and the replicated form would make every
    rank compute every exponential. M: the observable given to engine.set_observable (an array of the kind of psi0_shard).
    Returns (loss, grad_u[S][K], psi_final of this rank's columns or None, their energies or None); loss and gradient are the
    sums over all ranks' columns. Communication per step: 16 d^2 bytes all-gathered twice (forward, backward) and reduce-scattered
    once -- at d = 4096 that is 0.8 GB per step, which is why the time-sliced OPERATOR form (three collectives per gradient
    evaluation) is the faster choice wherever it applies; this form exists for completeness and for
    generators that are not Hermitian-checked operator-form candidates."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        rank, world = 0, 1
    else:
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    # DQ_FORCE_COLLECTIVES=1: issue the collectives even at world size 1 (rehearsal of the RCCL calls on a one-GPU box)
    local_only = world == 1 and not (os.environ.get("DQ_FORCE_COLLECTIVES") and dist.is_available() and dist.is_initialized())
    gen = _propagator_sharded_phases(engine, M, u, dt, psi0_shard, rank, world, per_rank, want_final, want_energies)
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


def virtual_propagator_sharded_energy_grad(engines, M, u, dt, psi0_shards, per_rank=1):
    """The same program for len(engines) VIRTUAL ranks inside one process (tests, single-GPU rehearsal)."""
    world = len(engines)
    gens = [_propagator_sharded_phases(engines[r], M, u, dt, psi0_shards[r], r, world, per_rank, True, True) for r in range(world)]
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
            if kind == "reduce_scatter":
                n = tot.shape[0] // world
                res = [tot[r * n:(r + 1) * n] for r in range(world)]
            else:
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
    if kind == "reduce_scatter":  # x: [world * n][...] -> the sum over ranks of this rank's block [n][...]
        tr = tr.contiguous()
        n = tr.shape[0] // world
        if backend == "nccl":
            out = torch.empty((n,) + tuple(tr.shape[1:]), dtype=tr.dtype, device=tr.device)
            dist.reduce_scatter_tensor(out, tr, op=dist.ReduceOp.SUM, group=group)
        else:
            dist.all_reduce(tr, op=dist.ReduceOp.SUM, group=group)
            r = dist.get_rank(group)
            out = tr[r * n:(r + 1) * n].contiguous()
        res = torch.view_as_complex(out) if t.is_complex() else out
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
    return _run_virtual([_sliced_phases(engines[r], u, dt, psi0_shards[r], r, world, None if targets is None else targets[r],
                                        want_final=True, want_energies=True) for r in range(world)])


def _run_virtual(gens):
    """Advance the rank programs `gens` in lockstep, carrying out their collectives on the spot."""
    world = len(gens)
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
