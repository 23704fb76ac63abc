"""Data parallelism over the batch of independent trajectories.

One process per GPU.  The B trajectories (columns of Psi0) are split contiguously over the ranks;
Hamiltonians, pulse table and therefore every propagator U_t are replicated, so the chain runs with
NO communication.  Each rank ends with a partial loss and a partial dL/du[S][K] (its columns' share of
the batch sum); ONE all-reduce(sum, fp64) of S*K+1 doubles -- RCCL over xGMI with the "nccl" backend,
gloo on CPU in the tests -- finishes the gradient.  At <= 200 KB the collective is latency-bound.
The reference has no distributed code; this is the only collective the path needs.
"""
from __future__ import annotations

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
    buf = torch.empty(g.numel() + 1, dtype=torch.float64, device=device if device is not None else g.device)
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
