#!/usr/bin/env python3
"""Single-GPU rehearsal of ONE rank's program of the time-sliced operator form at a given world size: the collectives are
answered locally (the all-gather returns this rank's slice product for every slice, the all-reduces return their input),
so the timing is the rank's compute: K/G steps built, composed and swept + (G + 2) d x d products + terminal products.
    python tools/perf_sliced.py [n=12] [B=4096] [K=20] [world=8] [rank=0]
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffquantum_amd import problems  # noqa: E402
from diffquantum_amd.distributed import _sliced_phases  # noqa: E402
from diffquantum_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
K = int(sys.argv[3]) if len(sys.argv) > 3 else 20
world = int(sys.argv[4]) if len(sys.argv) > 4 else 8
rank = int(sys.argv[5]) if len(sys.argv) > 5 else 0
dt = 1e-3
dev = torch.device("cuda", 0)
d = 1 << n
H0, Hs, M = problems.random_ising_torch(n, 0, dev)
u = torch.from_numpy(problems.pulse_table(n, K, dt * K, seed=0)).to(dev)
psi0 = torch.from_numpy(problems.random_states(d, B, 0)).to(dev)
eng = Engine(d, n, device=0)
eng.use_torch_stream()
eng.set_hamiltonians(H0, Hs)
eng.set_observable(M)
slice_steps = (2000 + world - 1) // world
eng.set_option("plan_steps", slice_steps)
if len(sys.argv) > 6 and sys.argv[6] == "fullcache":  # as bench.py's sliced form: the whole slice of the planned chain cached
    eng.set_option("cache_bytes", min(8.0 * d * d * 8 * slice_steps + (1 << 20),
                                      0.72 * torch.cuda.get_device_properties(0).total_memory))


def one():
    gen = _sliced_phases(eng, u, dt, psi0, rank, world)
    req = next(gen)
    marks = [time.perf_counter()]
    while True:
        kind, x = req
        torch.cuda.synchronize()
        marks.append(time.perf_counter())
        res = [x] * world if kind == "all_gather" else x
        try:
            req = gen.send(res)
        except StopIteration as done:
            torch.cuda.synchronize()
            marks.append(time.perf_counter())
            return done.value, np.diff(marks)


one()
torch.cuda.synchronize()
t0 = time.perf_counter()
(loss, grad, _, _), ph = one()
el = time.perf_counter() - t0
print("n={} B={} K={} world={} rank={}: {:.1f} ms per gradient evaluation -> {:.1f} steps/s per rank-chain, "
      "{:.1f} aggregate (world*K/t); phases [forward+compose, gather->terminal, reduce->backward] ms = {}; cached {}".format(
          n, B, K, world, rank, el * 1e3, K / el, world * K / el, [round(p * 1e3, 1) for p in ph], eng.info("cached_steps")))
free, total = torch.cuda.mem_get_info()
print("HBM in use after the run: {:.1f} GB of {:.1f}".format((total - free) / 1e9, total / 1e9))
if os.environ.get("DQ_SKIP_SINGLE"):
    sys.exit(0)
# the same chain in one call (the N = 1 form)
eng.set_option("plan_steps", 2000)
eng.set_option("cache_bytes", 0)
eng.energy_grad(u[:2], dt, psi0, want_final=False, want_energies=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
eng.energy_grad(u, dt, psi0, want_final=False, want_energies=False)
torch.cuda.synchronize()
el1 = time.perf_counter() - t0
print("single call, {} steps: {:.1f} ms -> {:.1f} steps/s".format(K, el1 * 1e3, K / el1))
