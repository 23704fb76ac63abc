#!/usr/bin/env python3
"""Operator form: the cost of a gradient evaluation does not depend on the batch.  The headline chain
(n = 12, d = 4096, fp64) at B = 4096, 8192, 32768 states on ONE GPU -- B = 32768 is BASELINE configs[3]'s whole batch.
    python tools/perf_batch_independence.py [K=20]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffquantum_amd import problems  # noqa: E402
from diffquantum_amd.engine import Engine  # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
n, dt = 12, 1e-3
d = 1 << n
dev = torch.device("cuda", 0)
H0, Hs, M = problems.random_ising_torch(n, 0, dev)
u = torch.from_numpy(problems.pulse_table(n, K, dt * K, seed=0)).to(dev)
with Engine(d, n) as e:
    e.use_torch_stream()
    e.set_hamiltonians(H0, Hs)
    e.set_observable(M)
    e.set_option("plan_steps", 2000)
    for B in (4096, 8192, 32768):
        base = torch.from_numpy(problems.random_states(d, 256, 0)).to(dev)
        psi0 = base.repeat(1, B // 256).contiguous()
        e.energy_grad(u[:2], dt, psi0, want_final=False, want_energies=False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss, g, _, _ = e.energy_grad(u, dt, psi0, want_final=False, want_energies=False)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        print("B = {:6d}: {:7.1f} ms for {} steps fwd+adjoint = {:5.1f} steps/s = {:9.0f} trajectory-steps/s "
              "(composed forward: {}, operator backward: {}); loss/B = {:.12f}".format(
                  B, el * 1e3, K, K / el, B * K / el, int(e.info("composed")), int(e.info("operator_backward")), loss / B))
        del psi0
