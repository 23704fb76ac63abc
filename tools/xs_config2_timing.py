"""BASELINE configs[1] (8-qubit ring QAOA, d = 256, 1000 steps, batch 1024, forward) and its fwd+adjoint variant: state-column
form against the explicit-propagator forms (development tool)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
p = problems.qaoa_ring(8)
u = torch.from_numpy(problems.pulse_table(16, S, 1.0, seed=0)).to(dev)
psi = torch.from_numpy(problems.random_states(256, B, 0)).to(dev)
with Engine(256, 16) as e:
    e.use_torch_stream()
    e.set_hamiltonians(p["H0"].astype(np.complex128), p["Hs"].astype(np.complex128))
    e.set_observable(p["H_cost"].astype(np.complex128))
    outs = {}
    for xs, var in ((1, 0), (1, 1), (1, 3), (0, 0)):
        e.set_option("state_columns", xs)
        e.set_option("state_columns_variant", var)
        e.propagate(u, 1.0 / S, psi)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fin = e.propagate(u, 1.0 / S, psi)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        el = float(np.median(ts))
        outs[xs] = fin.clone()
        print("forward   state_columns={} variant {} (took {}): {:8.3f} ms  {:10.0f} steps/s  degree {}".format(xs, var, int(e.info("state_columns")), el * 1e3, S / el, int(e.info("degree"))), flush=True)
    print("max |diff| of the two forward results", float((outs[1] - outs[0]).abs().max()))
    Sg = min(S, 200)
    for xs, var in ((1, 0), (1, 1), (0, 0)):
        e.set_option("state_columns", xs)
        e.set_option("state_columns_variant", var)
        e.energy_grad(u[:Sg], 1.0 / S, psi, want_final=False, want_energies=False)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            l, g, _, _ = e.energy_grad(u[:Sg], 1.0 / S, psi, want_final=False, want_energies=False)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        el = float(np.median(ts))
        outs[xs] = (l, g)
        print("fwd+adj   state_columns={} variant {} (took {}): {:8.3f} ms  {:10.0f} steps/s".format(xs, var, int(e.info("state_columns")), el * 1e3, Sg / el), flush=True)
    print("grad diff", float(np.max(np.abs(outs[1][1] - outs[0][1]))), "loss diff", abs(outs[1][0] - outs[0][0]))
