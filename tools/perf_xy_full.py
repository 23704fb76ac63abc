"""X and Y drives on EVERY qubit at n = 12 (24 controls, 24 XOR-diagonals: the 32-accumulator reverse kernels of the complex column
form) against the complex tape: fwd + adjoint steps/s on a 20-step chain, B = 4096."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine

dev = torch.device("cuda:0")
n, B, S, dt = 12, 4096, 20, 1e-3
H0, Hs, M = problems.xy_drives_torch(n, 0, dev, n_x=n, n_y=n)
K = Hs.shape[0]
u = torch.from_numpy(problems.pulse_table(K, S, 2000 * dt, seed=0)[:S].copy()).to(dev)
psi = problems.random_states_torch(1 << n, B, 0, dev) if hasattr(problems, "random_states_torch") else torch.from_numpy(problems.random_states(1 << n, B, 0)).to(dev)
with Engine(1 << n, K) as e:
    e.use_torch_stream()
    e.set_hamiltonians(H0, Hs)
    e.set_observable(M)
    del H0, Hs, M
    res = {}
    for mode in (1, 0):
        e.set_option("xor_regular", mode)
        g = torch.zeros((S, K), dtype=torch.float64, device=dev)
        e.energy_grad(u[:1].contiguous(), dt, psi, want_final=False, want_energies=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, _, _, _ = e.energy_grad(u, dt, psi, want_final=False, want_energies=False, grad_out=g)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        res[mode] = (loss, g.clone())
        print("xor_regular=%d (info %d, %d diagonals): %.2f steps/s (%.2f ms/step), loss %.12f" % (
            mode, e.info("xor_regular"), e.info("xor_regular_found"), S / el, el / S * 1e3, loss), flush=True)
    print("max |grad diff| = %.2e, |loss diff| = %.2e" % (float((res[1][1] - res[0][1]).abs().max()), abs(res[1][0] - res[0][0])))
