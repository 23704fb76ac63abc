"""Energy gradient of a batch wider than the dimension: the adjoint sweep over the B states against the sweep over dim columns
(option state_columns_wide_adjoint; engine.hip do_energy_grad_xs_wide).   python tools/wide_adjoint_timing.py [n] [B] [S]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32768
S = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dev = torch.device("cuda:0")
d = 1 << n
H0, Hs, M = problems.random_ising_torch(n, 0, dev)
u = torch.from_numpy(problems.pulse_table(n, S, 1e-3 * S, seed=0)).to(dev)
psi0 = torch.from_numpy(problems.random_states(d, B, 0)).to(dev)
with Engine(d, n) as e:
    e.use_torch_stream(); e.set_hamiltonians(H0, Hs); e.set_observable(M)
    del H0, Hs, M
    out = {}
    for mode in (0, 1, -1):
        e.set_option("state_columns_wide_adjoint", mode)
        e.energy_grad(u, 1e-3, psi0, want_final=False, want_energies=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        l, g, _, _ = e.energy_grad(u, 1e-3, psi0, want_final=False, want_energies=False)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        out[mode] = (l, g, el, e.info("state_columns_wide_adjoint"))
        print("n = {} B = {} S = {}  option {:2d}: wide form {}  {:.2f} ms per step ({:.1f} steps/s)  loss {:.9f}".format(
            n, B, S, mode, int(out[mode][3]), el / S * 1e3, S / el, l), flush=True)
    print("max |dL/du (dim columns) - dL/du (states)| = {:.2e}  (max |dL/du| {:.2e})".format(
        float(np.max(np.abs(out[1][1] - out[0][1]))), float(np.max(np.abs(out[0][1])))))
