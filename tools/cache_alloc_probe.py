import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine
n, B = 12, 4096; d = 1 << n
dev = torch.device("cuda:0")
H0, Hs, M = problems.random_ising_torch(n, 0, dev)
psi0 = torch.from_numpy(problems.random_states(d, B, 0)).to(dev)
e = Engine(d, n); e.use_torch_stream(); e.set_hamiltonians(H0, Hs); e.set_observable(M)
u2 = torch.from_numpy(problems.pulse_table(n, 2, 2e-3, 0)).to(dev)
e.set_option("cache_bytes", -1)
e.energy_grad(u2, 1e-3, psi0, want_final=False, want_energies=False)   # pool allocations
e.set_option("cache_bytes", 0)
for S in (25, 100, 250):
    u = torch.from_numpy(problems.pulse_table(n, S, 1e-3 * S, 0)).to(dev)
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time()
        e.energy_grad(u, 1e-3, psi0, want_final=False, want_energies=False)
        torch.cuda.synchronize(); t1 = time.time()
        print("S=%d rep %d: %.3f s -> %.2f steps/s, cached %d" % (S, rep, t1 - t0, S / (t1 - t0), e.info("cached_steps")))
