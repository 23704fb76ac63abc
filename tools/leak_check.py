import sys, numpy as np, torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine
free0 = torch.cuda.mem_get_info()[0]
for it in range(6):
    n = 9
    H0, Hs, M = problems.random_ising(n, it)
    with Engine(1 << n, n) as e:
        e.set_hamiltonians(H0, Hs); e.set_observable(M)
        u = problems.pulse_table(n, 50, 0.05, seed=it)
        psi = problems.random_states(1 << n, 600, it)
        l, g, f, en = e.energy_grad(u, 1e-3, psi)
        e.set_option("real_path", 0)
        l2, g2, f2, en2 = e.energy_grad(u, 1e-3, psi)
        e.set_option("small_dim", 0); e.set_option("compose", 0)
        l3, g3, f3, en3 = e.energy_grad(u, 1e-3, psi[:, :100])
    free = torch.cuda.mem_get_info()[0]
    print(it, "loss %.12f |g1-g2| %.2e free delta %.1f MB" % (l, np.abs(g - g2).max(), (free0 - free) / 1e6))
