"""Mid-dimension soak (run by hand on the GPU box: python tests/soak_mid_dimension.py [S]): S-step chains at d = 1024 -- the smallest
dimension at which the round-2 kernels run (16-wave linear-form complex GEMM with its transposing pre-pass, XCD-slab sparse passes) --
against the oracle's amplitudes, loss and adjoint gradient.  Lives under tests/ because it calls the oracle; not collected by pytest."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from diffquantum_amd import problems  # noqa: E402
from diffquantum_amd.engine import Engine  # noqa: E402
from oracle import trotter_oracle as orc  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n, B, dt = 10, 1024, 2e-3
d = 1 << n
for kind in ("ising", "complex"):
    if kind == "ising":
        H0, Hs, M = problems.random_ising(n, seed=3)
    else:
        rng = np.random.default_rng(4)

        def herm():
            a = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
            return (a + a.conj().T) / (4.0 * np.sqrt(d))

        H0, Hs, M = herm(), np.stack([herm() for _ in range(3)]), herm()
    K = len(Hs)
    u = problems.pulse_table(K, S, S * dt, seed=5)
    psi0 = problems.random_states(d, B, seed=6)
    for compose in (1, 0):
        with Engine(d, K) as e:
            e.set_hamiltonians(H0, Hs)
            e.set_observable(M)
            e.set_option("compose", compose)
            t0 = time.time()
            l, g, f, en = e.energy_grad(u, dt, psi0[:, : (B if compose else 256)])
            t1 = time.time()
            info = {k: e.info(k) for k in ("real_path", "sparse_generator", "operator_backward", "degree", "squarings")}
        if compose:
            lg, gg, fg = l, g, f
    t2 = time.time()
    rl, rg, rf = orc.energy_grad_one_frechet(H0, list(Hs), M, u, dt, psi0)
    t3 = time.time()
    print("d=%d S=%d B=%d %s: gpu %.2f s, oracle %.0f s; |dloss| %.2e (loss %.3f)  max|dgrad| %.2e (max|grad| %.2e)  max|dpsi| %.2e  %s" % (
        d, S, B, kind, t1 - t0, t3 - t2, abs(lg - rl), rl, np.abs(gg - rg).max(), np.abs(rg).max(), np.abs(fg - rf).max(), info), flush=True)
    rl2, rg2, rf2 = orc.energy_grad_one_frechet(H0, list(Hs), M, u, dt, psi0[:, :256]) if False else (None, None, None)
