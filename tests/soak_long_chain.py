"""Long-chain soak (run by hand on the GPU box: python tests/soak_long_chain.py): 20000-step chains against the oracle.
Lives under tests/ because it calls the oracle (test infrastructure); not collected by pytest."""
import sys, time, numpy as np
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine
from oracle import trotter_oracle as orc
for (n, K, B, S, dt, real) in ((4, 3, 5, 20000, 0.002, False), (4, 3, 5, 20000, 0.002, True), (5, 2, 40, 8000, 0.004, True),
                              (4, 4, 20, 20000, 0.002, "ising"), (5, 5, 3, 8000, 0.004, "ising")):
    if real == "ising":  # sparse controls (X_k): the scatter assembly and the row-dot-product gradient level
        H0, Hs, M = problems.random_ising(n, seed=11)
    else:
        H0, Hs, M = (problems.random_real_symmetric_problem if real else problems.random_hermitian_problem)(n, K, seed=11)
    u = problems.pulse_table(K, S, S * dt, seed=5)
    psi0 = problems.random_states(1 << n, B, seed=6)
    with Engine(1 << n, K) as e:
        e.set_hamiltonians(H0, Hs); e.set_observable(M); e.set_option("real_min_dim", 1); e.set_option("sparse_generator", 1)
        t0 = time.time(); l, g, f, en = e.energy_grad(u, dt, psi0); t1 = time.time()
        info = (e.info("real_path"), e.info("operator_backward"), e.info("time_parallel"), e.info("time_chunk"), e.info("cached_steps"), e.info("sparse_controls"), e.info("sparse_generator"))
    t2 = time.time(); rl, rg, rf = orc.energy_grad(H0, list(Hs), M, u, dt, psi0); t3 = time.time()
    print("n=%d S=%d real=%s: gpu %.2f s, oracle %.1f s; |dl| %.2e  max|dg| %.2e  max|dpsi| %.2e  info %s" % (
        n, S, real, t1 - t0, t3 - t2, abs(l - rl), np.abs(g - rg).max(), np.abs(f - rf).max(), info))
