"""fp32 gradient fidelity of the matrix-free engine's LDS-DMA forms over a 4000-step chain (n = 10 / 12, B = 32) against the
complex128 dense engine, and -- optionally -- the directional-derivative check of the n = 18 chain with three finite-difference steps.
    python tools/config5_fp32_fidelity.py [n ...]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine
from diffquantum_amd.structured import StructuredEngine, ising_x_terms, terms_to_dense

for n in [int(a) for a in sys.argv[1:]] or [10, 12]:
    K, S, B, dt = n, 4000, 32, 1e-3
    terms = ising_x_terms(n, seed=21)
    u = problems.pulse_table(K, S, S * dt, seed=22)
    psi0 = problems.random_states(1 << n, B, seed=23).astype(np.complex64)
    H0, Hs, M = terms_to_dense(terms, n, K)
    with Engine(1 << n, K) as de:
        de.set_hamiltonians(H0, Hs); de.set_observable(M)
        rl, rg, rf, _ = de.energy_grad(u, dt, psi0.astype(np.complex128))
    sb = (1 << n) * B * 8
    for levels in (2, 1, 0):
        with StructuredEngine(n, K, dtype="c64") as e:
            e.set_terms(terms)
            e.energy_grad(u[:4], dt, psi0)
            m = int(e.info("degree")); s2 = max(2, round((S / m) ** 0.5))
            if levels == 2:
                e.set_option("checkpoint_bytes", sb * (-(-S // s2) + m * s2 + 1))
            elif levels == 1:
                e.set_option("two_level", 0); e.set_option("checkpoint_bytes", sb * 100)
            t0 = time.time(); l1, g1, f1, _ = e.energy_grad(u, dt, psi0); el = time.time() - t0
            print("n={} levels={} two_level={} stride={} degree={}: {:.2f} s  max|f-f64| {:.2e}  |loss-l64| {:.2e} (B={})  grad rel err {:.2e} (max|g| {:.2e})".format(
                n, levels, int(e.info("two_level")), int(e.info("checkpoint_stride")), m, el, np.max(np.abs(f1 - rf)), abs(l1 - rl), B,
                np.max(np.abs(g1 - rg)) / np.max(np.abs(rg)), np.max(np.abs(rg))), flush=True)
