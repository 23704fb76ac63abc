"""The composed forward pass of wide batches from state-column segments (engine.hip xs_compose_segments) against the explicit-propagator
tree of the same engine: time and difference (development tool).   python tools/compose_check.py"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine

p = problems.qaoa_ring(8)
H0, Hs, M = p["H0"].astype(np.complex128), p["Hs"].astype(np.complex128), p["H_cost"].astype(np.complex128)
K, d = len(Hs), 256


def timed(e, u, dt, psi0, reps=5):
    e.propagate(u, dt, psi0)
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); f = e.propagate(u, dt, psi0); best = min(best, time.perf_counter() - t0)
    return f, best


for S, B in ((1000, 1024), (37, 300), (1, 513)):
    u = problems.pulse_table(K, S, 1e-3 * S, seed=3)
    psi0 = problems.random_states(d, B, seed=1)
    with Engine(d, K) as e:
        e.set_hamiltonians(H0, Hs); e.set_observable(M)
        out = {}
        for mode, var in ((-1, 0), (-1, 1), (0, 0)):
            e.set_option("state_columns_compose", mode)
            e.set_option("state_columns_variant", var)
            f, el = timed(e, u, 1e-3, psi0)
            out[(mode, var)] = (f, el, e.info("state_columns_compose_segments"), e.info("degree"))
        print("configs[1] shape, S", S, "B", B, ": segments", out[(-1, 0)][2], "degree", out[(-1, 0)][3], "(tree:", out[(0, 0)][3], ")",
              "time %.2f ms (variant 1: %.2f ms) vs explicit-propagator tree %.2f ms" % (out[(-1, 0)][1] * 1e3, out[(-1, 1)][1] * 1e3, out[(0, 0)][1] * 1e3),
              " max diff %.2e" % np.max(np.abs(out[(-1, 0)][0] - out[(0, 0)][0])), flush=True)
for n, S, B in ((10, 50, 3000), (12, 6, 5000)):
    H0, Hs, M = problems.random_ising(n, 4)
    u = problems.pulse_table(n, S, 1e-3 * S, seed=3); psi0 = problems.random_states(1 << n, B, seed=1)
    with Engine(1 << n, n) as e:
        e.set_hamiltonians(H0, Hs); e.set_observable(M)
        e.set_option("compose", 1)
        out = {}
        for mode in (-1, 0):
            e.set_option("state_columns_compose", mode)
            f, el = timed(e, u, 1e-3, psi0, 2)
            out[mode] = (f, el, e.info("state_columns_compose_segments"))
        print("n", n, "S", S, "B", B, "segments", out[-1][2], "time %.2f ms vs %.2f ms" % (out[-1][1] * 1e3, out[0][1] * 1e3),
              "max diff %.2e" % np.max(np.abs(out[-1][0] - out[0][0])), flush=True)
