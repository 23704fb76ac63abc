"""What the chain-aware bound of the step polynomial is for (development study, complex64): a 4000-step chain at a step size where the
economized polynomial meets the PER-STEP bound 6e-8 with degree 3 (2 J_4(theta) = 2e-8) and Taylor needs degree 4 (theta^5 / 120 = 1.5e-9)
-- against the complex128 dense engine.  The economized error is the same function of the spectrum at every step (to leading order the
real number -2 J_4(theta) T_4(H / ||H||): a change of NORM of each eigen-component), so it adds up over the chain; the chain-aware bound
(chain_tol / steps) takes degree 4 there.       python tools/poly_fidelity.py [n] [S] [theta]"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine
from diffquantum_amd.structured import StructuredEngine, ising_x_terms, terms_to_dense

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
S = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
theta = float(sys.argv[3]) if len(sys.argv) > 3 else 0.045
K, B = n, 32
terms = ising_x_terms(n, seed=21)
u = problems.pulse_table(K, S, S * 1e-3, seed=22)
psi0 = problems.random_states(1 << n, B, seed=23).astype(np.complex64)
with StructuredEngine(n, K, dtype="c64") as e:
    e.set_terms(terms)
    e.propagate(u, 1e-3, psi0)
    dt = 1e-3 * theta / e.info("theta")
H0, Hs, M = terms_to_dense(terms, n, K)
with Engine(1 << n, K) as de:
    de.set_hamiltonians(H0, Hs); de.set_observable(M)
    rl, rg, rf, _ = de.energy_grad(u, dt, psi0.astype(np.complex128))
print("n = {}  S = {}  B = {}  theta = {}  (dt = {:.3e})".format(n, S, B, theta, dt))
for name, opts in (("Taylor, per-step bound only", {"polynomial": 0, "chain_tol": 1e9}),
                   ("economized, per-step bound only", {"polynomial": 1, "chain_tol": 1e9}),
                   ("economized, chain-aware bound (default)", {"polynomial": 1})):
    with StructuredEngine(n, K, dtype="c64") as e:
        e.set_terms(terms)
        for k, v in opts.items():
            e.set_option(k, v)
        t0 = time.time(); l1, g1, f1, _ = e.energy_grad(u, dt, psi0); el = time.time() - t0
        nrm = np.linalg.norm(f1.astype(np.complex128), axis=0)
        print("{:42s} degree {} (Taylor {}): {:.2f} s  max|f-f64| {:.2e}  max| |psi_b| - 1 | {:.2e}  |loss-l64| {:.2e}  grad rel err {:.2e}".format(
            name, int(e.info("degree")), int(e.info("taylor_degree")), el, np.max(np.abs(f1 - rf)), np.max(np.abs(nrm - 1)), abs(l1 - rl),
            np.max(np.abs(g1 - rg)) / np.max(np.abs(rg))), flush=True)
