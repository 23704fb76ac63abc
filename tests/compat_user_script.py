"""A user script written against the REFERENCE's module names only (`qutip`, `sim_plain`, the way
/root/reference/demo_maxcut.py:2,4 imports them), run by tests/test_gpu_compat_route.py in a child process with
`compat/` first on PYTHONPATH.  Own code: the 4-cycle MaxCut instance is built here from Pauli matrices through the
simulator's `multi_kron`, trained for a few epochs, and the trace is printed as one JSON line."""
import json
import sys

import numpy as np
import qutip as qp
import diffqc
import logger
from sim_plain import SimulatorPlain as Sim

n_epoch, log_dir = int(sys.argv[1]), sys.argv[2]
np.random.seed(0)
sim = Sim(lr=2e-2, n_basis=6, n_epoch=n_epoch, log_dir=log_dir)
assert isinstance(sim.logger, logger.Logger)
nq, edges = 4, [(0, 1), (0, 3), (1, 2), (2, 3)]
P = {"I": np.eye(2), "X": np.array([[0.0, 1.0], [1.0, 0.0]]), "Z": np.diag([1.0, -1.0])}
zz = [sim.multi_kron(*[P["Z"] if q in e else P["I"] for q in range(nq)]) for e in edges]
xs = [sim.multi_kron(*[P["X"] if q == v else P["I"] for q in range(nq)]) for v in range(nq)]
acc = sum(np.eye(1 << nq) - m for m in zz)          # the demo's aliased accumulator: H0 = sum_e (I - Z_i Z_j)
sim.T = np.pi * (1.0 / np.pi + 1.0 / np.pi)
sim.omegas = [np.pi] * (len(zz) + len(xs))
psi0 = qp.Qobj(np.full(1 << nq, 0.25))
sim.train_energy(qp.Qobj(-0.5 * acc), qp.Qobj(acc), [qp.Qobj(h) for h in zz + xs], psi0)
state, prob = sim.find_state(sim.final_state)
print(json.dumps({"losses": [float(x) for x in sim.losses_energy], "state": int(state),
                  "prob": [float(p) for p in prob], "final": [[z.real, z.imag] for z in sim.final_state.full()[:, 0]],
                  "modules": {"sim_plain": sys.modules["sim_plain"].__file__, "qutip": qp.__file__,
                              "diffqc": diffqc.__file__, "logger": logger.__file__},
                  "diffqc_version": diffqc.__version__}))
