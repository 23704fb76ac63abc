"""4-vertex MaxCut by analog QAOA on the MI355X propagator engine.

Counterpart of the reference's demo (/root/reference/demo_maxcut.py): same problem, same optimiser
settings, same final print -- written against this package's API.

    python examples/demo_maxcut.py [--epochs 202] [--gradient mc|adjoint] [--seed 0]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from diffquantum_amd import problems  # noqa: E402
from diffquantum_amd.qobj import Qobj  # noqa: E402
from diffquantum_amd.sim_plain import SimulatorPlain  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=202)
    ap.add_argument("--gradient", default="mc", choices=["mc", "adjoint"])
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--log-dir", default=None)
    a = ap.parse_args()
    if a.seed is not None:
        np.random.seed(a.seed)
    p = problems.maxcut_problem()
    sim = SimulatorPlain(lr=2e-2, n_basis=6, n_epoch=a.epochs, gradient=a.gradient, log_dir=a.log_dir)
    sim.T = p["T"]
    sim.omegas = list(p["omegas"])
    sim.logger.write_text("sim.T: {}".format(sim.T))
    sim.train_energy(Qobj(p["H_cost"]), Qobj(p["H0"]), [Qobj(h) for h in p["Hs"]], Qobj(p["psi0"]))
    state, prob = sim.find_state(sim.final_state)
    print("cut result is ", bin(state)[2:])
    return state


if __name__ == "__main__":
    main()
