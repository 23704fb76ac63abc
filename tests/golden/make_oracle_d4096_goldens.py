"""Write tests/golden/oracle_d4096.npz: the oracle's answers (oracle.trotter_oracle.energy_grad_one_frechet -- scipy expm +
expm_frechet, the reference's arithmetic) for the three d = 4096 cases of tests/d4096_cases.py.  Test infrastructure only.

    python tests/golden/make_oracle_d4096_goldens.py            # compute (several minutes of host time) and write
    python tests/golden/make_oracle_d4096_goldens.py --check    # recompute, compare with the committed file, write nothing
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import d4096_cases as cases  # noqa: E402
from oracle import trotter_oracle as oracle  # noqa: E402


def main():
    arrays = {}
    for name, make in cases.CASES.items():
        t0 = time.time()
        P = make()
        a = cases.compute(P, oracle)
        arrays[name + "/signature"] = cases.signature(P)
        arrays[name + "/loss"] = np.array(a["loss"])
        arrays[name + "/grad"] = a["grad"]
        arrays[name + "/fin"] = a["fin"]
        arrays[name + "/energies"] = a["energies"]
        print("{}: oracle answer in {:.0f} s (loss {:.12f})".format(name, time.time() - t0, a["loss"]), flush=True)
    if "--check" in sys.argv:
        old = np.load(cases.GOLDEN)
        worst = 0.0
        for k, v in arrays.items():
            worst = max(worst, float(np.max(np.abs(np.asarray(old[k]) - v))))
        print("largest difference from {}: {:.3e}".format(cases.GOLDEN, worst))
        sys.exit(0 if worst < 1e-12 else 1)
    np.savez_compressed(cases.GOLDEN, **arrays)
    print("wrote {} ({:.0f} KB)".format(cases.GOLDEN, os.path.getsize(cases.GOLDEN) / 1024))


if __name__ == "__main__":
    main()
