"""Write tests/golden/oracle_s2000.npz: the oracle's answers (oracle.trotter_oracle.energy_grad_one_frechet -- scipy expm +
expm_frechet per step, the reference's arithmetic, diffqc.cc:190-200) over the FULL 2000-step chain for the cases of
tests/s2000_cases.py (d = 256 and 512, 3 states).  Test infrastructure only.

    python tests/golden/make_oracle_s2000_goldens.py [case ...]          # compute (~15 min of host time per case) and write / merge
    python tests/golden/make_oracle_s2000_goldens.py --check [case ...]  # recompute, compare with the committed file, write nothing
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import s2000_cases as cases  # noqa: E402
from oracle import trotter_oracle as oracle  # noqa: E402


def main():
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or list(cases.CASES)
    arrays = {}
    for name in names:
        t0 = time.time()
        P = cases.CASES[name]()
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
    if os.path.exists(cases.GOLDEN):                      # merge: cases not recomputed now keep their stored answers
        old = np.load(cases.GOLDEN)
        for k in old.files:
            arrays.setdefault(k, old[k])
    np.savez_compressed(cases.GOLDEN, **arrays)
    print("wrote {} ({:.0f} KB)".format(cases.GOLDEN, os.path.getsize(cases.GOLDEN) / 1024))


if __name__ == "__main__":
    main()
