"""Write tests/golden/oracle_d8192.npz: the oracle's answer (oracle.trotter_oracle.energy_grad_one_frechet: scipy expm + expm_frechet,
the reference's arithmetic, diffqc.cc:190-200) for the d = 8192 case of tests/d8192_case.py.  Test infrastructure only.
    python tests/golden/make_oracle_d8192_golden.py [--check]       # ~15 minutes, ~20 GB of host memory"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))
import d8192_case as case  # noqa: E402
from oracle import trotter_oracle as oracle  # noqa: E402

t0 = time.time()
H0, Hs, M = case.dense_inputs()
loss, grad, fin = oracle.energy_grad_one_frechet(H0, list(Hs), M, case.pulse(), case.DT, case.states())
en = np.asarray(oracle.energies(M, fin), dtype=np.float64)
arrays = dict(signature=case.signature(), loss=np.array(float(loss)), grad=np.asarray(grad), fin=np.asarray(fin), energies=en)
print("oracle answer in {:.0f} s (loss {:.12f}, max |grad| {:.3e})".format(time.time() - t0, loss, np.max(np.abs(grad))), flush=True)
if "--check" in sys.argv:
    old = np.load(case.GOLDEN)
    worst = max(float(np.max(np.abs(np.asarray(old[k]) - v))) for k, v in arrays.items())
    print("largest difference from {}: {:.3e}".format(case.GOLDEN, worst))
    sys.exit(0 if worst < 1e-12 else 1)
np.savez_compressed(case.GOLDEN, **arrays)
print("wrote {} ({:.0f} KB)".format(case.GOLDEN, os.path.getsize(case.GOLDEN) / 1024))
