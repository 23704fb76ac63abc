"""Rates of the shapes beyond the state-column kernels behind the dense API (development tool): n qubits, dtype, batch B, S steps --
the matrix-free route (auto) against the explicit-propagator chain of the same context (matrix_free = 0).
    python tools/perf_route.py n dtype B S [skip_dense]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine

n, dtype, B, S = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
skip_dense = len(sys.argv) > 5 and sys.argv[5] == "1"
dev = torch.device("cuda:0")
d = 1 << n
H0, Hs, M = problems.random_ising_torch(n, 0, dev)
if dtype == "c64":
    H0, Hs, M = H0.to(torch.complex64), Hs.to(torch.complex64), M.to(torch.complex64)
u = torch.from_numpy(problems.pulse_table(n, S, 1e-3 * S, seed=0)).to(dev)
psi0 = torch.from_numpy(problems.random_states(d, B, 0).astype(np.complex64 if dtype == "c64" else np.complex128)).to(dev)
with Engine(d, n, dtype=dtype) as e:
    e.use_torch_stream()
    e.set_hamiltonians(H0, Hs)
    e.set_observable(M)
    del H0, Hs, M
    torch.cuda.empty_cache()
    e.set_option("plan_steps", 2000)
    for name, mf in [("matrix-free route", -1)] + ([] if skip_dense else [("explicit-propagator chain", 0)]):
        e.set_option("matrix_free", mf)
        e.energy_grad(u, 1e-3, psi0, want_final=False, want_energies=False)     # warm-up over the same chain: checkpoint pools allocated here
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, g, _, _ = e.energy_grad(u, 1e-3, psi0, want_final=False, want_energies=False)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        extra = ""
        if e.info("matrix_free"):
            extra = "  passes/step {:.1f}  dma passes/step {:.1f}  degree {}  stride {}".format(
                e.info("matrix_free_passes") / S, e.info("matrix_free_dma_passes") / S, int(e.info("degree")), int(e.info("checkpoint_stride")))
        print("n={} {} B={} S={}: {:28s} matrix_free={}  {:9.2f} ms/step  {:8.2f} steps/s  loss {:.9f}{}".format(
            n, dtype, B, S, name, int(e.info("matrix_free")), el / S * 1e3, S / el, loss, extra), flush=True)
