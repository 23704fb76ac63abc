"""Mid-size regime (the launch-bound side): fwd+adjoint at n = 8 / 10 qubits, QAOA-ring Hamiltonians (real) and random
complex Hermitian ones, per-class profile."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine

dev = torch.device("cuda:0")
for n, B, S, kind in ((8, 1024, 1000, "qaoa"), (8, 256, 1000, "qaoa"), (8, 64, 1000, "qaoa"), (10, 1024, 200, "herm"),
                      (10, 4096, 200, "ising"), (6, 64, 1000, "herm")):
    d = 1 << n
    if kind == "qaoa":
        p = problems.qaoa_ring(n)
        H0, Hs, M = p["H0"].astype(complex), p["Hs"].astype(complex), p["H_cost"].astype(complex)
    elif kind == "ising":
        H0, Hs, M = problems.random_ising(n, 0)
    else:
        H0, Hs, M = problems.random_hermitian_problem(n, 4, seed=1)
    K = len(Hs)
    u = torch.from_numpy(problems.pulse_table(K, S, 1.0, seed=0)).to(dev)
    psi = torch.from_numpy(problems.random_states(d, B, 0)).to(dev)
    with Engine(d, K) as e:
        e.use_torch_stream()
        e.set_hamiltonians(H0, Hs)
        e.set_observable(M)
        import os
        if os.environ.get("DQ_CHUNK"): e.set_option("chunk_bytes", float(os.environ["DQ_CHUNK"]))
        dt = 1.0 / S if kind != "herm" else 0.02
        for mode in ("propagate", "energy_grad"):
            for rep in range(2):
                e.profile_enable(rep == 1); e.profile_reset()
                torch.cuda.synchronize(); t0 = time.time()
                if mode == "propagate":
                    e.propagate(u, dt, psi)
                else:
                    e.energy_grad(u, dt, psi, want_final=False, want_energies=False)
                torch.cuda.synchronize(); t1 = time.time()
            prof = e.profile()
            tot = sum(v["ms"] for v in prof.values()); nl = sum(v["launches"] for v in prof.values())
            print("n=%d d=%d B=%d S=%d %s %s: %.1f ms -> %.0f steps/s; kernels %.1f ms in %d launches; real_path %d opback %d chunk %d" % (
                n, d, B, S, kind, mode, (t1 - t0) * 1e3, S / (t1 - t0), tot, nl, e.info("real_path"),
                e.info("operator_backward") if mode == "energy_grad" else -1, e.info("time_chunk")))
            if mode == "energy_grad":
                print("      " + ", ".join("%s %.1f ms/%d" % (k, v["ms"], v["launches"]) for k, v in prof.items() if v["launches"]))
