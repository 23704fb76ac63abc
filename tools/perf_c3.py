"""Quick C3-shape timing (n=12, d=4096, B=4096) with the per-class HIP-event profile."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
S = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dtype = sys.argv[4] if len(sys.argv) > 4 else "c128"
d = 1 << n
dev = torch.device("cuda:0")
H0, Hs, M = problems.random_ising_torch(n, 0, dev)
if dtype == "c64":
    H0, Hs, M = H0.to(torch.complex64), Hs.to(torch.complex64), M.to(torch.complex64)
u = problems.pulse_table(n, S, 1e-3 * S, seed=0)
psi0 = torch.from_numpy(problems.random_states(d, min(B, 64), 0)).to(dev).repeat(1, (B + 63) // 64)[:, :B].contiguous()
if dtype == "c64":
    psi0 = psi0.to(torch.complex64)
e = Engine(d, n, dtype=dtype)
e.use_torch_stream()
e.set_hamiltonians(H0, Hs)
e.set_observable(M)
import os
e.set_option("zgemm_alg", int(os.environ.get("DQ_ALG", "0")))
tu = torch.from_numpy(u).to(dev)
for mode in ("propagate", "energy_grad"):
    for rep in range(2):
        e.profile_enable(rep == 1)
        e.profile_reset()
        torch.cuda.synchronize(); t0 = time.time()
        if mode == "propagate":
            e.propagate(tu, 1e-3, psi0)
        else:
            loss, grad, fin, en = e.energy_grad(tu, 1e-3, psi0, want_final=False, want_energies=False)
        torch.cuda.synchronize(); t1 = time.time()
        print("  rep %d (profiling %s): wall %.4f s" % (rep, "on" if rep == 1 else "off", t1 - t0))
    prof = e.profile()
    z = prof["zgemm"]
    print(mode, "S=%d wall %.3f s -> %.3f steps/s; degree %d sq %d theta %.3g ckstride %d" % (
        S, t1 - t0, S / (t1 - t0), e.info("degree"), e.info("squarings"), e.info("theta"), e.info("checkpoint_stride")))
    print("  zgemm: %d launches, %.1f ms total, %.2f TFLOP/s avg, %.2f ms/launch" % (
        z["launches"], z["ms"], z["flops"] / z["ms"] * 1e-9, z["ms"] / max(1, z["launches"])))
    for k in ("assemble", "lincomb", "gradc", "other"):
        p = prof[k]
        if p["launches"]:
            print("  %s: %d launches, %.2f ms total, %.1f GB/s" % (k, p["launches"], p["ms"], p["bytes"] / p["ms"] * 1e-6))
