"""Real-symmetric fast path at the C3 shape (n=12, d=4096, B=4096): real GEMM kernel configurations, then the
fwd+adjoint chain with the real path on (auto) and off, per-class HIP-event profile."""
import sys, time, ctypes as C
import numpy as np, torch
sys.path.insert(0, ".")
from diffquantum_amd import problems, _lib
from diffquantum_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
S = int(sys.argv[3]) if len(sys.argv) > 3 else 4
d = 1 << n
dev = torch.device("cuda:0")
H0, Hs, M = problems.random_ising_torch(n, 0, dev)
u = problems.pulse_table(n, S, 1e-3 * S, seed=0)
psi0 = torch.from_numpy(problems.random_states(d, min(B, 64), 0)).to(dev).repeat(1, (B + 63) // 64)[:, :B].contiguous()
e = Engine(d, n)
e.use_torch_stream()
e.set_hamiltonians(H0, Hs)
e.set_observable(M)
print("real_hamiltonians", e.info("real_hamiltonians"))
lib = _lib.load()

# --- bare real GEMM kernel ---
X = torch.randn(d, d, dtype=torch.float64, device=dev); X = (X + X.T) / (2 * d ** 0.5)
Y = X @ X
Cout = torch.empty_like(X)
for cfg in (1, 2, 3):
    e.set_option("rgemm_cfg", cfg)
    for sym in (0, 1):
        for rep in range(2):
            e.profile_enable(True); e.profile_reset()
            for _ in range(5):
                _lib.check(lib.dq_rgemm(e._ctx, 1, 0, d, d, d, 1.0, X.data_ptr(), d, 0, Y.data_ptr(), d, 0, 0.0,
                                        Cout.data_ptr(), d, 0, 1, sym))
            p = e.profile()["rsym" if sym else "rgemm"]
        err = float((Cout - X @ Y).abs().max())
        print("rgemm cfg %d sym %d: %.3f ms/launch, %.1f TF/s algorithmic (2d^3), max err %.2e" % (
            cfg, sym, p["ms"] / p["launches"], p["flops"] / p["ms"] * 1e-9, err))
e.profile_enable(False)

tu = torch.from_numpy(u).to(dev)
import os
combos = [(3, -1, 1), (2, -1, 1), (3, 0, 1)]
for cfg, rp, alg in combos:
  if True:
    e.set_option("rgemm_cfg", cfg)
    e.set_option("real_path", rp)
    e.set_option("zgemm_alg", alg)
    print("---- rgemm_cfg %d real_path %d zgemm_alg %d" % (cfg, rp, alg))
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
        prof = e.profile()
        print("cfg %d real_path %d (q used %d) %s S=%d wall %.3f s -> %.3f steps/s; degree %d products %d" % (
            cfg, rp, e.info("real_path"), mode, S, t1 - t0, S / (t1 - t0), e.info("degree"), e.info("products")))
        if mode == "energy_grad":
            print("   loss %.12f |grad| %.12e" % (loss, float(np.linalg.norm(grad))))
        for k in ("zgemm", "rgemm", "rsym", "assemble", "lincomb", "gradc", "other"):
            p = prof[k]
            if p["launches"]:
                print("   %s: %d launches, %.2f ms total, %.3f ms/launch, %.1f GB/s, %.1f TF/s" % (
                    k, p["launches"], p["ms"], p["ms"] / p["launches"], p["bytes"] / p["ms"] * 1e-6, p["flops"] / p["ms"] * 1e-9))
