"""Config-5 regime timing: n-qubit Ising + X controls as Pauli strings, matrix-free, complex64 (HBM-bound)."""
import sys, time, json
import numpy as np, torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.structured import StructuredEngine, ising_x_terms

n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = int(sys.argv[3]) if len(sys.argv) > 3 else 20
dtype = sys.argv[4] if len(sys.argv) > 4 else "c64"
dt = 1e-3
d = 1 << n
dev = torch.device("cuda:0")
e = StructuredEngine(n, n, dtype=dtype)
e.use_torch_stream()
e.set_terms(ising_x_terms(n, 0))
rng = np.random.default_rng(0)
psi = rng.standard_normal((d, B)) + 1j * rng.standard_normal((d, B))
psi /= np.linalg.norm(psi, axis=0)
tp = torch.from_numpy(psi.astype(np.complex64 if dtype == "c64" else np.complex128)).to(dev)
u = problems.pulse_table(n, S, S * dt, seed=0)
state_bytes = d * B * (8 if dtype == "c64" else 16)
for mode in ("propagate", "energy_grad"):
    for rep in range(2):
        e.set_option("profile", -1)
        e.set_option("profile", 1 if rep else 0)
        torch.cuda.synchronize(); t0 = time.time()
        if mode == "propagate":
            out = e.propagate(u, dt, tp)
        else:
            loss, grad, fin, en = e.energy_grad(u, dt, tp, want_final=False)
        torch.cuda.synchronize(); t1 = time.time()
    ms, by, nl = e.info("prof_ms"), e.info("prof_bytes"), e.info("prof_launches")
    print("%s n=%d B=%d S=%d %s: wall %.3f s -> %.1f steps/s; degree %d, groups %d, theta %.3g, ckstride %d" % (
        mode, n, B, S, dtype, t1 - t0, S / (t1 - t0), e.info("degree"), e.info("groups"), e.info("theta"),
        e.info("checkpoint_stride")))
    print("  passes: %d launches, %.2f ms, algorithmic %.2f GB -> %.2f TB/s (state = %.1f MB; %.1f state-units per step)" % (
        nl, ms, by / 1e9, by / ms * 1e-9, state_bytes / 1e6, by / state_bytes / S))
if mode == "energy_grad":
    print("  loss %.6f grad norm %.4e" % (loss, np.linalg.norm(grad)))
