"""Times the state-column form against the engine's explicit-propagator forms on one GPU (development tool; not a test).
    python tools/xs_timing.py [n] [B] [S]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
S = int(sys.argv[3]) if len(sys.argv) > 3 else 20
VARIANTS = [int(v) for v in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
OTHER = int(sys.argv[5]) if len(sys.argv) > 5 else 1
XY = int(sys.argv[6]) if len(sys.argv) > 6 else 0     # 1: X drives on half of the qubits, Y drives on the other half (complex table); 2: X / Z controls
ADJ = [int(v) for v in sys.argv[7].split(",")] if len(sys.argv) > 7 else [-1]
DTYPE = sys.argv[8] if len(sys.argv) > 8 else "c128"   # state_columns_adjoint: 1 / -1 quadrature form, 0 exact form
dev = torch.device("cuda:0")
d = 1 << n
H0, Hs, M = problems.xy_drives_torch(n, 0, dev, n_x=n // 2, n_y=n - n // 2) if XY == 1 else problems.random_ising_torch(n, 0, dev)
if XY == 2:     # X_k on the even qubits, Z_k (diagonal controls: per-step diagonal weights, row accumulators in the adjoint sweep) on the odd ones
    idx = torch.arange(d, device=dev)
    for q in range(1, n, 2):
        Hs[q].zero_()
        Hs[q, idx, idx] = (1.0 - 2.0 * ((idx >> (n - 1 - q)) & 1)).to(torch.complex128)
u = torch.from_numpy(problems.pulse_table(n, S, 1e-3 * S, seed=0)).to(dev)
psi0 = torch.from_numpy(problems.random_states(d, B, 0)).to(dev)
if DTYPE == "c64":
    H0, Hs, M, psi0 = H0.to(torch.complex64), Hs.to(torch.complex64), M.to(torch.complex64), psi0.to(torch.complex64)
res = {}
with Engine(d, n, dtype=DTYPE) as e:
    e.use_torch_stream()
    e.set_hamiltonians(H0, Hs)
    e.set_observable(M)
    del H0, Hs, M
    runs = [("state_columns v{} adj{}".format(v, j), 1, v, j) for v in VARIANTS for j in ADJ] + ([("other", 0, 0, -1)] if OTHER else [])
    for name, xs, var, adj in runs:
        e.set_option("state_columns", xs)
        e.set_option("state_columns_variant", var)
        e.set_option("state_columns_adjoint", adj)
        g = torch.zeros((S, n), dtype=torch.float64, device=dev)
        e.energy_grad(u[:2].contiguous(), 1e-3, psi0, want_final=False, want_energies=False)
        e.profile_enable(True); e.profile_reset()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            loss, _, _, _ = e.energy_grad(u, 1e-3, psi0, want_final=False, want_energies=False, grad_out=g)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        prof = e.profile(); e.profile_enable(False)
        el = float(np.median(ts))
        res["state_columns" if xs else "other"] = (loss, g.clone())
        print("{:24s} state_columns={} quad nodes {}: {:8.3f} ms/step  {:9.1f} steps/s   loss {:.12f}  degree {} theta {:.4f}".format(
            name, int(e.info("state_columns")), int(e.info("state_columns_quad_nodes")), el / S * 1e3, S / el, loss, int(e.info("degree")),
            e.info("theta")), flush=True)
        for k, v in prof.items():
            if v["launches"]:
                print("      {:9s} launches {:4d}  ms {:9.3f}  ms/launch {:8.3f}  GB/s(bytes) {:9.1f}  TF/s {:7.2f}".format(
                    k, v["launches"], v["ms"], v["ms"] / v["launches"], v["bytes"] / v["ms"] * 1e-6, v["flops"] / v["ms"] * 1e-9), flush=True)
    # forward only
    for name, xs, var in [("fwd state_columns v{}".format(v), 1, v) for v in VARIANTS] + ([("fwd other", 0, 0)] if OTHER else []):
        e.set_option("state_columns", xs)
        e.set_option("state_columns_variant", var)
        e.propagate(u, 1e-3, psi0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.propagate(u, 1e-3, psi0)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        print("{:24s}: {:8.3f} ms/step {:9.1f} steps/s".format(name, el / S * 1e3, S / el), flush=True)
if OTHER:
    print("max |grad diff|", float((res["state_columns"][1] - res["other"][1]).abs().max()), "loss diff", abs(res["state_columns"][0] - res["other"][0]))
