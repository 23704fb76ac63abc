"""A/B timing of two builds of the library on the same box: python tools/ab_timing.py <package root> n B S dtype [xy]
(the package root is put first on sys.path: e.g. a scratch tree of an older commit built under tools/_build/)."""
import sys, time
root = sys.argv[1]
sys.path.insert(0, root)
import numpy as np
import torch
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine
import diffquantum_amd
n, B, S, dtype = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
xy = len(sys.argv) > 6 and sys.argv[6] == "1"
dev = torch.device("cuda:0")
d = 1 << n
H0, Hs, M = problems.xy_drives_torch(n, 0, dev, n_x=n // 2, n_y=n - n // 2) if xy else problems.random_ising_torch(n, 0, dev)
psi0 = torch.from_numpy(problems.random_states(d, B, 0)).to(dev)
if dtype == "c64":
    H0, Hs, M, psi0 = H0.to(torch.complex64), Hs.to(torch.complex64), M.to(torch.complex64), psi0.to(torch.complex64)
u = torch.from_numpy(problems.pulse_table(n, S, 1e-3 * S, seed=0)).to(dev)
with Engine(d, n, dtype=dtype) as e:
    e.use_torch_stream()
    e.set_hamiltonians(H0, Hs); e.set_observable(M)
    del H0, Hs, M
    g = torch.zeros((S, n), dtype=torch.float64, device=dev)
    e.energy_grad(u[:2].contiguous(), 1e-3, psi0, want_final=False, want_energies=False)
    e.profile_enable(True); e.profile_reset()
    ts = []
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, _, _, _ = e.energy_grad(u, 1e-3, psi0, want_final=False, want_energies=False, grad_out=g)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    prof = e.profile()
    el = float(np.median(ts))
    print("{}: n={} B={} S={} {} xy={}: {:.3f} ms/step {:.1f} steps/s  state_columns={}  fwd {:.3f} ms/launch  bwd {:.3f} ms/launch".format(
        diffquantum_amd.__file__.replace("/__init__.py", "")[-40:], n, B, S, dtype, int(xy), el / S * 1e3, S / el, int(e.info("state_columns")),
        prof["statecol_fwd"]["ms"] / max(1, prof["statecol_fwd"]["launches"]), prof["statecol_bwd"]["ms"] / max(1, prof["statecol_bwd"]["launches"])), flush=True)
