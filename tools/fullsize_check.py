"""Full-size run of BASELINE configs[2]: n=12 (d=4096), K=12, B=4096, S=2000 Trotter steps, fwd + adjoint.
Reports wall time / steps/s, checkpoint stride, memory, and size-independent property checks:
norm conservation of psi(T), energy consistency, and a directional-derivative check of the gradient on a
shorter chain (two extra forward runs)."""
import json, sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine

n, B = 12, 4096
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
S_fd = int(sys.argv[2]) if len(sys.argv) > 2 else 100
d = 1 << n
dev = torch.device("cuda:0")
H0, Hs, M = problems.random_ising_torch(n, 0, dev)
psi0 = torch.from_numpy(problems.random_states(d, B, 0)).to(dev)
e = Engine(d, n); e.use_torch_stream(); e.set_hamiltonians(H0, Hs); e.set_observable(M)
res = {}
# warm-up (allocations)
e.energy_grad(torch.from_numpy(problems.pulse_table(n, 2, 2e-3, 0)).to(dev), 1e-3, psi0, want_final=False, want_energies=False)
u = problems.pulse_table(n, S, 1e-3 * S, seed=0)
tu = torch.from_numpy(u).to(dev)
torch.cuda.synchronize(); t0 = time.time()
loss, grad, fin, en = e.energy_grad(tu, 1e-3, psi0)
torch.cuda.synchronize(); t1 = time.time()
free, tot = torch.cuda.mem_get_info()
nrm = torch.linalg.norm(fin, dim=0)
res.update(S=S, wall_s=t1 - t0, steps_per_s=S / (t1 - t0), checkpoint_stride=e.info("checkpoint_stride"),
           degree=e.info("degree"), theta=e.info("theta"), hbm_used_GB=(tot - free) / 1e9,
           loss=loss, energies_sum=float(en.sum()), max_norm_drift=float((nrm - 1).abs().max().item()),
           grad_norm=float(np.linalg.norm(grad)), grad_max=float(np.abs(grad).max()),
           grad_finite=bool(np.isfinite(grad).all()))
print(json.dumps(res)); sys.stdout.flush()
# directional derivative on a shorter chain at full d, B
u2 = problems.pulse_table(n, S_fd, 1e-3 * S_fd, seed=1)
v = np.random.default_rng(0).standard_normal(u2.shape)
l0, g0, _, _ = e.energy_grad(u2, 1e-3, psi0, want_final=False, want_energies=False)
eps = 1e-4
lp = e.energy(e.propagate(torch.from_numpy(u2 + eps * v).to(dev), 1e-3, psi0))[1]
lm = e.energy(e.propagate(torch.from_numpy(u2 - eps * v).to(dev), 1e-3, psi0))[1]
fd = (lp - lm) / (2 * eps)
an = float(np.sum(g0 * v))
res2 = dict(S_fd=S_fd, directional_fd=fd, directional_adjoint=an, abs_diff=abs(fd - an), rel_diff=abs(fd - an) / max(abs(an), 1e-30))
print(json.dumps(res2))
