"""A/B on one box: matrix-free engine options (n = 18, B = 512, c64, 20 steps under the 4000-step plan)."""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.structured import StructuredEngine, ising_x_terms
n, B, S, dt = 18, 512, 20, 1e-3
d = 1 << n
dev = torch.device("cuda:0")
e = StructuredEngine(n, n, dtype="c64")
e.use_torch_stream()
e.set_terms(ising_x_terms(n, 0))
e.set_option("plan_steps", 4000)
rng = np.random.default_rng(0)
tp = torch.randn(d, B, dtype=torch.complex64, device=dev)
tp /= torch.linalg.norm(tp, dim=0)
u = problems.pulse_table(n, S, S * dt, seed=0)
res = {}
for rep in range(3):
    for name, opts in [("base", {}), ("keep_last", {"reverse_keep_last": 1}), ("no_handover", {"step_handover": 0})]:
        for k in ("reverse_keep_last", "step_handover"):
            e.set_option(k, {"reverse_keep_last": 0, "step_handover": 1}[k])
        for k, v in opts.items():
            e.set_option(k, v)
        if rep == 0:
            e.energy_grad(u, dt, tp, want_final=False)
        torch.cuda.synchronize(); t0 = time.time()
        loss, grad, _, _ = e.energy_grad(u, dt, tp, want_final=False)
        torch.cuda.synchronize(); t1 = time.time()
        res.setdefault(name, []).append((t1 - t0) / S * 1e3)
        print(name, "%.3f ms/step" % ((t1 - t0) / S * 1e3), "loss %.9f |g| %.6e passes %d" % (loss, np.linalg.norm(grad), e.info("passes")), flush=True)
for k, v in res.items():
    print(k, "median %.3f ms/step = %.2f steps/s" % (sorted(v)[1], 1e3 / sorted(v)[1]))
