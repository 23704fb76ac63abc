"""BASELINE configs[4] end to end on ONE GPU: the whole chain (default 4000 steps) at n = 18, batch B, complex64, forward + adjoint in ONE
dq_sctx_energy_grad call -- real checkpoint stride, real reversals.   python tools/config5_full_chain.py [S] [B] [n]
Checks: loss finite, the final states' norms, a directional derivative of the loss along a random pulse direction against central
differences of two more forward chains (propagate + energy)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffquantum_amd import problems  # noqa: E402
from diffquantum_amd.structured import StructuredEngine, ising_x_terms  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
n = int(sys.argv[3]) if len(sys.argv) > 3 else 18
EPS_RANDOM = [float(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0.25, 0.5, 1.0]
EPS_GRAD = [float(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0.05, 0.1]
dt = 1e-3
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
st = rng.standard_normal((1 << n, B)) + 1j * rng.standard_normal((1 << n, B))
st = torch.from_numpy((st / np.linalg.norm(st, axis=0)).astype(np.complex64)).to(dev)
u = problems.pulse_table(n, S, S * dt, seed=0)
with StructuredEngine(n, n, device=0, dtype="c64") as e:
    e.use_torch_stream()
    e.set_terms(ising_x_terms(n, 0))
    e.energy_grad(u[:3], dt, st, want_final=False)   # warm-up
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = e.propagate(u, dt, st)
    torch.cuda.synchronize(); t_f = time.perf_counter() - t0
    drift = float((torch.linalg.norm(out.to(torch.complex128), dim=0) - 1).abs().max())
    del out
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, grad, _, _ = e.energy_grad(u, dt, st, want_final=False)
    torch.cuda.synchronize(); t_g = time.perf_counter() - t0
    stride, passes, deg = int(e.info("checkpoint_stride")), int(e.info("passes")), int(e.info("degree"))
    print("n = {}  B = {}  S = {}  degree {}  checkpoint stride {}  passes {} ({:.2f} per step)".format(n, B, S, deg, stride, passes, passes / S))
    print("forward only : {:8.2f} s   {:8.2f} steps/s   max | |psi_b(T)| - 1 | = {:.2e}".format(t_f, S / t_f, drift))
    print("fwd + adjoint: {:8.2f} s   {:8.2f} steps/s   loss = {:.9f}   |grad| = {:.4e}".format(t_g, S / t_g, loss, float(np.linalg.norm(grad))))
    print("two checkpoint levels: {}   kept states in pinned host memory: {:.1f} GB   device checkpoint pools: {:.1f} GB".format(
        int(e.info("two_level")), e.info("pinned_host_bytes") / 1e9, e.info("checkpoint_pool_bytes") / 1e9))

    def L(uu):
        f = e.propagate(uu, dt, st)
        return float(torch.real(torch.sum(torch.conj(f).to(torch.complex128) * e.apply(f, observable=True).to(torch.complex128))))

    # (1) a random pulse direction: the derivative along it is ~1e-3 of |grad| -- the loss difference it makes is of the size of the fp32
    # noise of two 4000-step forward chains, whatever the step; (2) the direction of the gradient itself: a derivative of |grad|
    dirs = [("random direction", rng.standard_normal(u.shape), EPS_RANDOM), ("gradient direction", np.array(grad, dtype=np.float64), EPS_GRAD)]
    for name, v, steps in dirs:
        v = v / np.linalg.norm(v)
        an = float(np.sum(grad * v))
        for eps in steps:
            fd = (L(u + eps * v) - L(u - eps * v)) / (2 * eps)
            print("directional derivative, {}: adjoint {:.6e}   central difference (eps = {}) {:.6e}   relative difference {:.2e}".format(
                name, an, eps, fd, abs(an - fd) / max(abs(fd), 1e-30)), flush=True)
