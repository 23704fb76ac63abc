"""Run-to-run reproducibility soak of the state-column kernels (by hand on the GPU box: python tests/soak_reproducibility.py).
The split sweeps (round 5) moved barriers; the step sums go into the partial-sum buffer by adds without return.  A missing barrier
or an unordered addition would show as run-to-run differences: every repeat of a call must return the same bits.  d = 4096 with
B = 4096 columns (16 columns per workgroup, 256 workgroups) for the three table classes of bench.py, and the smaller shapes."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.engine import Engine

dev = torch.device("cuda:0")
ok = True
for name, n, B, S, build in (("ising_x", 12, 4096, 40, lambda: problems.random_ising_torch(12, 0, dev)),
                             ("xy_drives", 12, 4096, 40, lambda: problems.xy_drives_torch(12, 0, dev)),
                             ("qaoa_ring", 12, 4096, 40, lambda: problems.qaoa_ring_torch(12, dev)),
                             ("ising_x", 10, 1024, 200, lambda: problems.random_ising_torch(10, 0, dev)),
                             ("xy_drives", 9, 700, 300, lambda: problems.xy_drives_torch(9, 0, dev, 5, 4)),
                             ("ising_x", 8, 1024, 500, lambda: problems.random_ising_torch(8, 0, dev))):
    H0, Hs, M = build()
    K, d = Hs.shape[0], 1 << n
    dt = 1e-3
    u = torch.from_numpy(problems.pulse_table(K, S, S * dt, seed=1)).to(dev)
    psi0 = torch.from_numpy(problems.random_states(d, B, 2)).to(dev)
    with Engine(d, K) as e:
        e.use_torch_stream()
        e.set_hamiltonians(H0, Hs)
        e.set_observable(M)
        del H0, Hs, M
        ref = None
        for rep in range(6):
            loss, g, fin, en = e.energy_grad(u, dt, psi0)
            out = (np.float64(loss).tobytes(), np.asarray(g).tobytes(), fin.cpu().numpy().tobytes() if hasattr(fin, "cpu") else np.asarray(fin).tobytes())
            if ref is None:
                ref = out
            same = out == ref
            ok = ok and same
        info = (int(e.info("state_columns")), int(e.info("state_columns_quad_nodes")))
        fw = [e.propagate(u, dt, psi0) for _ in range(3)]
        fsame = all((np.asarray(x.cpu() if hasattr(x, "cpu") else x) == np.asarray(fw[0].cpu() if hasattr(fw[0], "cpu") else fw[0])).all() for x in fw)
        ok = ok and fsame
        print("%-10s n=%2d B=%5d S=%4d: state columns %d, quad nodes %d; 6 gradient calls bit-identical: %s; 3 forward calls: %s" % (
            name, n, B, S, info[0], info[1], same and ok, fsame), flush=True)
    torch.cuda.empty_cache()
print("REPRODUCIBLE" if ok else "DIFFERENCES FOUND")
sys.exit(0 if ok else 1)
