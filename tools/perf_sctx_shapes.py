"""Matrix-free engine at a given shape, DMA forms on / off: wall time, passes, checkpoint plan, kernel time by the engine's own events.
    python tools/perf_sctx_shapes.py n dtype B S [plan_steps]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from diffquantum_amd import problems
from diffquantum_amd.structured import StructuredEngine, ising_x_terms

n, dtype, B, S = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
plan = int(sys.argv[5]) if len(sys.argv) > 5 else 2000
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
st = rng.standard_normal((1 << n, B)) + 1j * rng.standard_normal((1 << n, B))
st = torch.from_numpy((st / np.linalg.norm(st, axis=0)).astype(np.complex64 if dtype == "c64" else np.complex128)).to(dev)
u = problems.pulse_table(n, S, S * 1e-3, seed=0)
for dma in (1, 0):
    with StructuredEngine(n, n, device=0, dtype=dtype) as e:
        e.use_torch_stream()
        e.set_terms(ising_x_terms(n, 0))
        e.set_option("dma", dma)
        e.set_option("plan_steps", plan)
        e.energy_grad(u[:2], 1e-3, st, want_final=False)
        e.set_option("profile", -1); e.set_option("profile", 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.energy_grad(u, 1e-3, st, want_final=False)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        print("n={} {} B={} S={} dma={}: {:8.2f} ms/step  passes/step {:.1f} (dma {:.1f})  degree {} stride {} two_level {} host {}  kernel ms/step {:.2f}  GB/s {:.0f}  pool GB {:.1f} pinned GB {:.1f}".format(
            n, dtype, B, S, dma, el / S * 1e3, e.info("passes") / S, e.info("dma_passes") / S, int(e.info("degree")), int(e.info("checkpoint_stride")),
            int(e.info("two_level")), int(e.info("host_checkpoints")), e.info("prof_ms") / S, e.info("prof_bytes") / max(e.info("prof_ms"), 1e-9) * 1e-6,
            e.info("checkpoint_pool_bytes") / 1e9, e.info("pinned_host_bytes") / 1e9), flush=True)
