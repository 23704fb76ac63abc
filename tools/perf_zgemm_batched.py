"""Batched ZGEMM rate at small/mid dimensions (the launches of the time-parallel sweep and the composed forward)."""
import sys
import torch
sys.path.insert(0, ".")
from diffquantum_amd import _lib
from diffquantum_amd.engine import Engine

lib = _lib.load()
dev = torch.device("cuda:0")
for d, nb in ((64, 1000), (128, 1000), (256, 1000), (512, 256), (1024, 76), (2048, 8), (4096, 1)):
    A = torch.randn(nb, d, d, dtype=torch.complex128, device=dev)
    B = torch.randn(nb, d, d, dtype=torch.complex128, device=dev)
    Cm = torch.empty_like(A)
    with Engine(d, 0) as e:
        e.use_torch_stream()
        for tile in (0, 64, 128):
            if tile > d:
                continue
            e.set_option("force_tile", tile)
            for alg in (0, 1):
                e.set_option("zgemm_alg", alg)
                for shared in (0, 1):
                    for rep in range(2):
                        e.profile_enable(True); e.profile_reset()
                        for _ in range(3):
                            _lib.check(lib.dq_zgemm(e._ctx, 0, 0, d, d, d, 1.0, A.data_ptr(), d, 0 if shared else d * d,
                                                    B.data_ptr(), d, d * d, 0.0, Cm.data_ptr(), d, d * d, nb))
                        p = e.profile()["zgemm"]
                    print("d=%d batch=%d tile=%d alg=%s sharedA=%d: %.3f ms/launch, %.1f TF/s algorithmic" % (
                        d, nb, tile, "3M" if alg else "4M", shared, p["ms"] / p["launches"], p["flops"] / p["ms"] * 1e-9))
