"""The LDS-DMA passes of the matrix-free engine request their register operands by inline asm (csrc/structured.hip, S_GLOAD16 /
S_PIN): between the request and the vmcnt wait the destination registers must stay put, which holds as long as those kernels have no
VGPR spills.  This compiles the file for gfx950 with hipcc's resource remarks (no GPU needed) and checks exactly that."""
import os
import re
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "diffquantum_amd", "csrc")


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_dma_kernels_have_no_scratch(tmp_path):
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "--cuda-device-only", "-c", "structured.hip", "-o", str(tmp_path / "s.o"),
           "-Rpass-analysis=kernel-resource-usage"]
    res = subprocess.run(cmd, cwd=SRC, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    kernels, cur = {}, None
    for line in res.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            continue
        m = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", line)
        if m and cur:
            kernels[cur] = int(m.group(1))
    dma = {k: v for k, v in kernels.items() if "dma_kernel" in k}
    assert len(dma) >= 30, sorted(kernels)[:5]           # sapply_dma (flags x forms), schain, srchain, soverlap_dma
    assert all(v == 0 for v in dma.values()), {k: v for k, v in dma.items() if v}
