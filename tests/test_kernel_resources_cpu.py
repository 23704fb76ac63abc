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


def _regs(text):
    """VGPR numbers an instruction's text mentions (v7, v[4:7])."""
    out = set()
    for a, b in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(a), int(b) + 1))
    out.update(int(a) for a in re.findall(r"\bv(\d+)\b", text))
    return out


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_dma_kernels_leave_asm_load_destinations_alone_until_the_wait(tmp_path):
    """ScratchSize == 0 does not rule out a register copy (live-range split, coalescing) of an asm load's destination between the
    request and its vmcnt wait: such a copy reads stale data and needs no scratch.  Scan the gfx950 assembly of every *dma_kernel: from
    an inline-asm `global_load_dword[x4] vDST, ...` to the next `s_waitcnt vmcnt` no instruction may mention a DST register."""
    asm = tmp_path / "s.s"
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "--cuda-device-only", "-S", "structured.hip", "-o", str(asm)]
    res = subprocess.run(cmd, cwd=SRC, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-2000:]
    func, in_asm, pending, loads, bad, funcs = None, False, {}, 0, [], set()
    for ln, line in enumerate(asm.read_text().splitlines(), 1):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            func, pending = (m.group(1) if "dma_kernel" in m.group(1) else None), {}
            continue
        if func is None:
            continue
        text = line.split(";")[0].strip()
        if ";;#ASMSTART" in line:
            in_asm = True
            continue
        if ";;#ASMEND" in line:
            in_asm = False
            continue
        if not text or text.endswith(":") or text.startswith("."):
            continue
        if text.startswith("s_endpgm"):
            func = None
            continue
        if re.match(r"s_waitcnt\b.*vmcnt", text):
            pending = {}
            continue
        m = re.match(r"global_load_dword(?:x4)?\s+(v\[\d+:\d+\]|v\d+),", text)
        if in_asm and m:
            for r in _regs(m.group(1)):
                pending[r] = ln
            loads += 1
            funcs.add(func)
            rest = text[m.end():]
            hit = _regs(rest) & (set(pending) - _regs(m.group(1)))
        else:
            hit = _regs(text) & set(pending)
        if hit:
            bad.append((func[:60], ln, text, sorted(hit)))
    assert loads >= 100 and len(funcs) >= 20, (loads, len(funcs))       # the scan saw the asm loads of the DMA kernels
    assert not bad, bad[:5]


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not on PATH")
def test_partner_ring_slots_are_untouched_between_read_and_wait(tmp_path):
    """The partner ring of the state-column pairing sweeps (csrc/statecol_dev.h, xs_lds_ring): inline-asm ds_read_b128 into a slot, a
    counted s_waitcnt lgkmcnt(N) tied to the slot before its use.  LDS returns in order, so after a wait with count N all but the N
    youngest reads have landed: scan the gfx950 assembly of the d = 4096 adjoint kernels and require that no instruction mentions a
    slot's registers while its read is still in flight (a compiler-made copy there would read stale data)."""
    bad, reads, funcs = [], 0, set()
    for src in ("statecol_bwdq.hip", "statecol_bwd.hip"):
        asm = tmp_path / (src + ".s")
        cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "--cuda-device-only", "-DXS_PROBE_4096_ONLY", "-S", src, "-o", str(asm)]
        res = subprocess.run(cmd, cwd=SRC, capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-2000:]
        func, in_asm, fifo = None, False, []
        for ln, line in enumerate(asm.read_text().splitlines(), 1):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                func, fifo = (m.group(1) if "xs_backward" in m.group(1) else None), []
                continue
            if func is None:
                continue
            if ";;#ASMSTART" in line:
                in_asm = True
                continue
            if ";;#ASMEND" in line:
                in_asm = False
                continue
            text = line.split(";")[0].strip()
            if not text or text.endswith(":") or text.startswith("."):
                continue
            if text.startswith("s_endpgm"):
                func = None
                continue
            if text.startswith("s_branch"):
                # an unconditional branch: the next line of the text is not its successor (a block reached from elsewhere, e.g. the
                # prologue of the next class's ring, entered with every read of the previous ring drained) -- the scan is per
                # straight-line path, so the in-flight list does not carry over
                fifo = []
                continue
            m = re.match(r"s_waitcnt\b.*lgkmcnt\((\d+)\)", text)
            if m:
                fifo = fifo[len(fifo) - int(m.group(1)):] if int(m.group(1)) < len(fifo) else fifo
                continue
            m = re.match(r"ds_read_b128\s+(v\[\d+:\d+\]),", text)
            if in_asm and m:
                dst = _regs(m.group(1))
                hit = _regs(text[m.end():]) & set().union(*fifo) if fifo else set()
                fifo.append(dst)
                reads += 1
                funcs.add(func)
            else:
                hit = _regs(text) & set().union(*fifo) if fifo else set()
            if hit:
                bad.append((func[:70], ln, text, sorted(hit)))
    assert reads >= 100 and len(funcs) >= 6, (reads, len(funcs))
    assert not bad, bad[:5]
