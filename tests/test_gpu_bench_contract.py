"""The driver-facing entry points on a GPU box: `bench.py` prints ONE JSON line with the contract's fields (at a reduced
size so that the test takes seconds: n = 8, 256 states, 6 steps), `__graft_entry__.smoke()` passes. The bench's CPU
baseline leg is the oracle (the reference's scipy expm + matmul step, sim_plain.py:145-146) -- exercised here at n = 8."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_bench(*extra):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--qubits", "8", "--batch", "256", "--steps", "6",
                          "--warmup", "1", "--skip-other-configs", *extra], cwd=ROOT, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_json_contract():
    d = _run_bench()
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["value"] > 0 and abs(d["value"] * d["ms_per_step"] * 1e-3 - 1.0) < 1e-9       # steps/s x s/step (one GPU)
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    assert d["config"]["multi_gpu_form"] == "replicated" and d["config"]["other_multi_gpu_form"] is None


def test_bench_general_path_only():
    d = _run_bench("--real-path", "0", "--skip-cpu-baseline")
    assert d["config"]["path"] == "general complex path" and d["general_complex_path"] is None
    assert d["roofline"]["kernel"].startswith("zgemm_kernel")


def test_graft_entry_smoke():
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("smoke ok") == 3
