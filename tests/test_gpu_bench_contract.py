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


def _run(args, env=None):
    e = dict(os.environ)
    e.update(env or {})
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True,
                         timeout=900, env=e)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]   # stdout carries the ONE JSON line and nothing else
    return json.loads(lines[0])


def _run_bench(*extra):
    return _run(["--qubits", "8", "--batch", "256", "--steps", "6", "--warmup", "1", "--skip-other-configs", *extra])


def _check_roofline(r):
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert 0.0 < r["frac"] <= 1.0, r["frac"]            # a FRACTION of the matrix pipe / of HBM bandwidth


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
    _check_roofline(d["roofline"])
    assert d["backend"].startswith("none")
    ee = d["end_to_end"]
    assert 0.0 <= ee["executed_mfma_frac_of_peak"] <= 1.0 and ee["executed_mfma_tflops"] >= 0.0
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1
    cs = c["c_single_thread"]                                   # the single-thread C restatement beside it
    assert cs["cores"] == 1 and cs["forward_steps_per_s_at_sample"] > 0 and "sample" in cs
    assert d["config"]["multi_gpu_form"] == "replicated" and d["config"]["multi_gpu_forms"] is None
    assert "input_structure" in d["config"]
    # the same chain for other input structures, each with a roofline of its own
    st = d["by_input_structure"]
    for name in ("ising_inputs_dense_real_path", "ising_inputs_complex_tape", "dense_real_symmetric_H", "dense_complex_hermitian_H"):
        assert st[name]["value"] > 0 and st[name]["chain_steps"] == 6, name
        _check_roofline(st[name]["roofline"])
    for name in ("ising_inputs_sparse_row_kernels", "ising_inputs_sparse_controls_dense_drift", "ising_inputs_dense_real_path",
                 "ising_inputs_complex_tape"):
        assert st[name]["max_abs_grad_diff_vs_headline"] < 1e-10, name
    assert st["ising_inputs_sparse_row_kernels"]["value"] > 0
    xy, xt = st["xy_drives_complex_pauli_H"], st["xy_drives_complex_tape"]   # complex Hermitian Pauli generator: column form vs tape
    assert xy["xor_regular"] == 2 and xy["value"] > 0 and xt["value"] > 0 and xt["max_abs_grad_diff_vs_column_form"] < 1e-10


def test_bench_general_path_only():
    d = _run_bench("--real-path", "0", "--skip-cpu-baseline")
    assert d["config"]["path"] == "general complex path" and d["by_input_structure"] is None
    assert d["roofline"]["kernel"].startswith("zgemm_kernel")
    _check_roofline(d["roofline"])


def test_bench_dense_workloads_as_the_measured_workload():
    for wl, path in (("dense-real", "real-symmetric fast path (H real)"), ("dense-complex", "general complex path")):
        d = _run_bench("--workload", wl, "--skip-cpu-baseline")
        assert d["config"]["path"] == path and d["value"] > 0 and d["by_input_structure"] is None
        _check_roofline(d["roofline"])


def test_bench_config2_and_config5_modes():
    d = _run(["--config", "2", "--steps", "200", "--batch", "128", "--warmup", "1"])
    assert d["n_gpus"] == 1 and d["steps"] == 200 and d["dtype"] == "f64" and d["scaling"] == "weak" and d["value"] > 0
    assert "configs[1]" in d["config"]["workload"] and d["max_norm_drift"] < 1e-12
    _check_roofline(d["roofline"])
    d = _run(["--config", "5", "--qubits", "14", "--batch", "16", "--steps", "4", "--warmup", "1"])
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["dtype"] == "f32" and d["scaling"] == "strong" and d["value"] > 0
    assert "configs[4]" in d["config"]["workload"] and d["roofline"]["bound"] == "hbm"
    _check_roofline(d["roofline"])


def test_bench_launches_its_own_ranks_and_runs_the_collectives_through_rccl():
    """`bench.py --gpus N` starts its ranks itself (torch.distributed.run as a child process, before any GPU call).  Here with
    N = 1 (`--launch`), backend nccl and DQ_FORCE_COLLECTIVES=1: the all-gather / all-reduce of the sliced form and the
    all-reduce of the replicated form run on device tensors through RCCL, both forms are checked against each other and
    timed, and the replicated rate agrees with a plain single-process run of the same chain."""
    common = ["--steps", "10", "--warmup", "2", "--skip-other-configs", "--skip-cpu-baseline", "--skip-general-path"]
    plain = _run(common)
    d = _run(["--gpus", "1", "--launch", *common], env={"DQ_FORCE_COLLECTIVES": "1"})
    assert d["n_gpus"] == 1 and d["backend"].startswith("nccl") and d["config"]["collectives_forced_at_world_1"] is True
    forms = d["config"]["multi_gpu_forms"]
    assert forms["sliced"]["value"] > 0 and forms["replicated"]["value"] > 0
    assert forms["max_abs_grad_diff_sliced_vs_replicated"] < 1e-10
    assert d["config"]["sliced_warmup_check"]["agreed_on_every_rank"] is True
    assert abs(d["config"]["loss"] - plain["config"]["loss"]) < 1e-9
    rep, ref = forms["replicated"]["value"], plain["value"]
    assert abs(rep - ref) < 0.10 * ref, (rep, ref)          # same chain, same GPU, seconds apart (measured: 1-2 %; the bound leaves room for a busy box)


def test_bench_propagator_sharded_form_through_rccl():
    """SURVEY 8-F2 as written (`--multi-gpu propagators`), ranks started by the launcher, N = 1 with forced collectives: the
    all-gather of the propagators and the reduce-scatter of their cotangents run on device tensors through RCCL; the warm-up
    chain's gradient equals the replicated form's."""
    d = _run(["--gpus", "1", "--launch", "--multi-gpu", "propagators", "--qubits", "10", "--batch", "256", "--steps", "5",
              "--warmup", "3", "--skip-other-configs", "--skip-cpu-baseline", "--skip-general-path"], env={"DQ_FORCE_COLLECTIVES": "1"})
    assert d["backend"].startswith("nccl") and d["config"]["multi_gpu_form"] == "propagators" and d["value"] > 0
    chk = d["config"]["propagators_warmup_check"]
    assert chk["max_abs_grad_diff_vs_replicated"] < 1e-10 and chk["abs_loss_diff_vs_replicated"] < 1e-9


def test_bench_refuses_more_gpus_than_the_node_has():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "2"], cwd=ROOT,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "GPU" in out.stderr


def test_graft_entry_smoke():
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("smoke ok") == 5
