"""The driver-facing entry points on a GPU box: `bench.py` prints ONE JSON line with the contract's fields (at a reduced
size so that the test takes seconds: n = 8, 256 states, 6 steps), `__graft_entry__.smoke()` passes. The bench's CPU
baseline leg is the oracle (the reference's scipy expm + matmul step, sim_plain.py:145-146) -- exercised here at n = 8."""
import json
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LINE_LIMIT = 4096   # the driver keeps an 8000-character stdout tail: the ONE line must fit with room to spare


def _run(args, env=None, extras=True):
    """Run bench.py; returns (line, full): the parsed ONE line of stdout and the extras record it points at."""
    e = dict(os.environ)
    e.update(env or {})
    with tempfile.TemporaryDirectory() as tmp:
        ex = ["--extras", os.path.join(tmp, "extras.json")] if extras else []
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, *ex], cwd=ROOT, capture_output=True, text=True,
                             timeout=1500, env=e)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]   # stdout carries the ONE JSON line and nothing else
        assert len(lines[0]) < LINE_LIMIT, len(lines[0])
        line = json.loads(lines[0])
        path = line["extras"] if os.path.isabs(line["extras"]) else os.path.join(ROOT, line["extras"])
        with open(path) as f:
            full = json.load(f)
    return line, full


def _run_bench(*extra):
    return _run(["--qubits", "8", "--batch", "256", "--steps", "6", "--warmup", "1", "--skip-other-configs", *extra])


def _check_roofline(r, rel=1e-12):
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma", "lds") and r["peak"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < rel * max(1.0, r["frac"])
    assert 0.0 < r["frac"] <= 1.0, r["frac"]            # a FRACTION of the matrix pipe / of HBM bandwidth / of the LDS array
    assert len(r["kernel"]) <= 80


def _check_line(d):
    """The ONE line: the contract's fields in their short form."""
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "backend", "extras", "trials"):
        assert key in d, key
    assert "workload" in d["config"] and "model" not in d["config"]
    assert all(not isinstance(v, (dict, list)) for v in d["config"].values())      # sizes and short strings only
    _check_roofline(d["roofline"], rel=1e-5)                                         # (six significant digits in the line)
    t = d["trials"]
    assert t["n"] >= 3 and t["ms_per_step_min"] <= t["ms_per_step_median"] <= t["ms_per_step_max"]
    assert abs(t["ms_per_step_median"] - d["ms_per_step"]) < 1e-5 * d["ms_per_step"]


def test_bench_json_contract():
    line, d = _run_bench()
    _check_line(line)
    assert line["n_gpus"] == 1 and line["steps"] == 6 and line["warmup"] == 1
    assert line["higher_is_better"] is True and line["scaling"] == "weak" and line["vs_baseline"] is None
    assert line["dtype"] == "f64" and line["data"] == "synthetic"
    assert line["value"] > 0 and abs(line["value"] * line["ms_per_step"] * 1e-3 - 1.0) < 1e-4       # steps/s x s/step (one GPU)
    assert line["backend"].startswith("none")
    assert 0.0 <= line["end_to_end"]["executed_mfma_frac_of_peak"] <= 1.0
    c = line["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("port", "reference") and c["value"] > 0 and c["cores"] >= 1 and len(c["sample"]) <= 160
    cs = c["c_single_thread"]                                   # the single-thread C restatement beside it
    assert cs["cores"] == 1 and cs["forward_steps_per_s_at_sample"] > 0 and cs["extrapolated_forward_steps_per_s"] > 0
    assert line["config"]["multi_gpu_form"] == "replicated"
    for name, (rate, frac) in line["by_input_structure"].items():                    # {name: [steps_per_s, roofline_frac]}
        assert rate > 0 and 0.0 < frac <= 1.0, name
    # ---- the extras record: everything else -------------------------------------------------------------------
    assert d["value"] == pytest.approx(line["value"], rel=1e-5) and d["config"]["multi_gpu_forms"] is None
    assert "input_structure" in d["config"] and "sample" in d["cpu_baseline"]["c_single_thread"]
    ee = d["end_to_end"]
    assert 0.0 <= ee["executed_mfma_frac_of_peak"] <= 1.0 and ee["executed_mfma_tflops"] >= 0.0
    _check_roofline(d["roofline"])
    # the same chain for other input structures, each with a roofline of its own
    st = d["by_input_structure"]
    assert set(st) == set(line["by_input_structure"])
    for name in ("ising_inputs_dense_real_path", "ising_inputs_complex_tape", "dense_real_symmetric_H", "dense_complex_hermitian_H"):
        assert st[name]["value"] > 0 and st[name]["chain_steps"] == 6, name
        _check_roofline(st[name]["roofline"])
    for name in ("ising_inputs_sparse_row_kernels", "ising_inputs_sparse_controls_dense_drift", "ising_inputs_dense_real_path",
                 "ising_inputs_complex_tape"):
        assert st[name]["max_abs_grad_diff_vs_headline"] < 1e-10, name
    assert st["ising_inputs_sparse_row_kernels"]["value"] > 0
    xy, xt = st["xy_drives_complex_pauli_H"], st["xy_drives_complex_tape"]   # complex Hermitian Pauli generator: column form vs tape
    assert xy["xor_regular"] == 2 and xy["value"] > 0 and xt["value"] > 0 and xt["max_abs_grad_diff_vs_column_form"] < 1e-10


def test_bench_line_fits_the_driver_window_as_the_driver_runs_it():
    """`python3 bench.py --gpus 1 --steps 20 --warmup 5` exactly as the driver runs it (default extras path; the flag-less
    default differs by the warm-up count only): ONE line under 4 KB with roofline and cpu_baseline in it (BENCH_r02 lost its 31.5 KB line)."""
    for args in (["--gpus", "1", "--steps", "20", "--warmup", "5"],):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], cwd=ROOT, capture_output=True, text=True, timeout=1500)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1 and len(lines[0]) < LINE_LIMIT, (len(lines), len(lines[0]))
        d = json.loads(lines[0])
        _check_line(d)
        assert d["steps"] == 20 and d["cpu_baseline"]["value"] > 0 and d["roofline"]["frac"] > 0
        assert d["extras"] == "bench_extras.json" and os.path.exists(os.path.join(ROOT, d["extras"]))
        # round 4: the north-star kernel and the full chain are in the driver-parsed line
        m = d["roofline_mfma"]
        assert m["kernel"].startswith("zgemm_lin_kernel") and m["peak"] == 78.6 and 0.5 < m["frac"] < 1.0 and m["launches"] > 0
        assert abs(m["frac"] - m["achieved"] / m["peak"]) < 1e-5   # (both printed with six significant digits)
        # BASELINE configs[2] at FULL size under -m gpu: the S = 2000, d = B = 4096 gradient of the state-column kernels
        # against the checkpointed explicit-propagator GEMM chain on the same inputs (benchlib/measure.py full_chain)
        fc = d["full_chain"]
        assert fc[0] == 2000 and fc[1] > 0 and fc[2] is not None and fc[2] < 1e-10      # max |dL/du - the checkpointed GEMM chain's|
        # round 5: the reference's own control sets run the quadrature adjoint (4-point rule: 2 interior nodes) at the headline's size
        ex = json.load(open(os.path.join(ROOT, d["extras"])))
        st = ex["by_input_structure"]
        assert st["xy_drives_state_columns"]["state_columns_quad_nodes"] == 2 and st["xy_drives_state_columns"]["state_columns"] == 1
        assert st["qaoa_zz_x_n12"]["state_columns_quad_nodes"] == 2 and st["qaoa_zz_x_n12"]["state_columns"] == 1
        assert "qaoa_zz_x_n12" in d["by_input_structure"] and d["by_input_structure"]["qaoa_zz_x_n12"][0] > 0


def test_bench_general_path_only():
    line, d = _run_bench("--real-path", "0", "--skip-cpu-baseline")
    assert d["config"]["path"] == "general complex path" and d["by_input_structure"] is None and "by_input_structure" not in line
    assert d["roofline"]["kernel"].startswith("zgemm_kernel") and line["roofline"]["kernel"].startswith("zgemm_kernel")
    _check_roofline(d["roofline"])
    _check_line(line)


def test_bench_dense_workloads_as_the_measured_workload():
    for wl, path in (("dense-real", "real-symmetric fast path (H real)"), ("dense-complex", "general complex path")):
        line, d = _run_bench("--workload", wl, "--skip-cpu-baseline")
        assert d["config"]["path"] == path and d["value"] > 0 and d["by_input_structure"] is None
        _check_roofline(d["roofline"])
        _check_line(line)


def test_bench_config2_and_config5_modes():
    line, d = _run(["--config", "2", "--steps", "200", "--batch", "128", "--warmup", "1"])
    _check_line(line)
    assert d["n_gpus"] == 1 and d["steps"] == 200 and d["dtype"] == "f64" and d["scaling"] == "weak" and d["value"] > 0
    assert "configs[1]" in d["config"]["workload"] and d["max_norm_drift"] < 1e-12
    _check_roofline(d["roofline"])
    line, d = _run(["--config", "5", "--qubits", "14", "--batch", "16", "--steps", "4", "--warmup", "1"])
    _check_line(line)
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["dtype"] == "f32" and d["scaling"] == "strong" and d["value"] > 0
    assert "configs[4]" in d["config"]["workload"] and d["roofline"]["bound"] == "hbm"
    _check_roofline(d["roofline"])


def test_bench_launches_its_own_ranks_and_runs_the_collectives_through_rccl():
    """`bench.py --gpus N` starts its ranks itself (torch.distributed.run as a child process, before any GPU call).  Here with
    N = 1 (`--launch`), backend nccl and DQ_FORCE_COLLECTIVES=1.  The timed form is north_star's partitioning (batch shards +
    ONE all-reduce of loss and dL/du, 'replicated' -- the default); the all-gather / all-reduce of the sliced form run on device
    tensors through RCCL in the warm-up check and in the `beside` measurement; the rate agrees with a plain single-process run."""
    common = ["--steps", "10", "--warmup", "2", "--skip-other-configs", "--skip-cpu-baseline", "--skip-general-path"]
    pline, plain = _run(common)
    line, d = _run(["--gpus", "1", "--launch", *common], env={"DQ_FORCE_COLLECTIVES": "1"})
    _check_line(line)
    assert line["n_gpus"] == 1 and line["backend"].startswith("nccl") and line["config"]["collectives_forced_at_world_1"] is True
    assert line["config"]["multi_gpu_form"] == "replicated" and line["config"]["parallelism"] == "dp1"
    bs = line["beside"]                                   # [aggregate steps/s, ms per chain step] of the forms NOT timed
    assert bs["sliced_form"][0] > 0 and bs["one_gpu_whole_global_batch"][0] > 0
    assert d["config"]["sliced_warmup_check"]["agreed_on_every_rank"] is True
    assert d["config"]["sliced_warmup_check"]["max_abs_grad_diff_vs_replicated"] < 1e-10
    assert abs(d["config"]["loss"] - plain["config"]["loss"]) < 1e-9
    rep, ref = line["value"], pline["value"]
    assert abs(rep - ref) < 0.10 * ref, (rep, ref)          # same chain, same GPU, seconds apart (measured: 1-2 %; the bound leaves room for a busy box)
    # at world 1 the whole global batch IS the shard: the beside figure must agree with the headline too
    assert abs(bs["one_gpu_whole_global_batch"][0] - rep) < 0.15 * rep
    # 'auto' keeps the old behaviour: both forms checked, tried once, the faster one timed
    line, d = _run(["--gpus", "1", "--launch", "--multi-gpu", "auto", *common], env={"DQ_FORCE_COLLECTIVES": "1"})
    forms = d["config"]["multi_gpu_forms"]
    assert forms["sliced"]["value"] > 0 and forms["replicated"]["value"] > 0
    assert forms["max_abs_grad_diff_sliced_vs_replicated"] < 1e-10


def test_bench_propagator_sharded_form_through_rccl():
    """SURVEY 8-F2 as written (`--multi-gpu propagators`), ranks started by the launcher, N = 1 with forced collectives: the
    all-gather of the propagators and the reduce-scatter of their cotangents run on device tensors through RCCL; the warm-up
    chain's gradient equals the replicated form's."""
    line, d = _run(["--gpus", "1", "--launch", "--multi-gpu", "propagators", "--qubits", "10", "--batch", "256", "--steps", "5",
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
    assert out.stdout.count("smoke ok") == 8 and "quadrature adjoint" in out.stdout and "routed to the matrix-free passes" in out.stdout
