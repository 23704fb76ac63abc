"""bench.py's roofline arithmetic on a synthetic profile (no GPU): `frac` is a fraction of EXECUTED work -- 6 M N K per
launch for the 3M complex product, the triangle's tiles only for the Hermitian-output form -- the algorithmic 8 M N K rate is
kept beside it, classes are ranked by their share of the timed region, and the HBM class is bytes / time / 8 TB/s."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def _prof(**kw):
    z = {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0}
    p = {k: dict(z) for k in ("zgemm", "assemble", "lincomb", "gradc", "other", "rgemm", "rsym", "zherm", "spmm")}
    p.update(kw)
    return p


def test_fractions_at_the_headline_shape():
    d = 4096
    W = 8.0 * d ** 3
    prof = _prof(zgemm={"launches": 42, "ms": 42 * 5.83, "flops": 42 * W, "bytes": 42 * 805306368.0},
                 zherm={"launches": 20, "ms": 20 * 3.01, "flops": 20 * W, "bytes": 20 * 805306368.0},
                 spmm={"launches": 360, "ms": 360 * 0.146, "flops": 0.0, "bytes": 360 * 343e6},
                 lincomb={"launches": 160, "ms": 16.0, "flops": 0.0, "bytes": 1e9})
    b = bench.roofline_blocks(prof, {"zgemm_alg": 2.0}, d)
    z, h, s = b["zgemm"], b["zherm"], b["spmm"]
    assert abs(z["achieved_algorithmic"] - W / 5.83e-3 / 1e12) < 1e-9
    assert abs(z["achieved"] - 0.75 * z["achieved_algorithmic"]) < 1e-9          # 3M: 6 of every 8 algorithmic flops executed
    assert abs(z["frac"] - z["achieved"] / 78.6) < 1e-12 and 0.88 < z["frac"] < 0.92
    assert abs(h["executed_over_algorithmic_flops"] - 0.75 * 528 / 1024) < 1e-12   # triangle: 528 of 1024 tiles
    assert 0.88 < h["frac"] < 0.92
    assert s["bound"] == "hbm" and abs(s["frac"] - (343e6 / 0.146e-3 / 1e9) / 8000.0) < 1e-9
    assert abs(z["traffic_over_algorithmic"] - bench.TRAFFIC["zgemm"][0] / 805306368.0) < 1e-9
    assert z["traffic_source"].startswith("profiles/") and os.path.exists(os.path.join(ROOT, z["traffic_source"]))
    rk = bench.ranked(b)
    assert [r is x for r, x in zip(rk[:3], (z, h, s))] == [True, True, True]
    total = sum(v["ms"] for v in prof.values())
    assert abs(z["share_of_kernel_time"] - 42 * 5.83 / total) < 1e-12
    ex = bench.executed_tflops(b, 0.381)
    assert abs(ex - (z["achieved"] * 42 * 5.83e-3 + h["achieved"] * 20 * 3.01e-3) / 0.381) < 1e-9
    for blk in b.values():
        assert 0.0 <= blk["frac"] <= 1.0


def test_small_dimensions_the_3m_count_from_the_128_tile_up_and_no_traffic_constant():
    """d = 256 (configs[1]): the batched products run the 3M form on the 128 tile -- executed = 0.75 of the algorithmic count, so the
    fraction never counts flops that were not executed; below d = 128 everything is 4M.  PMC traffic constants hold at d = 4096 only."""
    for d, factor in ((256, 0.75), (64, 1.0)):
        W = 8.0 * d ** 3
        b = bench.roofline_blocks(_prof(zgemm={"launches": 10, "ms": 1.0, "flops": 10 * W, "bytes": 1.0}), {"zgemm_alg": 2.0}, d)
        z = b["zgemm"]
        assert z["executed_over_algorithmic_flops"] == factor and z["traffic"] is None and z["traffic_source"] is None
        assert abs(z["achieved"] - factor * z["achieved_algorithmic"]) < 1e-12
        assert bench.ranked(b) == [z]


def test_column_form_class_is_priced_against_the_lds_array():
    """XOR-regular generator: the class holds the column-form kernels; achieved = LDS bytes gathered / time against 256 B/clk/CU."""
    d, R, q = 4096, 12, 3
    nn = float(d) * d
    fwd = 2.0 * (R + 1) * nn * 2 * q            # flops booked by the engine per forward launch (diagonal included)
    rev = 2.0 * (R + 1) * nn * (4 * q + 1)
    prof = _prof(spmm={"launches": 60, "ms": 40 * 0.2 + 20 * 0.83, "flops": 40 * fwd + 20 * rev, "bytes": 60 * 16 * nn})
    b = bench.roofline_blocks(prof, {"zgemm_alg": 2.0, "xor_regular": 1.0, "xor_regular_found": float(R)}, d)
    s = b["spmm"]
    lds = (40 * fwd + 20 * rev) * 4.0 * R / (R + 1.0)
    assert s["bound"] == "lds" and s["peak"] == 256 * 256 * 2.4
    assert abs(s["achieved"] - lds / ((40 * 0.2 + 20 * 0.83) * 1e-3) / 1e9) < 1e-6
    assert 0.0 < s["frac"] < 1.0 and s["kernel"].startswith("xr_forward_kernel") and len(s["kernel"]) <= 80
    assert abs(s["hbm_gbps_algorithmic"] - 60 * 16 * nn / ((40 * 0.2 + 20 * 0.83) * 1e-3) / 1e9) < 1e-6
    assert bench.executed_tflops(b, 1.0) == 0.0          # (no MFMA class in this synthetic profile)


def test_the_one_line_stays_under_4_kb_whatever_the_full_record_holds():
    """the driver keeps an 8000-character stdout tail; BENCH_r02's 31.5 KB line was lost.  The line is a
    short form of the full record (which goes to the extras file): built here from round 2's committed 31.5 KB record."""
    import json
    with open(os.path.join(ROOT, "profiles", "r2_bench_final.json")) as f:
        full = json.load(f)
    assert len(json.dumps(full)) > 30000
    full["extras"] = "bench_extras.json"
    full["trials"] = {"n": 3, "ms_per_step_median": 16.1, "ms_per_step_min": 16.0, "ms_per_step_max": 16.3}
    full["beside"] = {"sliced_form": [4115.123456789, 1.9], "one_gpu_whole_global_batch": [70.1, 114.2]}
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) < 4096
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "extras"):
        assert key in line, key
    r = line["roofline"]
    assert r["bound"] == "mfma" and len(r["kernel"]) <= 80 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-5
    assert line["cpu_baseline"]["cores"] == 128 and line["cpu_baseline"]["c_single_thread"]["cores"] == 1
    assert line["by_input_structure"]["dense_complex_hermitian_H"] == [12.78, 0.91]
    assert all(not isinstance(v, (dict, list)) for v in line["config"].values())
    # a record that is too big even in short form loses its optional objects, never the contract's fields
    full["by_input_structure"] = {"case_{:04d}".format(i): {"value": 1.0, "roofline": {"frac": 0.5}} for i in range(400)}
    line = bench.compact_line(full)
    assert len(json.dumps(line)) <= bench.LINE_LIMIT and "by_input_structure" not in line and "roofline" in line and "cpu_baseline" in line


def test_state_column_classes_are_priced_against_the_lds_array_with_the_scratch_stream_as_traffic():
    """statecol.hip's classes: `bytes` = LDS bytes gathered (booked by the library), `flops` = fp64 vector work; the adjoint
    launch's PMC traffic is its scratch stream per step x the chain's steps, far above the states in and out."""
    d, B, K, R, L, lds_diag = 4096, 4096, 20, 12, 7, 9
    rows = float(d) * B
    fwd_lds = rows * L * K * lds_diag * 16.0
    bwd_lds = rows * K * ((2 * L - 1) * lds_diag + L * R) * 16.0
    prof = _prof()
    prof["statecol_fwd"] = {"launches": 3, "ms": 3 * 5.25, "flops": 3 * rows * L * K * (4.0 * R + 8.0), "bytes": 3 * fwd_lds}
    prof["statecol_bwd"] = {"launches": 3, "ms": 3 * 23.9, "flops": 3 * rows * K * ((3 * L - 1) * (4.0 * R + 8.0) + L * 4.0 * R), "bytes": 3 * bwd_lds}
    b = bench.roofline_blocks(prof, {"zgemm_alg": 2.0, "batch": B, "chain_steps": K}, d)
    f, w = b["statecol_fwd"], b["statecol_bwd"]
    assert f["bound"] == "lds" and w["bound"] == "lds" and len(w["kernel"]) <= 80 and len(f["kernel"]) <= 80
    assert abs(f["achieved"] - fwd_lds / 5.25e-3 / 1e9) < 1e-6 and 0.3 < f["frac"] < 0.5
    assert 0.15 < w["frac"] < 0.4 and 0.0 < w["vector_f64_frac_of_peak"] < 1.0
    assert abs(f["traffic_over_algorithmic"] - 1.0) < 0.01                       # states in and out, nothing else
    assert abs(w["traffic"] - bench.TRAFFIC["statecol_bwd_per_step"][0] * K) < 1.0 and w["traffic_over_algorithmic"] > 50   # the scratch stream of forward terms
    assert os.path.exists(os.path.join(ROOT, w["traffic_source"]))
    assert bench.ranked(b)[0] is w and bench.ranked(b)[1] is f
    other = bench.roofline_blocks(prof, {"zgemm_alg": 2.0, "batch": 512, "chain_steps": K}, 1024)
    assert other["statecol_bwd"]["traffic"] is None                              # PMC constants hold for the headline shape only


def test_the_eight_gpu_line_parses_and_fits():
    """The N = 8 line (configs[3]: B = 32768 over 8 GPUs) as rank 0 would print it: built from the record of the 4-rank gloo rehearsal on
    one GPU (profiles/r4_bench_gloo4_rehearsal.json -- more than 6 processes may not share the card of a gpurun box, so 8 ranks were never
    started there) with the world size set to 8: one line < 4 KB that carries n_gpus, backend, the global batch and, beside the timed
    form, the whole global batch on one GPU."""
    rec = json.load(open(os.path.join(ROOT, "profiles", "r4_bench_gloo4_rehearsal.json")))
    assert rec["n_gpus"] == 4 and rec["config"]["global_batch"] == 4 * 4096 and rec["backend"].startswith("gloo")
    assert set(rec["beside"]) == {"sliced_form", "one_gpu_whole_global_batch"} and rec["config"]["parallelism"] == "dp4"
    full = dict(rec)
    full.update(n_gpus=8, backend="nccl (RCCL over xGMI)", value=8 * 20 / 0.0212)
    full["config"] = dict(rec["config"], global_batch=8 * 4096, parallelism="dp8", parallelism_detail="dp8 (batch shards, one all-reduce of loss+dL/du)",
                          workload=rec["config"]["workload"])
    full["beside"] = {"sliced_form": [120.0, 66.0], "one_gpu_whole_global_batch": [8 * 20 / 0.1604, 8.02]}
    line = bench.compact_line(full)
    text = json.dumps(line)
    assert len(text) <= bench.LINE_LIMIT
    back = json.loads(text)
    assert back["n_gpus"] == 8 and back["scaling"] == "weak" and back["config"]["global_batch"] == 32768
    assert back["backend"].startswith("nccl") and back["beside"]["one_gpu_whole_global_batch"][1] == 8.02
    for k in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline"):
        assert k in back, k
