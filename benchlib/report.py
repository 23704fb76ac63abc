"""bench.py, reporting half: peaks and PMC traffic constants, the roofline blocks made from the library's per-launch HIP events,
and the ONE JSON line (short form < 4 KB) + the extras file."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PEAK_F64_MFMA_TFLOPS = 78.6  # v_mfma_f64_16x16x4_f64: 64 cycles/instr/SIMD measured (tools/mfma_f64_probe.hip)
                             # = 32 flop/clk/SIMD * 1024 SIMDs * 2.4 GHz; equals the datasheet FP64-matrix figure
PEAK_HBM_GBPS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)
PEAK_LDS_GBPS = 256 * 256 * 2.4  # same guide, LDS table: ds_read_b64 = 2 LDS cycles per wave = 256 B/clk/CU x 256 CUs x 2.4 GHz
# HBM-side bytes per launch at d = B = 4096, fp64: rocprofv3 PMC, separate --pmc FETCH_SIZE / WRITE_SIZE passes,
# read = 2 * FETCH_SIZE (gfx950 correction of the guide's HBM section), write = WRITE_SIZE.  (bytes, profiles/ file)
TRAFFIC = {
    "zgemm": (3.51e9, "profiles/r2_bench_headline_summary.txt"),     # zgemm_lin_kernel<.,0>: 3230 MB read + 277 MB written
    "zherm": (3.84e9, "profiles/r2_bench_headline_summary.txt"),     # zgemm_lin_kernel<1,1> 3778 MB + the 64 MB fix-up pass
    "rgemm": (1.81e9, "profiles/r1_bench_v5_realpath_summary.txt"),
    "rsym": (2.68e9, "profiles/r2_bench_dense_real_summary.txt"),    # symmetric-output rgemm 2614 MB + fix-up 64 MB
    "spmm": (8.50e8, "profiles/r2_bench_headline_row_kernels_summary.txt"),     # (2 x 754 MB rsp_mm_slab + 1043 MB rsp_mm_dot_slab) / 3
    # state-column form (profiles/r3_bench_headline_summary.txt): the forward launch moves the states in and out and nothing else
    # (269 MB read + 268 MB written = 1.0x algorithmic); the adjoint launch streams its per-workgroup scratch of forward terms:
    # 1.67 GB read + 1.93 GB written PER STEP of the chain on the fabric side (L2 misses; Infinity-Cache hits are counted)
    "statecol_fwd": (5.376e8, "profiles/r3_bench_headline_summary.txt"),
    "statecol_bwd_per_step": (3.60e9, "profiles/r3_bench_headline_summary.txt"),     # exact form (xs_backward_kernel)
    # quadrature form (xs_backward_q_kernel, round 4): no forward-term scratch; what is left per step is register spill traffic
    # (profiles/r4_bench_headline_summary.txt: 752 MB read + 640 MB written per average launch of 11 steps = the states and costates in
    # and out, 1074 MB, + 29 MB per step -- at degree 6, the economized step polynomial: 25-29 MB by run; 30.6 MB at Taylor's degree 7)
    # (round 5's last build, profiles/r5c_bench_headline_summary.txt: 724 MB read + 627 MB written per average launch of 11 steps =
    # 1074 MB + 25 MB per step; 21 MB per step in profiles/r5b_bench_headline_summary.txt)
    "statecol_bwdq_per_step": (2.5e7, "profiles/r5c_bench_headline_summary.txt"),
    "xr": (6.60e8, "profiles/r2_bench_headline_summary.txt"),        # (2 x 268.8 MB xr_forward [U written] + 1441 MB xr_reverse [Ubar
                                                                     # read + its power scratch written back and re-read]) / 3
}



_JSON_FD = None


def claim_stdout():
    """Keep stdout for the ONE JSON line: everything else a rank (or a library it loads -- RCCL prints a version banner, gloo
    its rank table -- on stdout) writes goes to stderr from here on."""
    global _JSON_FD
    if _JSON_FD is None:
        sys.stdout.flush()
        _JSON_FD = os.dup(1)
        os.dup2(2, 1)


CONFIG5_CHAIN = 4000   # BASELINE configs[4]: 4000 Trotter steps
LINE_LIMIT = 4000   # bytes: the driver keeps an 8000-character tail of stdout; the ONE line must fit with room to spare


def _sig(x, nd=6):
    """Floats to nd significant digits (the line is a summary; the extras file keeps full precision)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float("inf"), float("-inf")):
        return None
    return float("{:.{}g}".format(x, nd))


def short_roofline(b):
    """The contract's roofline object in its short form: numbers and a kernel name of at most 80 characters."""
    if not b:
        return None
    out = {"bound": b["bound"], "kernel": str(b.get("kernel", ""))[:80]}
    for k in ("achieved", "peak", "unit", "frac", "traffic", "traffic_over_algorithmic", "launches", "ms_per_launch"):
        if k in b:
            out[k] = _sig(b[k])
    out.setdefault("traffic", None)
    return out


def compact_line(full):
    """What rank 0 prints: the contract's fields, `roofline` / `cpu_baseline` in short form, `by_input_structure` as
    {name: [steps_per_s, roofline_frac]}, and the path of the extras file that holds everything else."""
    out = {}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "backend", "trials"):
        if k in full:
            v = full[k]
            out[k] = {kk: _sig(vv) for kk, vv in v.items()} if isinstance(v, dict) else _sig(v)
    cfg = full.get("config", {})
    out["config"] = {k: _sig(v) for k, v in cfg.items()
                     if k == "workload" or (isinstance(v, (int, float, bool)) or (isinstance(v, str) and len(v) <= 48))}
    out["config"]["workload"] = str(cfg.get("workload", ""))[:200]
    out["roofline"] = short_roofline(full.get("roofline"))
    if full.get("roofline_other"):
        out["roofline_other"] = short_roofline(full["roofline_other"])
    if full.get("roofline_mfma"):   # the north-star kernel (batched complex GEMM of the explicit-propagator chain, same inputs)
        m = short_roofline(full["roofline_mfma"])
        out["roofline_mfma"] = {k: m[k] for k in ("kernel", "achieved", "peak", "frac", "ms_per_launch", "launches") if k in m}
    if full.get("full_chain"):
        out["full_chain"] = [_sig(x) for x in full["full_chain"]]
    ee = full.get("end_to_end") or {}
    if "executed_mfma_frac_of_peak" in ee:
        out["end_to_end"] = {"executed_mfma_frac_of_peak": _sig(ee["executed_mfma_frac_of_peak"])}
    if full.get("beside"):
        out["beside"] = {k: [_sig(x) for x in v] if isinstance(v, (list, tuple)) else _sig(v) for k, v in full["beside"].items()}
    cb = full.get("cpu_baseline")
    if cb:
        c = {k: _sig(cb.get(k)) for k in ("value", "unit", "cores", "kind")}
        c["sample"] = str(cb.get("sample", ""))[:160]
        cs = cb.get("c_single_thread")
        if isinstance(cs, dict) and "forward_steps_per_s_at_sample" in cs:
            c["c_single_thread"] = {k: _sig(cs[k]) for k in ("forward_steps_per_s_at_sample", "extrapolated_forward_steps_per_s", "cores")}
        out["cpu_baseline"] = c
    st = full.get("by_input_structure")
    if st:
        out["by_input_structure"] = {k: [_sig(v["value"], 4), _sig((v.get("roofline") or {}).get("frac"), 3)] for k, v in st.items()}
    for k in ("max_norm_drift",):
        if k in full:
            out[k] = _sig(full[k])
    out["extras"] = full.get("extras")
    # never exceed the limit: drop the optional objects, least important first
    for drop in (None, "by_input_structure", "beside", "roofline_other", "trials", "roofline_mfma"):
        if drop:
            out.pop(drop, None)
        if len(json.dumps(out)) <= LINE_LIMIT:
            break
    return out


def write_extras(full, path):
    """Everything measured (notes, per-class tables, other configs, the full chain) beside the line; returns the path written
    (relative to the repo root when inside it), or None when nowhere is writable."""
    for cand in (path, os.path.join("/tmp", os.path.basename(path))):
        try:
            os.makedirs(os.path.dirname(os.path.abspath(cand)), exist_ok=True)
            with open(cand, "w") as f:
                json.dump(full, f, indent=1)
            ab = os.path.abspath(cand)
            return os.path.relpath(ab, ROOT) if ab.startswith(ROOT + os.sep) else ab
        except OSError:
            continue
    return None


_EXTRAS_PATH = os.path.join(ROOT, "bench_extras.json")


def set_extras_path(path):
    global _EXTRAS_PATH
    _EXTRAS_PATH = path


def emit(full):
    full["extras"] = None
    full["extras"] = write_extras(full, _EXTRAS_PATH)
    line = (json.dumps(compact_line(full)) + "\n").encode()
    assert len(line) <= LINE_LIMIT + 1, len(line)
    if _JSON_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


# ----------------------------------------------------------------------------------------------------------------
# roofline blocks from the library's per-launch HIP events
# ----------------------------------------------------------------------------------------------------------------

def roofline_blocks(prof, info, d):
    """One block per kernel class of a dense-engine profile (`dq_profile_get`), each with its share of the timed region."""
    # 3M runs on the 128 x 128 tile (fp64), which a launch takes when M, N >= 128 and it has at least 64 tiles (zgemm_pick_tile):
    # every d x d product from d = 1024 up, and below that the BATCHED launches, which carry nearly all of the class's flops
    # (configs[1]: zgemm_kernel<double,128,128,8,2,4,0,0,1,...> in the kernel trace).  Launches that fall to the 64 tile run 4M and
    # execute more than is counted here: `frac` may understate, never overstate.  Below d = 128 everything is 4M.
    three_m = bool(info["zgemm_alg"]) and d >= 128
    f3 = 0.75 if three_m else 1.0
    nt = (d + 127) // 128
    tri = (nt * (nt + 1) / 2) / float(nt * nt) if d >= 1024 else 1.0   # Hermitian-output form: tiles on/above the diagonal

    def rate(p, scale=1.0):
        return p["flops"] * scale / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0

    def per(p, key):
        return p[key] / max(1, p["launches"])

    def mfma(name, p, kernel, exec_factor, alg_bytes, note):
        tr, src = TRAFFIC[name]
        use_tr = d == 4096 and p["launches"] > 0
        return {"bound": "mfma", "kernel": kernel,
                "achieved": rate(p, exec_factor), "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": rate(p, exec_factor) / PEAK_F64_MFMA_TFLOPS,
                "achieved_algorithmic": rate(p), "executed_over_algorithmic_flops": exec_factor,
                "traffic": tr if use_tr else None, "traffic_source": src if use_tr else None,
                "traffic_over_algorithmic": (tr / alg_bytes) if (use_tr and alg_bytes) else None,
                "launches": p["launches"], "flops_per_launch": per(p, "flops"), "ms_per_launch": per(p, "ms"),
                "ms_in_timed_region": p["ms"], "note": note}

    z, zh, rg, rs, sp = prof["zgemm"], prof["zherm"], prof["rgemm"], prof["rsym"], prof["spmm"]
    alg = "3M: 3 real MFMAs per complex tile step" if three_m else "4M"
    lin = int(info["zgemm_alg"]) == 2 and d >= 1024
    blocks = {
        "zgemm": mfma("zgemm", z, "zgemm_lin_kernel<.,0> (complex GEMM, 3M, 16 waves x 32x32, LDS-DMA)" if lin else
                      "zgemm_kernel<128,128,8,2,4,*,*,{}> (batched complex GEMM, {})".format(int(three_m), "3M" if three_m else "4M"),
                      f3, per(z, "bytes"),
                      "v_mfma_f64_16x16x4_f64, {}; linear form: 128x128 tile, 16 waves x 32x32, LDS-DMA staging, 4 waves/SIMD (operands "
                      "without a k-major form stay on zgemm_kernel<128,128,8,2,4>).  achieved = EXECUTED MFMA flops (6*M*N*K per launch in "
                      "the 3M form, 8*M*N*K in 4M) / sum of the launches' HIP-event durations (rank 0); achieved_algorithmic = 8*M*N*K / "
                      "the same time (can exceed the peak in the 3M form)".format(alg)),
        "zherm": mfma("zherm", zh, "zgemm_lin_kernel<1,1> ((anti-)Hermitian-output complex GEMM, 3M)" if lin else
                      "zgemm_kernel<...,HERM> ((anti-)Hermitian-output complex GEMM)",
                      f3 * tri, per(zh, "bytes"),
                      "tiles on/above the diagonal + mirror, K-split tail, {}; executed = the {}/{} tiles on and above the diagonal x (6 or "
                      "8)*K*128*128; algorithmic = 8*M*N*K of the delivered product".format(alg, nt * (nt + 1) // 2, nt * nt)),
        "rsym": mfma("rsym", rs, "rgemm_kernel<128,128,16,4,4,sym> (symmetric-output real GEMM, LDS-DMA)",
                     1.0, per(rs, "bytes"),
                     "one- and two-product launches; executed = algorithmic: 2*K*128*128 per tile on/above the diagonal (the mirror "
                     "half is written, not computed)"),
        "rgemm": mfma("rgemm", rg, "rgemm_kernel<128,128,16,4,4> (real GEMM, full output, LDS-DMA)", 1.0, per(rg, "bytes"),
                      "executed = algorithmic = 2*M*N*K"),
    }
    if info.get("xor_regular", 0) > 0:
        # column form: the class holds xr_forward_kernel / xr_reverse_kernel; booked flops = 2 per (row, diagonal, power, column)
        # with the diagonal included, so LDS reads = flops / 2 * R / (R + 1) doubles
        R = max(1.0, float(info["xor_regular_found"]))
        # (complex column form, xor_regular = 2: 8 booked flops and 16 B per gathered value)
        lds_bytes = sp["flops"] * (2.0 if info["xor_regular"] > 1 else 4.0) * R / (R + 1.0)
        gbps = lds_bytes / sp["ms"] * 1e-6 if sp["ms"] > 0 else 0.0
        tr, src = TRAFFIC["xr"]
        use_tr = d == 4096 and sp["launches"] > 0 and tr is not None
        blocks["spmm"] = {
            "bound": "lds", "kernel": "xr_forward_kernel / xr_reverse_kernel (columns of U / of Ubar in LDS)",
            "kernel_detail": "XOR-regular sparse generator: a column of U = cos X + i sin X, or of its reverse mode, as a chain of sparse "
                             "matrix-vector products on a vector held in LDS; partner rows = ds_read_b64 at i ^ x_j",
            "achieved": gbps, "peak": PEAK_LDS_GBPS, "unit": "GB/s", "frac": gbps / PEAK_LDS_GBPS,
            "hbm_gbps_algorithmic": sp["bytes"] / sp["ms"] * 1e-6 if sp["ms"] > 0 else 0.0,
            "traffic": tr if use_tr else None, "traffic_source": src if use_tr else None,
            "traffic_over_algorithmic": (tr / per(sp, "bytes")) if (use_tr and sp["bytes"]) else None,
            "launches": sp["launches"], "ms_per_launch": per(sp, "ms"), "ms_in_timed_region": sp["ms"],
            "note": "achieved = LDS bytes gathered (8 B per row, off-diagonal XOR-diagonal, power and column) / HIP-event time, against the "
                    "LDS array's 256 B/clk/CU; algorithmic HBM bytes are one complex d x d matrix per launch (U written, or Ubar read): "
                    "hbm_gbps_algorithmic.  The kernels are latency-bound (8-16 waves per CU, one barrier per power), not at either roof"}
    else:
        tr, src = TRAFFIC["spmm"]
        use_tr = d == 4096 and sp["launches"] > 0
        gbps = sp["bytes"] / sp["ms"] * 1e-6 if sp["ms"] > 0 else 0.0
        blocks["spmm"] = {
            "bound": "hbm", "kernel": "rsp_mm_slab_kernel / rsp_mm_dot_slab_kernel (sparse x dense passes)",
            "kernel_detail": "out[i,:] = sum over nnz(X[i,:]) of w X_m[j,:]",
            "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
            "traffic": tr if use_tr else None, "traffic_source": src if use_tr else None,
            "traffic_over_algorithmic": (tr / per(sp, "bytes")) if (use_tr and sp["bytes"]) else None,
            "launches": sp["launches"], "ms_per_launch": per(sp, "ms"), "ms_in_timed_region": sp["ms"],
            "note": "achieved = ALGORITHMIC bytes (operand read once, product written once: 2-3 x 8 d^2) / HIP-event time; the partner "
                    "rows of a row's nonzeros are re-read from L2 / Infinity Cache: `traffic` is the fabric-side PMC figure per launch"}
    # state-column form (statecol.hip): `bytes` of these classes = LDS bytes gathered by the sweeps (diffqc_hip.h); HBM traffic is
    # the states in and out (2 x 16 d B per launch) -- negligible, so the kernels are priced against the LDS array
    quad = info.get("state_columns_quad_nodes", 0) > 0
    for name, kernel, detail in (
            ("statecol_bwd", "xs_backward_q_kernel (adjoint sweep, quadrature form, of whole chains on state columns in LDS)",
             "per step and column: lambda_t = U_t^H lambda_{t+1} as L plain sweeps that also accumulate the interior node values of a "
             "Gauss-Lobatto rule on the step integral of dexp; psi_t = U_t^H psi_{t+1} as L sweeps that pair the partner values they "
             "fetch with the thread's rows of the rule's combinations q_m: 2L sweeps of R partner reads per row, no scratch") if quad else
            ("statecol_bwd", "xs_backward_kernel (adjoint sweep of whole chains on state columns in LDS)",
             "per step and column: reversal psi_t = U_t^H psi_{t+1} (L sweeps), the forward terms x_l (L - 1 sweeps, parked in an L2 / "
             "Infinity-Cache resident scratch), Horner on lambda with the gradient pairings (L sweeps): 3L - 1 sweeps of R partner reads per row"),
            ("statecol_fwd", "xs_forward_kernel (whole Trotter chains on state columns in LDS)",
             "per step and column: L sweeps of R partner reads per row (economized polynomial of exp(-i dt H(t)) applied to the column)")):
        sc = prof.get(name)
        if not sc:
            continue
        gbps = sc["bytes"] / sc["ms"] * 1e-6 if sc["ms"] > 0 else 0.0
        # fabric-side bytes per launch (PMC) at the headline shape (d = 4096, 4096 columns, fp64) and the algorithmic HBM bytes: the
        # states in and out (forward), states and costates in and out (adjoint) -- per launch, whatever the chain length
        batch = info.get("batch", 0)
        use_tr = d == 4096 and batch == 4096 and sc["launches"] > 0
        alg_hbm = 16.0 * d * batch * (2 if name == "statecol_fwd" else 4)
        if name == "statecol_fwd":
            tr, src = TRAFFIC["statecol_fwd"]
        else:
            tr, src = TRAFFIC["statecol_bwdq_per_step" if quad else "statecol_bwd_per_step"]
            tr = tr * float(info.get("chain_steps", 0)) if tr is not None else 0.0
            if quad and tr > 0:
                tr += alg_hbm   # (the PMC figure of the quadrature form is the states in and out + the per-step remainder)
        blocks[name] = {
            "bound": "lds", "kernel": kernel, "kernel_detail": detail,
            "achieved": gbps, "peak": PEAK_LDS_GBPS, "unit": "GB/s", "frac": gbps / PEAK_LDS_GBPS,
            "vector_f64_tflops": rate(sc), "vector_f64_frac_of_peak": rate(sc) / PEAK_F64_MFMA_TFLOPS,
            "traffic": tr if use_tr and tr > 0 else None, "traffic_source": src if use_tr and tr > 0 else None,
            "traffic_over_algorithmic": (tr / alg_hbm) if use_tr and tr > 0 and alg_hbm > 0 else None,
            "algorithmic_hbm_bytes_per_launch": alg_hbm if batch else None,
            "launches": sc["launches"], "ms_per_launch": per(sc, "ms"), "ms_in_timed_region": sc["ms"],
            "note": "achieved = LDS bytes the sweeps gather (16 B per row, XOR-diagonal with lane bits and sweep; partners at offsets "
                    "without lane bits come from the thread's own registers) / HIP-event time, against the LDS array's 256 B/clk/CU; "
                    "vector_f64_* = the kernel's fp64 FMA work against the 78.6 TF/s vector peak (the other resource the sweeps load); "
                    "algorithmic HBM bytes: the states (and costates) in and out, once per CALL (not per step); `traffic` of the adjoint "
                    "launch: exact form = its scratch stream of forward terms (2 L d 16 B per column and step); quadrature form = "
                    "register spills only"}
    total = sum(v["ms"] for v in prof.values())
    for b in blocks.values():
        b["share_of_kernel_time"] = b["ms_in_timed_region"] / total if total > 0 else 0.0
    return blocks


def ranked(blocks):
    return [b for _, b in sorted(blocks.items(), key=lambda kv: -kv[1]["ms_in_timed_region"]) if b["launches"] > 0]


def executed_tflops(blocks, seconds):
    fl = sum(b["achieved"] * 1e12 * b["ms_in_timed_region"] * 1e-3 for b in blocks.values() if b["bound"] == "mfma")
    return fl / seconds / 1e12 if seconds > 0 else 0.0


INFO_KEYS = ("degree", "squarings", "products", "theta", "checkpoint_stride", "time_chunk", "zgemm_alg", "real_path",
             "real_hamiltonians", "sparse_controls", "sparse_generator", "control_nonzeros", "xor_regular", "xor_regular_found", "state_columns",
             "state_columns_quad_nodes", "state_columns_taylor_degree", "state_columns_compose_segments")


def describe_expm(info):
    if info.get("state_columns_compose_segments", 0) > 0:
        return ("composed form from state columns: H is a sum of Pauli strings ({} XOR-diagonals), so the product of the chain's propagators "
                "is what the state-column forward kernel makes of the identity's columns -- {} time segments of dim columns each, "
                "economized polynomial of degree {} (Taylor would need {}), their propagators multiplied by a tree of batched products and "
                "applied to the batch once (no step exponential, no product per step), theta<={:.3g}".format(
                    int(info["xor_regular_found"]), int(info["state_columns_compose_segments"]), int(info["degree"]),
                    int(info.get("state_columns_taylor_degree", 0)), info["theta"]))
    if info.get("state_columns", 0) > 0:
        return ("state-column form: H is a sum of Pauli strings ({} XOR-diagonals), so NO propagator is built -- every state column "
                "stays in LDS for the whole chain and each step applies the Chebyshev-economized polynomial of degree {} (Taylor would "
                "need {}) of exp(-i dt H(t)) as sweeps of partner reads; adjoint sweep the same way (statecol.hip), theta<={:.3g}".format(
                    int(info["xor_regular_found"]), int(info["degree"]), int(info.get("state_columns_taylor_degree", 0)), info["theta"]))
    if info.get("xor_regular", 0) > 1:
        return ("Taylor degree {0}: H complex Hermitian and a sum of Pauli strings ({1} XOR-diagonals with entries +-c or +-i c, a real "
                "diagonal): every row of U = exp(-i dt H) is a chain of {0} complex sparse matrix-vector products held in LDS (xorexp.hip, "
                "zxr_* kernels; no assembly, no tape, no MFMA product in the step exponential), theta<={2:.3g}".format(
                    int(info["degree"]), int(info["xor_regular_found"]), info["theta"]))
    if info["sparse_generator"] > 0 and info.get("xor_regular", 0) > 0:
        return ("cos X + i sin X, Taylor degree {0}: drift and controls are sums of Pauli strings ({1} XOR-diagonals + the diagonal), so every "
                "column of U is a chain of {2} sparse matrix-vector products held in LDS (xorexp.hip: no dense power, no MFMA product and "
                "no HBM traffic but U itself in the step exponential; reverse mode from the rows of Ubar the same way), theta<={3:.3g}".format(
                    int(info["degree"]), int(info["xor_regular_found"]), int(info["products"]), info["theta"]))
    if info["sparse_generator"] > 0:
        return ("cos X + i sin X, Taylor degree {0}: drift and controls sparse, so X_t is sparse -- the powers X^2 .. X^{0} by {1} sparse x dense "
                "passes, cos/sin as linear combinations of them (no MFMA product in the step exponential), theta<={2:.3g}".format(
                    int(info["degree"]), int(info["products"]), info["theta"]))
    if info["real_path"] > 0:
        return "cos X + i sin X, Taylor degree {} ({} real symmetric-output products{}), theta<={:.3g}".format(
            int(info["degree"]), int(info["products"]),
            "; sparse controls: scatter assembly, last reverse-mode level as row dot products" if info["sparse_controls"] > 0 else "",
            info["theta"])
    return "Taylor degree {} ({} complex products, {} squarings), theta<={:.3g}".format(
        int(info["degree"]), int(info["products"]), int(info["squarings"]), info["theta"])


