#!/usr/bin/env python3
"""Benchmark of the propagator hot path on MI355X.  Default = the headline: Trotter steps/s (forward + adjoint gradient),
n = 12 qubits, batch = 4096 per GPU (BASELINE.json configs[2]; configs[3] = the same per-GPU load on 8 GPUs).

    python bench.py --gpus N --steps K --warmup W            # N > 1: starts N child ranks itself (torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               # ... or is started as one rank of N by a launcher

Rank 0 prints ONE JSON line.  `--config 3` (default) is the headline; `--config 2` is BASELINE configs[1] (8-qubit ring
QAOA, 1000 steps, batch 1024, forward); `--config 5` is configs[4] (18 qubits, complex64, batch 512 over the GPUs, the
matrix-free engine, HBM-bound).  `--workload dense-real | dense-complex` replaces the Ising + X_k inputs of config 3 by
dense random matrices of the same size and step norm (what `set_H`, diffqc.cc:43-67, accepts in general).

A "step" (config 3) is ONE Trotter step of the fwd+adjoint pass over one GPU's whole batch: 12-qubit random Ising H stored
dense (d = 4096), K = 12 controls X_k, B = 4096 states, fp64 complex.  The timed region is one `dq_energy_grad` call over a
chain of K steps -- K forward steps (assembly, exp, U.Psi), then the K-step adjoint sweep -- plus, for N > 1, the
collectives of the multi-GPU form.  At this shape (batch >= dim) the backward sweep runs in operator form and keeps no
state checkpoints, so the per-step cost does not depend on the chain length (`other_configs.full_chain_S2000` is the whole
2000-step configuration in one call).  Inputs (H0, H_k, M, Psi0, pulse table) are resident in HBM before the clock starts.
N GPUs = N batch shards of 4096 (weak scaling): value = N*K / max-over-ranks wall time.

THE HEADLINE NUMBER DEPENDS ON THE STRUCTURE OF THE INPUTS.  The engine inspects H0 / H_k at `dq_set_hamiltonians`: real
matrices take the cos X + i sin X form on real products, sparse controls their nonzeros only, a sparse drift too turns the
whole step exponential into sparse x dense passes, and a generator that is a sum of Pauli strings (XOR-regular pattern)
builds every column of U in LDS with no dense intermediate at all.  BASELINE's workload (diagonal Ising drift, X_k controls)
is of the last kind.  `by_input_structure` in the JSON line gives, in the same run and with their own `roofline` blocks, the
rates for dense real symmetric and dense complex Hermitian inputs of the same size (random data, 20-step chains), and for
the Ising inputs with the structure ignored step by step (gradients equal to rounding).

JSON extras: `roofline` = the kernel class with the largest share of the timed region, decided live from the per-launch
HIP events the library records on the launch stream.  For MFMA classes `achieved` is the EXECUTED matrix-pipe rate (the
3M complex product executes 6*M*N*K flops for 8*M*N*K algorithmic ones; the Hermitian-output form only the tiles on and
above the diagonal), so `frac` = achieved / peak is a fraction of the matrix pipe; the algorithmic rate is
`achieved_algorithmic`.  `traffic` = HBM-side bytes per launch from rocprofv3 PMC passes (`traffic_source` names the
profiles/ file), `traffic_over_algorithmic` its ratio to the algorithmic bytes.  `cpu_baseline` = the oracle (the
reference's scipy expm + matmul step) on this box's host cores, plus the single-thread C restatement of diffqc.cc.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
    "statecol_bwd_per_step": (3.60e9, "profiles/r3_bench_headline_summary.txt"),
    "xr": (6.60e8, "profiles/r2_bench_headline_summary.txt"),        # (2 x 268.8 MB xr_forward [U written] + 1441 MB xr_reverse [Ubar
                                                                     # read + its power scratch written back and re-read]) / 3
}


# ----------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` with no WORLD_SIZE starts the N ranks itself
# ----------------------------------------------------------------------------------------------------------------

def launch_ranks(a, argv):
    """Start N fresh child ranks (one per GPU) BEFORE this process makes any GPU call and pass rank 0's JSON line
    through.  The parent never initialises HIP (torch.cuda.device_count() does not, on this image), and it starts the
    ranks as child processes -- it never re-execs itself."""
    backend = os.environ.get("DQ_DIST_BACKEND", "nccl")
    if backend == "nccl":
        import torch
        have = torch.cuda.device_count()
        if have < a.gpus:
            print("bench.py: --gpus {} but this node shows {} GPU(s)".format(a.gpus, have), file=sys.stderr)
            return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + [x for x in argv if x != "--launch"]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


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
    for drop in (None, "by_input_structure", "beside", "roofline_other", "trials"):
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


def emit(full):
    full["extras"] = None
    full["extras"] = write_extras(full, _EXTRAS_PATH)
    line = (json.dumps(compact_line(full)) + "\n").encode()
    assert len(line) <= LINE_LIMIT + 1, len(line)
    if _JSON_FD is None:
        sys.stdout.write(line.decode()); sys.stdout.flush()
    else:
        os.write(_JSON_FD, line)


class Dist:
    """One process per GPU.  RCCL ("nccl") on device tensors; DQ_DIST_BACKEND=gloo stages the collectives through the host
    and shares the devices round-robin (rehearsal on a box with fewer GPUs than ranks: RCCL refuses two ranks per device)."""

    def __init__(self, a):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != a.gpus:
            if self.rank == 0:
                print("bench.py: --gpus {} but WORLD_SIZE {}".format(a.gpus, self.world), file=sys.stderr)
            sys.exit(2)
        self.backend = os.environ.get("DQ_DIST_BACKEND", "nccl")
        if self.backend != "nccl":
            local = local % max(1, torch.cuda.device_count())
        self.local = local
        torch.cuda.set_device(local)
        self.dev = torch.device("cuda", local)
        self.on = self.world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or "MASTER_PORT" in os.environ
        if self.on:  # launched by torch.distributed.run (also exercised at world size 1)
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            os.environ.setdefault("RANK", str(self.rank))
            os.environ.setdefault("WORLD_SIZE", str(self.world))
            import datetime
            tmo = datetime.timedelta(minutes=5)  # a mismatched collective fails the run instead of hanging it
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev, timeout=tmo)
            else:
                dist.init_process_group(self.backend, timeout=tmo)
            self.world = dist.get_world_size()

    def all_reduce_(self, t, op):  # in place on a device tensor, whatever the backend
        if not self.on:
            return
        if self.backend == "nccl":
            self.dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            self.dist.all_reduce(h, op=op)
            t.copy_(h)

    def fence(self):
        self.torch.cuda.synchronize()
        if self.on:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, seconds):
        if not self.on:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.dev)
        self.all_reduce_(t, self.dist.ReduceOp.MAX)
        return float(t.item())

    def name(self):
        return ("nccl (RCCL over xGMI)" if self.backend == "nccl" else self.backend + " (host-staged rehearsal)") if self.on else "none (single process)"

    def close(self):
        if self.on:
            self.dist.barrier()
            self.dist.destroy_process_group()


# ----------------------------------------------------------------------------------------------------------------
# CPU baseline (the oracle: test infrastructure, used here only as the reported baseline)
# ----------------------------------------------------------------------------------------------------------------

def cpu_baseline(n, B, seed, dt, u_row):
    """One fwd+adjoint Trotter step of the SAME workload on the host cores, through the oracle: the reference's
    per-step calls (scipy.linalg.expm + np.matmul, sim_plain.py:145-146) for the forward half, and the exact
    adjoint with ONE scipy.linalg.expm_frechet per step (oracle.energy_grad_one_frechet) for the gradient half.
    Beside it: the single-thread C restatement of diffqc.cc's trotter() (Eigen without OpenMP is
    single-threaded) on a bounded sample."""
    from oracle import trotter_oracle as orc
    from diffquantum_amd import problems
    try:
        from threadpoolctl import threadpool_info
        cores = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        cores = os.cpu_count() or 1
    H0, Hs, M = problems.random_ising(n, seed)
    psi = problems.random_states(1 << n, min(B, 64), seed)
    psi = np.ascontiguousarray(np.tile(psi, (1, (B + psi.shape[1] - 1) // psi.shape[1]))[:, :B])
    t0 = time.perf_counter()
    orc.propagate(H0, list(Hs), u_row[None, :], dt, psi)
    t_fwd = time.perf_counter() - t0
    t0 = time.perf_counter()
    orc.energy_grad_one_frechet(H0, list(Hs), M, u_row[None, :], dt, psi)
    t_all = time.perf_counter() - t0
    out = {"value": 1.0 / t_all, "unit": "Trotter steps/s (fwd+adjoint)", "cores": int(cores), "kind": "port",
           "forward_only_steps_per_s": 1.0 / t_fwd,
           "sample": "1 fwd+adjoint step of the same workload (n={}, d={}, B={}) through the oracle (scipy expm, matmul, expm_frechet): "
                     "{:.1f} s".format(n, 1 << n, B, t_all),
           "sample_detail": "dH assembly, scipy.linalg.expm, np.matmul, one scipy.linalg.expm_frechet, {} contractions; forward-only "
                            "step: {:.1f} s".format(n, t_fwd)}
    try:
        out["c_single_thread"] = c_baseline(n, seed, dt)
    except Exception as e:  # noqa: BLE001 -- a reported extra
        out["c_single_thread"] = {"error": repr(e)}
    return out


def c_baseline(n, seed, dt, n_small=9):
    """oracle/diffqc_oracle.c (own Higham-2005 Pade + LU, plain loops, one thread): forward steps of diffqc.cc's
    trotter() (`psi0 = dH.exp() * psi0`, diffqc.cc:192-198, batch 1).  At d = 4096 one step would take the better
    part of an hour, so the sample is the same problem family at n = 9 (d = 512); the d = 4096 figure is the O(d^3)
    extrapolation and is labelled as such."""
    from oracle import c_oracle
    from diffquantum_amd import problems
    ns = min(n, n_small)
    H0, Hs, _ = problems.random_ising(ns, seed)
    u = problems.pulse_table(ns, 1, dt, seed=seed)
    psi = problems.random_states(1 << ns, 1, seed)
    t0 = time.perf_counter()
    c_oracle.propagate(H0, Hs, u, dt, psi)
    el = time.perf_counter() - t0
    scale = float(1 << (n - ns)) ** 3
    return {"forward_steps_per_s_at_sample": 1.0 / el, "cores": 1, "kind": "port",
            "extrapolated_forward_steps_per_s": 1.0 / (el * scale),
            "sample": "1 forward Trotter step, batch 1, n={} (d={}) in {:.2f} s on one thread (C restatement of diffqc.cc: "
                      "assembly + Pade-13/LU expm + matvec); x{:.0f} (d^3) to d={}".format(ns, 1 << ns, el, scale, 1 << n)}


# ----------------------------------------------------------------------------------------------------------------
# roofline blocks from the library's per-launch HIP events
# ----------------------------------------------------------------------------------------------------------------

def roofline_blocks(prof, info, d):
    """One block per kernel class of a dense-engine profile (`dq_profile_get`), each with its share of the timed region."""
    three_m = bool(info["zgemm_alg"]) and d >= 1024          # 3M runs on the 128 tile (fp64); smaller tiles are 4M
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
    for name, kernel, detail in (
            ("statecol_bwd", "xs_backward_kernel (adjoint sweep of whole chains on state columns in LDS)",
             "per step and column: reversal psi_t = U_t^H psi_{t+1} (L sweeps), the forward terms x_l (L - 1 sweeps, parked in an L2 / "
             "Infinity-Cache resident scratch), Horner on lambda with the gradient pairings (L sweeps): 3L - 1 sweeps of R partner reads per row"),
            ("statecol_fwd", "xs_forward_kernel (whole Trotter chains on state columns in LDS)",
             "per step and column: L sweeps of R partner reads per row (Taylor polynomial of exp(-i dt H(t)) applied to the column)")):
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
            tr, src = TRAFFIC["statecol_bwd_per_step"]
            tr = tr * float(info.get("chain_steps", 0))
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
                    "launch is its scratch stream of forward terms (2 L d 16 B per column and step)"}
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
             "real_hamiltonians", "sparse_controls", "sparse_generator", "control_nonzeros", "xor_regular", "xor_regular_found", "state_columns")


def describe_expm(info):
    if info.get("state_columns", 0) > 0:
        return ("state-column form: H is a sum of Pauli strings ({} XOR-diagonals), so NO propagator is built -- every state column "
                "stays in LDS for the whole chain and each step applies the Taylor polynomial of degree {} of exp(-i dt H(t)) as sweeps "
                "of partner reads; adjoint sweep the same way (statecol.hip), theta<={:.3g}".format(
                    int(info["xor_regular_found"]), int(info["degree"]), info["theta"]))
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


def timed_chain(eng, torch, u_dev, dt, psi0, grad_dev):
    """One profiled dq_energy_grad over the chain in u_dev (single GPU, no collective); returns (seconds, loss, profile)."""
    eng.energy_grad(u_dev[:1].contiguous(), dt, psi0, want_final=False, want_energies=False)   # allocations, plan
    eng.profile_enable(True)
    eng.profile_reset()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, _, _, _ = eng.energy_grad(u_dev, dt, psi0, want_final=False, want_energies=False, grad_out=grad_dev)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    prof = eng.profile()
    eng.profile_enable(False)
    return el, loss, prof


def scheme_entry(eng, torch, u_dev, dt, psi0, d, K, name_note):
    g = torch.zeros((u_dev.shape[0], eng.n_controls), dtype=torch.float64, device=u_dev.device)
    el, loss, prof = timed_chain(eng, torch, u_dev, dt, psi0, g)
    info = {k: eng.info(k) for k in INFO_KEYS}
    info["batch"], info["chain_steps"] = int(psi0.shape[1]), int(u_dev.shape[0])
    blocks = roofline_blocks(prof, info, d)
    rk = ranked(blocks)
    return {"value": K / el, "unit": "Trotter steps/s (fwd+adjoint)", "ms_per_step": el / K * 1e3, "chain_steps": K,
            "path": describe_expm(info), "loss": loss, "grad_norm": float(torch.linalg.norm(g).item()),
            "roofline": rk[0] if rk else None, "roofline_other": rk[1] if len(rk) > 1 else None,
            "executed_mfma_tflops": executed_tflops(blocks, el),
            "executed_mfma_frac_of_peak": executed_tflops(blocks, el) / PEAK_F64_MFMA_TFLOPS,
            "kernel_ms": {k: v["ms"] for k, v in prof.items()}, "kernel_launches": {k: v["launches"] for k, v in prof.items()},
            "note": name_note}, g


# ----------------------------------------------------------------------------------------------------------------
# extras of the headline run (single GPU, outside the timed region)
# ----------------------------------------------------------------------------------------------------------------

def by_input_structure(eng, a, dev, u_all, psi0, grad_ref, loss_ref):
    """What the SAME chain costs for other input structures, 20-step chains each with its own roofline."""
    import torch
    from diffquantum_amd import problems
    from diffquantum_amd.engine import Engine
    n, K = a.qubits, a.steps
    d = 1 << n
    u_dev = torch.from_numpy(u_all[:K].copy()).to(dev)
    out = {}
    # (1) the headline's own inputs with their structure ignored step by step: gradients must agree to rounding
    for name, opts, note in (
            ("ising_inputs_gemm_chain", {"state_columns": 0},
             "same inputs on the explicit-propagator chain (state_columns = 0; what any batch > 8 x dim, and every generator that is "
             "not a Pauli sum beyond its exponential, runs): U_t column by column in LDS, then the chain's complex MFMA products"),
            ("ising_inputs_sparse_row_kernels", {"xor_regular": 0},
             "same inputs, the XOR-regular (Pauli-string) structure of the generator ignored: the step exponential as 2q sparse x dense "
             "passes over stored dense powers (what any sparse drift + sparse controls get)"),
            ("ising_inputs_sparse_controls_dense_drift", {"sparse_generator": 0},
             "same inputs, the drift treated as dense: real symmetric MFMA products for the step exponential, controls by their nonzeros"),
            ("ising_inputs_dense_real_path", {"sparse_generator": 0, "sparse_controls": 0},
             "same inputs, drift and controls treated as dense real symmetric matrices"),
            ("ising_inputs_complex_tape", {"real_path": 0},
             "same inputs on the general complex tape (real_path = 0): 11 full + 3 Hermitian-output complex products per step")):
        if "state_columns" not in opts:
            eng.set_option("state_columns", 0)   # (the structure-ignoring schemes all concern the propagator chain)
        for k_, v_ in opts.items():
            eng.set_option(k_, v_)
        ent, g = scheme_entry(eng, torch, u_dev, a.dt, psi0, d, K, note)
        ent["max_abs_grad_diff_vs_headline"] = float((g - grad_ref).abs().max().item())
        ent["abs_loss_diff_vs_headline"] = abs(ent["loss"] - loss_ref)
        out[name] = ent
        eng.set_option("state_columns", -1)
        eng.set_option("xor_regular", -1)
        eng.set_option("sparse_generator", -1)
        eng.set_option("sparse_controls", 1)
        eng.set_option("real_path", a.real_path)
    # (2) dense random data of the same size and step norm: what an arbitrary `set_H` caller gets
    for name, cplx, note in (
            ("dense_real_symmetric_H", False, "dense random real symmetric H0 and 12 dense real symmetric H_k (1-norms 12 and 1, as the Ising + X_k "
                                              "workload): cos X + i sin X on real symmetric-output MFMA products"),
            ("dense_complex_hermitian_H", True, "dense random complex Hermitian H0 and 12 dense complex Hermitian H_k (same 1-norms): the general "
                                                "complex tape -- what diffqc.cc's set_H accepts in general")):
        H0, Hs, M = problems.random_dense_torch(n, n, a.seed, dev, complex_h=cplx)
        with Engine(d, n, device=dev.index or 0) as e2:
            e2.use_torch_stream()
            e2.set_hamiltonians(H0, Hs)
            e2.set_observable(M)
            e2.set_option("plan_steps", max(a.chain, K))  # propagator-cache share / checkpoint stride of the full chain
            del H0, Hs, M
            ent, _ = scheme_entry(e2, torch, u_dev, a.dt, psi0, d, K, note)
        out[name] = ent
        torch.cuda.empty_cache()
    # (3) X and Y drives: a complex Hermitian, sparse (Pauli-string) generator -- column form in complex arithmetic, and the same
    # inputs on the general complex tape
    H0, Hs, M = problems.xy_drives_torch(n, a.seed, dev, n_x=n // 2, n_y=n - n // 2)
    with Engine(d, n, device=dev.index or 0) as e3:
        e3.use_torch_stream()
        e3.set_hamiltonians(H0, Hs)
        e3.set_observable(M)
        e3.set_option("plan_steps", max(a.chain, K))
        del H0, Hs, M
        ent, g5 = scheme_entry(e3, torch, u_dev, a.dt, psi0, d, K,
                               "random Ising drift, X_k controls on half of the qubits and Y_k controls on the other half (I/Q drives): complex "
                               "Hermitian and a sum of Pauli strings -- the state-column form in complex arithmetic (statecol.hip)")
        ent["state_columns"] = e3.info("state_columns")
        out["xy_drives_state_columns"] = ent
        e3.set_option("state_columns", 0)
        ent, g3 = scheme_entry(e3, torch, u_dev, a.dt, psi0, d, K,
                               "the same X / Y-drive inputs on the explicit-propagator chain: the rows of U = exp(-i dt H) built in LDS "
                               "(complex column form, xorexp.hip), then the complex MFMA products")
        ent["xor_regular"] = e3.info("xor_regular")
        ent["max_abs_grad_diff_vs_state_columns"] = float((g3 - g5).abs().max().item())
        out["xy_drives_complex_pauli_H"] = ent
        e3.set_option("xor_regular", 0)
        ent, g4 = scheme_entry(e3, torch, u_dev, a.dt, psi0, d, K,
                               "the same X / Y-drive inputs on the general complex tape (structure ignored)")
        ent["max_abs_grad_diff_vs_column_form"] = float((g4 - g3).abs().max().item())
        out["xy_drives_complex_tape"] = ent
    torch.cuda.empty_cache()
    return out


def full_chain(eng, a, dev, psi0):
    """BASELINE configs[2] in full: the S = 2000 chain in ONE dq_energy_grad call, with the size-independent
    checks a driver can read: norm drift of Psi(T), sum of energies = loss, directional derivative of the gradient against central
    differences of the forward path on a 100-step chain at full d and B."""
    import torch
    from diffquantum_amd import problems
    n, S = a.qubits, a.chain
    u = torch.from_numpy(problems.pulse_table(n, S, a.dt * S, seed=a.seed)).to(dev)
    eng.set_option("plan_steps", 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, grad, fin, en = eng.energy_grad(u, a.dt, psi0)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    free, tot = torch.cuda.mem_get_info()
    res = {"chain_steps": S, "wall_s": el, "steps_per_s": S / el, "hbm_in_use_GB": (tot - free) / 1e9, "loss": loss,
           "sum_of_energies_minus_loss": float(en.sum()) - loss, "grad_norm": float(np.linalg.norm(grad)),
           "grad_finite": bool(np.isfinite(grad).all()),
           "max_norm_drift": float((torch.linalg.norm(fin, dim=0) - 1).abs().max().item())}
    del fin
    S_fd = min(100, S)
    u2 = problems.pulse_table(n, S_fd, a.dt * S_fd, seed=a.seed + 1)
    v = np.random.default_rng(0).standard_normal(u2.shape)
    _, g0, _, _ = eng.energy_grad(u2, a.dt, psi0, want_final=False, want_energies=False)
    eps = 1e-4
    lp = eng.energy(eng.propagate(torch.from_numpy(u2 + eps * v).to(dev), a.dt, psi0))[1]
    lm = eng.energy(eng.propagate(torch.from_numpy(u2 - eps * v).to(dev), a.dt, psi0))[1]
    fd, an = (lp - lm) / (2 * eps), float(np.sum(g0 * v))
    res["directional_derivative_check"] = {"chain_steps": S_fd, "central_difference": fd, "adjoint": an,
                                           "rel_diff": abs(fd - an) / max(abs(an), 1e-30)}
    return res


def other_configs(dev, seed, headline=None):
    """Short measurements of the other BASELINE configs on this GPU (reported as extras, outside the timed region);
    `--config 2` / `--config 5` make them the measured workload of a run of their own."""
    import torch
    from diffquantum_amd.structured import StructuredEngine, ising_x_terms
    out = {}
    out["config2_qaoa_n8_S1000_B1024_fwd"] = measure_config2(dev, seed, 1000, 1024, reps=1)
    c5 = measure_config5(dev, seed, 20, 64, None)
    out["config5_pergpu_n18_B64_c64_structured"] = {
        "fwd_steps_per_s": c5["fwd_steps_per_s"], "fwd_adjoint_steps_per_s": c5["steps_per_s"], "taylor_degree": c5["degree"],
        "roofline": c5["roofline"]}
    # the headline problem itself, given as Pauli strings to the matrix-free engine (complex128): what the same chain
    # costs when the Hamiltonian is never densified (HBM-bound state passes instead of d x d MFMA products)
    if headline is not None:
        n, dt, u, psi0, loss_dense = headline
        with StructuredEngine(n, n, device=dev.index or 0, dtype="c128") as e:
            e.use_torch_stream()
            e.set_terms(ising_x_terms(n, seed))
            e.energy_grad(u[:2], dt, psi0, want_final=False)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ls, gs, _, _ = e.energy_grad(u, dt, psi0, want_final=False)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
            out["headline_problem_matrix_free_c128"] = {
                "fwd_adjoint_steps_per_s": len(u) / el, "taylor_degree": int(e.info("degree")),
                "abs_loss_diff_vs_dense_engine": abs(ls - loss_dense),
                "note": "same H0 / H_k / M as Pauli strings on the structured engine (dq_sctx_*), same states and pulse "
                        "table: an extra, not the headline -- the dense engine takes ANY dense Hermitian H"}
    return out


def measure_config2(dev, seed, S, B, reps=3, first=0):
    """BASELINE configs[1]: 8-qubit ring QAOA (K = 16), S Trotter steps, B states, fp64, forward only."""
    import torch
    from diffquantum_amd import problems
    from diffquantum_amd.engine import Engine
    p = problems.qaoa_ring(8)
    u = torch.from_numpy(problems.pulse_table(16, S, 1.0, seed=seed)).to(dev)
    psi = torch.from_numpy(problems.random_states(256, B, seed, first=first)).to(dev)
    with Engine(256, 16, device=dev.index or 0) as e:
        e.use_torch_stream()
        e.set_hamiltonians(p["H0"].astype(np.complex128), p["Hs"].astype(np.complex128))
        e.propagate(u, 1.0 / S, psi)
        best, fin, times = None, None, []
        for _ in range(max(1, reps)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            fin = e.propagate(u, 1.0 / S, psi)
            torch.cuda.synchronize(); el = time.perf_counter() - t0
            times.append(el)
            best = el if best is None else min(best, el)
        e.profile_enable(True); e.profile_reset()
        e.propagate(u, 1.0 / S, psi)
        prof = e.profile()
        e.profile_enable(False)
        info = {k: e.info(k) for k in INFO_KEYS}
        return {"steps_per_s": S / best, "wall_s": best, "times": times, "operator_form": bool(e.info("composed")),
                "max_norm_drift": float((torch.linalg.norm(fin, dim=0) - 1).abs().max().item()),
                "path": describe_expm(info), "kernel_ms": {k: v["ms"] for k, v in prof.items() if v["launches"]},
                "kernel_launches": {k: v["launches"] for k, v in prof.items() if v["launches"]}, "_prof": prof, "_info": info}


def measure_config5(dev, seed, S, B, dist, n=18, first=0, trials=1):
    """BASELINE configs[4] per GPU: n-qubit Ising + X controls as Pauli strings (never densified), complex64, B states,
    S steps forward and fwd+adjoint on the matrix-free engine; roofline = HBM from the engine's per-pass HIP events."""
    import torch
    from diffquantum_amd import problems
    from diffquantum_amd.structured import StructuredEngine, ising_x_terms
    rng = np.random.default_rng([seed, first])
    st = rng.standard_normal((1 << n, B)) + 1j * rng.standard_normal((1 << n, B))
    st = torch.from_numpy((st / np.linalg.norm(st, axis=0)).astype(np.complex64)).to(dev)
    us = problems.pulse_table(n, S, S * 1e-3, seed=seed)
    with StructuredEngine(n, n, device=dev.index or 0, dtype="c64") as e:
        e.use_torch_stream()
        e.set_terms(ising_x_terms(n, seed))
        e.set_option("plan_steps", CONFIG5_CHAIN)   # the timed slice pays the checkpoint stride (state reversals) of the full chain
        e.propagate(us[:2], 1e-3, st)
        e.set_option("profile", -1); e.set_option("profile", 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.propagate(us, 1e-3, st)
        torch.cuda.synchronize(); el_f = time.perf_counter() - t0
        ms_f, by_f = e.info("prof_ms"), e.info("prof_bytes")
        e.set_option("profile", 0)
        e.energy_grad(us, 1e-3, st, want_final=False)   # warm-up over the SAME chain: the checkpoint buffers are allocated here
        e.set_option("profile", -1); e.set_option("profile", 1)
        times = []
        for _ in range(max(1, trials)):
            if dist is not None:
                dist.fence()
            t0 = time.perf_counter()
            loss, grad, _, _ = e.energy_grad(us, 1e-3, st, want_final=False)
            if dist is not None and dist.on:
                from diffquantum_amd.distributed import allreduce_loss_grad
                loss, grad = allreduce_loss_grad(loss, grad, device=dev)
            if dist is not None:
                dist.fence()
            else:
                torch.cuda.synchronize()
            times.append(time.perf_counter() - t0)
        el_g = float(np.median(times))
        ms_g, by_g, nl = e.info("prof_ms"), e.info("prof_bytes"), e.info("prof_launches")
        passes, dma_passes, stride = int(e.info("passes")), int(e.info("dma_passes")), int(e.info("checkpoint_stride"))
        e.set_option("profile", 0)
        # beside it: the same slice with a checkpoint at every step (what a chain short enough for the HBM would cost; rounds 1-2 timed this)
        e.set_option("plan_steps", 0)
        e.energy_grad(us, 1e-3, st, want_final=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.energy_grad(us, 1e-3, st, want_final=False)
        torch.cuda.synchronize(); el_ck = time.perf_counter() - t0
        stride_ck = int(e.info("checkpoint_stride"))
        gb = by_g / ms_g * 1e-6 if ms_g > 0 else 0.0
        gbf = by_f / ms_f * 1e-6 if ms_f > 0 else 0.0
        return {"steps_per_s": S / el_g, "seconds": el_g, "times": times, "fwd_steps_per_s": S / el_f, "degree": int(e.info("degree")),
                "loss": loss, "grad_norm": float(np.linalg.norm(grad)), "checkpoint_stride": stride,
                "every_step_checkpointed": {"steps_per_s": S / el_ck, "checkpoint_stride": stride_ck},
                "roofline": {"bound": "hbm",
                             "kernel": ("sapply_dma_kernel / schain_dma_kernel (matrix-free passes, LDS-DMA tiles)" if dma_passes
                                        else "sapply_kernel / soverlap_kernel (matrix-free H.psi passes, 2^9-row sub-cubes)"),
                             "passes_per_step": passes / max(S, 1), "dma_passes_per_step": dma_passes / max(S, 1),
                             "achieved": gb, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gb / PEAK_HBM_GBPS,
                             "traffic": (by_g / nl * 1.003 if nl else None), "traffic_over_algorithmic": 1.003,
                             "traffic_source": "profiles/r3_bench_config5_summary.txt: PMC bytes per pass = its state-units x 2^n x B x 8 B (n = 18, B = 512)",
                             "launches": int(nl), "ms_in_timed_region": ms_g, "ms_per_launch": (ms_g / nl if nl else None),
                             "forward_only": {"achieved": gbf, "frac": gbf / PEAK_HBM_GBPS},
                             "note": "fwd+adjoint passes of the timed region: algorithmic bytes (state-units x 2^n x B x 8 B per pass) / "
                                     "HIP-event time; 6.29 TB/s is the measured copy rate of the part (guide), 8 TB/s the spec used as peak"}}


# ----------------------------------------------------------------------------------------------------------------
# config 2 / config 5 as the measured workload
# ----------------------------------------------------------------------------------------------------------------

def run_config2(a, D):
    import torch
    S = a.steps if a.steps is not None else 1000
    B = a.batch if a.batch is not None else 1024
    measure_config2(D.dev, a.seed, min(S, max(a.warmup, 1) * 10), B, reps=1, first=D.rank * B)   # warm-up (library, allocator)
    D.fence()
    m = measure_config2(D.dev, a.seed, S, B, reps=max(1, a.trials), first=D.rank * B)   # its own warm call, then `trials` timed dq_propagate calls
    D.fence()
    el = D.max_over_ranks(float(np.median(m["times"])))
    if D.rank == 0:
        blocks = roofline_blocks(m["_prof"], m["_info"], 256)
        rk = ranked(blocks)
        W_alg = 8.0 * 256 * 256 * B
        out = {"metric": "Trotter steps/sec (forward), n=8 qubits batch={}".format(B), "value": D.world * S / el,
               "unit": "Trotter steps/s (one step = forward over one GPU's {}-state batch; sum over GPUs)".format(B),
               "n_gpus": D.world, "steps": S, "warmup": a.warmup, "ms_per_step": el / S * 1e3, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "backend": D.name(),
               "trials": {"n": len(m["times"]), "ms_per_step_median": el / S * 1e3, "ms_per_step_min": min(m["times"]) / S * 1e3,
                          "ms_per_step_max": max(m["times"]) / S * 1e3},
               "config": {"workload": "BASELINE configs[1]: 8-qubit ring analog QAOA (d=256, 16 controls), {} Trotter steps, batch {} per "
                                      "GPU, fp64 complex, forward only".format(S, B),
                          "workload_detail": "16 controls = 8 edge ZZ + 8 vertex X; the timed region is ONE dq_propagate over the whole chain",
                          "n_qubits": 8, "dim": 256, "n_controls": 16, "batch_per_gpu": B, "global_batch": B * D.world, "chain_steps": S,
                          "operator_form": m["operator_form"], "path": m["path"],
                          "parallelism": "dp{}".format(D.world),
                          "parallelism_detail": "independent batch shards, forward only: no collective"},
               "roofline": rk[0] if rk else None, "roofline_other": rk[1] if len(rk) > 1 else None,
               "end_to_end": {"algorithmic_flops_per_step": W_alg, "algorithmic_tflops": W_alg * S / el / 1e12,
                              "note": "d = 256: the chain is composed by a log-depth tree of batched d x d products (operator form), "
                                      "8 d^3 per step instead of 8 d^2 B; launch-latency, not a roofline, governs at this size",
                              "kernel_ms": m["kernel_ms"], "kernel_launches": m["kernel_launches"]},
               "max_norm_drift": m["max_norm_drift"]}
        emit(out)


def run_config5(a, D):
    S = a.steps if a.steps is not None else 20
    total = a.batch if a.batch is not None else 512
    if total % D.world:
        raise SystemExit("--batch {} (the WHOLE batch of config 5) must divide over {} ranks".format(total, D.world))
    B = total // D.world
    n = a.qubits if a.qubits != 12 else 18
    m = measure_config5(D.dev, a.seed, S, B, D, n=n, first=D.rank * B, trials=a.trials)
    el = D.max_over_ranks(m["seconds"])
    if D.rank == 0:
        out = {"metric": "Trotter steps/sec (fwd+adjoint), n={} qubits batch={} (complex64, matrix-free)".format(n, total),
               "value": S / el, "unit": "Trotter steps/s (one step = fwd+adjoint over the WHOLE {}-state batch, sharded over the GPUs)".format(total),
               "n_gpus": D.world, "steps": S, "warmup": a.warmup, "ms_per_step": el / S * 1e3, "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "backend": D.name(),
               "trials": {"n": len(m["times"]), "ms_per_step_median": el / S * 1e3, "ms_per_step_min": min(m["times"]) / S * 1e3,
                          "ms_per_step_max": max(m["times"]) / S * 1e3},
               "config": {"workload": "BASELINE configs[4]: {}-qubit Ising + X_k as Pauli strings (d={}), batch {} = {} per "
                                      "GPU, complex64, fwd+adjoint, dt=1e-3; {} of {} Trotter steps timed at the full chain's checkpoint stride".format(n, 1 << n, total, B, S, CONFIG5_CHAIN),
                          "n_qubits": n, "dim": 1 << n, "n_controls": n, "batch_per_gpu": B, "global_batch": total, "chain_steps": S,
                          "taylor_degree": m["degree"], "loss": m["loss"], "grad_norm": m["grad_norm"],
                          "forward_only_steps_per_s_rank0": m["fwd_steps_per_s"],
                          "checkpoint_stride": m["checkpoint_stride"], "every_step_checkpointed_steps_per_s_rank0": m["every_step_checkpointed"]["steps_per_s"],
                          "parallelism": "dp{}".format(D.world),
                          "parallelism_detail": "batch shards, one all-reduce of loss + dL/du"},
               "roofline": m["roofline"]}
        emit(out)


# ----------------------------------------------------------------------------------------------------------------
# config 3: the headline
# ----------------------------------------------------------------------------------------------------------------

def run_config3(a, D):
    import torch
    from diffquantum_amd import problems
    from diffquantum_amd.distributed import allreduce_loss_grad, propagator_sharded_energy_grad, time_sliced_energy_grad
    from diffquantum_amd.engine import Engine
    dist = D.dist
    rank, world, dev = D.rank, D.world, D.dev
    n, K, W = a.qubits, a.steps if a.steps is not None else 20, a.warmup
    B = a.batch if a.batch is not None else 4096
    a.steps = K
    d = 1 << n
    # ---- inputs, resident in HBM before the clock starts --------------------------------------
    if a.workload == "ising":
        H0, Hs, M = problems.random_ising_torch(n, a.seed, dev)
    else:
        H0, Hs, M = problems.random_dense_torch(n, n, a.seed, dev, complex_h=(a.workload == "dense-complex"))
    u_all = problems.pulse_table(n, max(K, W, 1), a.dt * max(K, W, 1), seed=a.seed)
    psi0 = torch.from_numpy(problems.random_states(d, B, a.seed, first=rank * B)).to(dev)
    eng = Engine(d, n, device=D.local)
    eng.use_torch_stream()
    eng.set_hamiltonians(H0, Hs)
    eng.set_observable(M)
    M_dev = M if a.multi_gpu == "propagators" else None
    del H0, Hs, M
    eng.set_option("real_path", a.real_path)
    eng.set_option("plan_steps", max(a.chain, K))  # checkpoint/rebuild ratio of the full 2000-step chain
    u_t = torch.from_numpy(u_all[:K].copy()).to(dev)
    grad_t = torch.zeros((K, n), dtype=torch.float64, device=dev)

    # Propagator cache of the sliced form: a rank's slice of the planned chain is short enough to be cached WHOLE
    # ((q + 3) d^2 fp64 words per step, 2q + 2 with the sparse-generator scheme: 0.8-1.07 GB per step at q = 3), so the budget is
    # what the slice needs, capped at 72 % of the HBM (the arena itself only grows to what the steps of THIS call need)
    slice_steps = (max(a.chain, K) + world - 1) // world
    sliced_cache_bytes = min(8.0 * d * d * 8 * slice_steps + (1 << 20), 0.72 * torch.cuda.get_device_properties(D.local).total_memory)

    def run_replicated(u_dev, grad_dev):
        eng.set_option("plan_steps", max(a.chain, K))
        eng.set_option("cache_bytes", 0)
        loss, g, _, _ = eng.energy_grad(u_dev, a.dt, psi0, want_final=False, want_energies=False, grad_out=grad_dev)
        if not D.on:
            return loss, g
        return allreduce_loss_grad(loss, g, device=dev)

    def run_sliced(u_dev, grad_dev):
        # chain split over TIME across the ranks: each rank builds and sweeps 1/world of the steps for the
        # WHOLE batch (its own shard enters through W_S only); all-gather of the slice products, all-reduce of W_S,
        # all-reduce of the gradient rows + loss
        eng.set_option("plan_steps", (max(a.chain, K) + world - 1) // world)
        eng.set_option("cache_bytes", sliced_cache_bytes)
        loss, g, _, _ = time_sliced_energy_grad(eng, u_dev, a.dt, psi0, device=dev)
        grad_dev.copy_(torch.from_numpy(g).to(dev))
        return loss, grad_dev

    def run_propagators(u_dev, grad_dev):
        # SURVEY 8-F2 in its literal form: every rank builds 1/world of each window's propagators, all-gather, state-form chain on
        # its own batch shard, reduce-scatter of the propagators' cotangents to their owners, one all-reduce of the gradient rows
        loss, g, _, _ = propagator_sharded_energy_grad(eng, M_dev, u_dev, a.dt, psi0, device=dev)
        grad_dev.copy_(torch.from_numpy(g).to(dev))
        return loss, grad_dev

    # Which multi-GPU form runs.  Default = north_star's partitioning: every rank runs the whole chain on its own
    # batch shard, ONE all-reduce of (loss, dL/du) -- 'replicated'.  'sliced' / 'propagators' / 'auto' stay as options.  No
    # exception handling around anything that contains a collective: every rank issues the same collectives in the same order
    # with the same sizes, and a rank-local failure (OOM, DiffqcError) aborts the job non-zero instead of leaving the peers in
    # a mismatched collective.
    force = bool(os.environ.get("DQ_FORCE_COLLECTIVES")) and D.on
    multi = world > 1 or force
    both = multi and a.multi_gpu == "auto"
    mode = "sliced" if (a.multi_gpu == "sliced" and D.on) or both else ("propagators" if a.multi_gpu == "propagators" else "replicated")
    want_beside = multi and a.multi_gpu in ("replicated", "auto") and not a.no_beside
    forms = {}
    sliced_check = None
    if W > 0:
        uw = torch.from_numpy(u_all[:W].copy()).to(dev)
        lr, gr = run_replicated(uw, torch.zeros((W, n), dtype=torch.float64, device=dev))
        if mode == "sliced" or want_beside:
            # the warm-up doubles as the check: same loss and gradient from both forms on every rank
            ls, gs = run_sliced(uw, torch.zeros((W, n), dtype=torch.float64, device=dev))
            err = float((gs - gr).abs().max().item())
            ok = err <= 1e-9 * (1.0 + float(gr.abs().max().item())) and abs(ls - lr) <= 1e-9 * (1.0 + abs(lr))
            flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev)
            D.all_reduce_(flag, dist.ReduceOp.MIN)
            sliced_check = {"max_abs_grad_diff_vs_replicated": err, "abs_loss_diff_vs_replicated": abs(ls - lr),
                            "agreed_on_every_rank": bool(flag.item() >= 1.0)}
            if flag.item() < 1.0:
                if a.multi_gpu == "sliced":
                    raise SystemExit("sliced form disagrees with the replicated form: {}".format(sliced_check))
                mode, both = "replicated", False

    # beside the timed form (outside the timed region, max over ranks, one trial each on the same K-step chain): the other form,
    # and -- rank 0 alone -- the WHOLE global batch on one GPU, so that a reader sees what the scaling is worth
    if both or want_beside:
        for name, fn in (("sliced", run_sliced), ("replicated", run_replicated)):
            if name == "replicated" and not both:
                continue
            g2 = torch.zeros((K, n), dtype=torch.float64, device=dev)
            D.fence()
            t1 = time.perf_counter()
            l2, g2 = fn(u_t, g2)
            D.fence()
            forms[name] = {"seconds": D.max_over_ranks(time.perf_counter() - t1), "loss": l2, "grad": g2.clone()}
    if both:
        mode = "replicated" if forms["replicated"]["seconds"] < forms["sliced"]["seconds"] else "sliced"
    whole = None
    if want_beside and rank == 0:
        psi_all = torch.from_numpy(problems.random_states(d, B * world, a.seed, first=0)).to(dev) if world > 1 else psi0
        eng.set_option("plan_steps", max(a.chain, K))
        eng.set_option("cache_bytes", 0)
        g2 = torch.zeros((K, n), dtype=torch.float64, device=dev)
        eng.energy_grad(u_t[:1].contiguous(), a.dt, psi_all, want_final=False, want_energies=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        eng.energy_grad(u_t, a.dt, psi_all, want_final=False, want_energies=False, grad_out=g2)
        torch.cuda.synchronize(); whole = time.perf_counter() - t1
        del psi_all
    prop_check = None
    if mode == "propagators" and W > 0:   # the warm-up chain doubles as the check against the replicated form
        lp, gp = run_propagators(uw, torch.zeros((W, n), dtype=torch.float64, device=dev))
        prop_check = {"max_abs_grad_diff_vs_replicated": float((gp - gr).abs().max().item()), "abs_loss_diff_vs_replicated": abs(lp - lr)}
    run = {"sliced": run_sliced, "propagators": run_propagators}.get(mode, run_replicated)
    eng.profile_enable(True)
    eng.profile_reset()

    # ---- the timed region: EXACTLY K steps between two fences, `trials` times; the line reports the median trial ----------
    times = []
    for _ in range(max(1, a.trials)):
        D.fence()
        t0 = time.perf_counter()
        loss, grad = run(u_t, grad_t)
        D.fence()
        times.append(D.max_over_ranks(time.perf_counter() - t0))
    elapsed = float(np.median(times))
    elapsed_all = float(np.sum(times))
    prof = eng.profile()
    eng.profile_enable(False)
    info = {k: eng.info(k) for k in INFO_KEYS}
    info["batch"], info["chain_steps"] = B, K
    opback = bool(eng.info("operator_backward"))
    forms_out = None
    if forms:
        forms_out = {name: {"value": world * K / f["seconds"], "ms_per_step": f["seconds"] / max(K, 1) * 1e3, "loss": f["loss"]}
                     for name, f in forms.items()}
        if "replicated" in forms:
            forms_out["max_abs_grad_diff_sliced_vs_replicated"] = float((forms["sliced"]["grad"] - forms["replicated"]["grad"]).abs().max().item())
            forms_out["abs_loss_diff_sliced_vs_replicated"] = abs(forms["sliced"]["loss"] - forms["replicated"]["loss"])
        forms_out["note"] = "one untimed trial of each form on the same K-step chain (max over ranks); the timed region ran '{}'".format(mode)

    if rank != 0:
        eng.close()
        return
    structure = None
    if world == 1 and not D.on and a.workload == "ising" and info["real_path"] > 0 and not a.skip_general_path:
        structure = by_input_structure(eng, a, dev, u_all, psi0, grad, loss)
    blocks = roofline_blocks(prof, info, d)
    rk = ranked(blocks)
    steps_per_s = world * K / elapsed
    W_alg = 8.0 * d * d * B  # one complex (d x d).(d x B) product
    ex_tf = executed_tflops(blocks, elapsed_all)
    structure_note = {
        "ising": ("diagonal Ising drift + X_k controls: a sum of Pauli strings -- the engine keeps every state column in LDS for the whole chain "
                  "and builds no propagator (input-structure-specific path; by_input_structure has the same inputs on the explicit-"
                  "propagator GEMM chain and dense inputs of the same size)") if info.get("state_columns", 0) > 0 else
                 ("diagonal Ising drift + X_k controls: real, sparse and a sum of Pauli strings -- the engine builds the step exponential column "
                  "by column in LDS (input-structure-specific path; see by_input_structure for the same inputs on the general sparse kernels "
                  "and for dense inputs of the same size)") if info.get("xor_regular", 0) > 0 else
                 ("diagonal Ising drift + X_k controls: real AND sparse -- the engine runs the step exponential as sparse x dense passes "
                  "(input-structure-specific path; see by_input_structure for dense inputs of the same size)"),
        "dense-real": "dense random real symmetric H0 / H_k: cos X + i sin X on real symmetric-output MFMA products",
        "dense-complex": "dense random complex Hermitian H0 / H_k: the general complex tape"}[a.workload]
    beside = None
    if want_beside:
        # [aggregate steps/s in the headline's unit (one step = fwd+adjoint over a 4096-state batch), ms per chain step]
        beside = {}
        if "sliced" in forms:
            beside["sliced_form"] = [world * K / forms["sliced"]["seconds"], forms["sliced"]["seconds"] / K * 1e3]
        if whole is not None:
            beside["one_gpu_whole_global_batch"] = [world * K / whole, whole / K * 1e3]
    out = {
        "metric": "Trotter steps/sec (fwd+adjoint), n={} qubits batch={}".format(n, B),
        "value": steps_per_s,
        "unit": "Trotter steps/s (one step = fwd+adjoint over one GPU's {}-state batch; sum over GPUs)".format(B),
        "n_gpus": world,
        "steps": K,
        "warmup": W,
        "ms_per_step": elapsed / max(K, 1) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "backend": D.name(),
        "trials": {"n": len(times), "ms_per_step_median": elapsed / max(K, 1) * 1e3,
                   "ms_per_step_min": min(times) / max(K, 1) * 1e3, "ms_per_step_max": max(times) / max(K, 1) * 1e3},
        "config": {
            "workload": "BASELINE configs[{}]: {}-qubit {} (d={}), {} controls, B={} states per GPU, fwd+adjoint, fp64 complex, "
                        "{} of {} Trotter steps timed".format(2 if world == 1 else 3, n,
                                                              {"ising": "random Ising H + X_k", "dense-real": "dense random real symmetric H",
                                                               "dense-complex": "dense random complex Hermitian H"}[a.workload], d, n, B, K,
                                                              max(a.chain, K)),
            "workload_detail": "stored dense; dt={}; batch >= dim: operator-form backward sweep, no state checkpoints, per-step cost "
                               "independent of the chain length; otherwise the slice uses the checkpoint stride of the full chain".format(a.dt),
            "input_structure": structure_note,
            "n_qubits": n, "dim": d, "n_controls": n, "batch_per_gpu": B, "global_batch": B * world,
            "chain_steps": K, "dt": a.dt,
            "parallelism": "dp{}".format(world),
            "parallelism_detail": ("dp{} (batch shards, one all-reduce of loss+dL/du)".format(world) if mode == "replicated" else
                                   "dp{0} over the batch, the propagators of every window of {0} steps built one per rank and all-gathered; "
                                   "state-form chain, reduce-scatter of the propagators' cotangents (SURVEY 8-F2 as written)".format(world)
                                   if mode == "propagators" else
                                   "dp{0} over the batch x {0} time slices: operator form, each rank builds and sweeps 1/{0} of "
                                   "the chain for the whole batch; all-gather of the slice products, all-reduce of W_S "
                                   "(d x d) and of dL/du".format(world)),
            "multi_gpu_form": mode, "sliced_warmup_check": sliced_check, "propagators_warmup_check": prop_check,
            "multi_gpu_forms": forms_out,
            "collectives_forced_at_world_1": force and world == 1,
            "expm": describe_expm(info),
            "path": ("state-column form (Pauli-string H)" if info.get("state_columns", 0) > 0 else
                     "real-symmetric fast path (H real)" if info["real_path"] > 0 else "general complex path"),
            "state_columns": bool(info.get("state_columns", 0) > 0),
            "checkpoint_stride": int(info["checkpoint_stride"]), "operator_backward": opback,
            "trajectory_steps_per_s": steps_per_s * B,
            "loss": loss, "grad_norm": float(torch.linalg.norm(grad).item()) if hasattr(grad, "norm") else float(np.linalg.norm(grad)),
        },
        "roofline": rk[0] if rk else None,
        "roofline_other": rk[1] if len(rk) > 1 else None,
        "roofline_all_classes": {k: {kk: b[kk] for kk in ("frac", "achieved", "unit", "launches", "ms_per_launch", "share_of_kernel_time")}
                                 for k, b in blocks.items() if b["launches"] > 0},
        "beside": beside,
        "by_input_structure": structure,
        "end_to_end": {
            "algorithmic_flops_per_step": 3 * W_alg,
            "algorithmic_tflops": 3 * W_alg * world * K / elapsed / 1e12 / world,
            "algorithmic_frac_of_peak": 3 * W_alg * K / elapsed / 1e12 / PEAK_F64_MFMA_TFLOPS,
            "executed_mfma_tflops": ex_tf, "executed_mfma_frac_of_peak": ex_tf / PEAK_F64_MFMA_TFLOPS,
            "note": "algorithmic = SURVEY 8-D4's 3 x 8 d^2 B flops per fwd+adjoint step (state apply, costate apply, outer product) x steps/s per "
                    "GPU -- a work-normalised rate, NOT a fraction of what the chip executed: the sparse-generator scheme runs the exponential "
                    "on HBM-bound passes and the 3M product executes 6 of every 8 flops.  executed_mfma_tflops = the flops the matrix "
                    "pipe actually executed in the timed trials (rank 0, from the per-launch events) / their wall time",
            "kernel_ms": {k: v["ms"] for k, v in prof.items()},
            "kernel_launches": {k: v["launches"] for k, v in prof.items()},
            "hbm_GBps": {k: (v["bytes"] / v["ms"] * 1e-6 if v["ms"] > 0 else 0.0) for k, v in prof.items()
                         if k not in ("zgemm", "rgemm", "rsym", "zherm")},
        },
    }
    if world == 1 and not D.on and not a.skip_other_configs:
        oc = other_configs(dev, a.seed, (n, a.dt, u_all[:K].copy(), psi0, loss) if a.workload == "ising" else None)
        for v in oc.values():
            if isinstance(v, dict):
                v.pop("_prof", None); v.pop("_info", None)
        if not a.skip_full_chain and n == 12:
            oc["full_chain_S{}".format(a.chain)] = full_chain(eng, a, dev, psi0)
        out["other_configs"] = oc
    if world == 1 and not D.on and not a.skip_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(n, B, a.seed, a.dt, u_all[0])
        except Exception as e:  # the baseline is a reported extra; never lose the GPU line over it
            out["cpu_baseline"] = {"value": None, "unit": "Trotter steps/s (fwd+adjoint)", "cores": os.cpu_count(),
                                   "kind": "port", "sample": "failed: {}".format(e)}
    emit(out)
    eng.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed Trotter steps (default 20; config 2: 1000)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=3, choices=(2, 3, 5),
                    help="3 = headline (BASELINE configs[2]/[3]); 2 = configs[1] (n=8 QAOA forward); 5 = configs[4] (n=18, c64, matrix-free)")
    ap.add_argument("--workload", choices=("ising", "dense-real", "dense-complex"), default="ising",
                    help="config 3 inputs: BASELINE's random Ising + X_k (default) or dense random matrices of the same size and norm")
    ap.add_argument("--qubits", type=int, default=12)
    ap.add_argument("--batch", type=int, default=None, help="states per GPU (config 5: the WHOLE batch, default 512)")
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-other-configs", action="store_true")
    ap.add_argument("--skip-full-chain", action="store_true", help="do not run the full S = 2000 chain extra (about a minute)")
    ap.add_argument("--skip-general-path", action="store_true",
                    help="do not run the by_input_structure extras (profiling: only the headline path's kernels)")
    ap.add_argument("--chain", type=int, default=2000,
                    help="length of the full configuration's chain: the timed slice uses ITS checkpoint stride")
    ap.add_argument("--real-path", type=int, default=-1,
                    help="-1 auto (real H -> cos/sin form on real MFMA products), 0 = general complex path only")
    ap.add_argument("--multi-gpu", choices=("auto", "sliced", "replicated", "propagators"), default="replicated",
                    help="N > 1: 'replicated' (default; north_star / SURVEY 8-E1) = every rank runs the whole chain on its batch shard, one "
                         "all-reduce of loss + dL/du; 'sliced' = chain split over time across the ranks (operator form; three collectives); "
                         "auto = both checked and tried, the faster one timed; 'propagators' = SURVEY 8-F2 as written: state-form "
                         "chain on propagators built 1/N per rank and all-gathered (never chosen by auto: 0.8 GB of collectives per step)")
    ap.add_argument("--no-beside", action="store_true",
                    help="N > 1: do not measure, beside the timed form, the sliced form and the whole global batch on one GPU")
    ap.add_argument("--trials", type=int, default=3, help="how many times the K-step region is timed (the line reports the median)")
    ap.add_argument("--extras", default=None, help="where the full record goes (default: bench_extras.json beside bench.py)")
    ap.add_argument("--launch", action="store_true", help="start the ranks through torch.distributed.run even for --gpus 1")
    a = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or a.launch):
        sys.exit(launch_ranks(a, sys.argv[1:]))
    claim_stdout()
    if a.extras:
        global _EXTRAS_PATH
        _EXTRAS_PATH = os.path.abspath(a.extras)
    D = Dist(a)
    {2: run_config2, 3: run_config3, 5: run_config5}[a.config](a, D)
    D.close()  # (only on success: a failing rank must not enter a barrier its peers will never reach -- it exits non-zero)


if __name__ == "__main__":
    main()
