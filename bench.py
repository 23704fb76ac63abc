#!/usr/bin/env python3
"""Headline benchmark: Trotter steps/s (forward + adjoint gradient), n = 12 qubits, batch = 4096 per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is ONE Trotter step of the fwd+adjoint pass over one GPU's whole batch (BASELINE.json configs[2]:
12-qubit random Ising H stored dense (d = 4096), K = 12 controls X_k, B = 4096 states, fp64 complex).  The timed
region is one `dq_energy_grad` call over a chain of K steps -- K forward steps (assembly, exp, U.Psi), then the K-step
adjoint sweep (rebuild U_t, Psi_t, costate, G_t, reverse-mode exp, control-gradient contractions) -- plus, for N > 1,
the single RCCL all-reduce of (loss, dL/du).  K steps of the 2000-step configuration are timed. At this shape
(batch >= dim) the backward sweep runs in operator form and keeps no state checkpoints, so the per-step cost does not depend
on the chain length; for narrower batches `plan_steps` gives the slice the checkpoint stride of the FULL chain, so that it
pays the same share of psi_t rebuilds per step as the whole run (the full 2000-step run itself is in DESIGN.md).  Inputs (H0, H_k, M, Psi0, pulse
table) are resident in HBM before the clock starts.  N GPUs = N independent batch shards of 4096 (weak scaling):
value = N*K / max-over-ranks wall time.

The Ising Hamiltonian is real, so the engine takes its real-symmetric fast path by default (exp(-i dt H) = cos X + i sin X
with real symmetric-output MFMA products; same results to rounding); `general_complex_path` in the JSON
line is the same chain with that path switched off (what an arbitrary complex Hermitian H of this size costs).

Adds to the JSON line: `roofline` for the dominant kernel = the kernel class with the largest share of the timed region,
decided live from the per-launch HIP events the library records on the launch stream (on the real path: the real
symmetric-output GEMM of the exp / reverse-mode products; on the general complex path: the batched complex GEMM),
`roofline_other` for the second MFMA class, and `cpu_baseline` (the oracle = the reference's scipy expm + matmul step,
timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TRAFFIC_BYTES_PER_LAUNCH = 3.51e9  # HBM-side bytes per full-output ZGEMM launch from rocprofv3 PMC (profiles/r1_bench_s3_final_summary.txt):
                                   # 2*FETCH_SIZE (gfx950 correction) + WRITE_SIZE = 3.23 GB + 0.28 GB, d = B = 4096
TRAFFIC_RGEMM = {"rgemm": 1.81e9, "rsym": 2.51e9}  # real GEMM: full-output (profiles/r1_bench_v5_realpath_summary.txt);
                                                   # symmetric-output launches (average over the one- and two-product forms of
                                                   # the headline step: 8 + 4 per step) + fix-up pass, LDS-DMA staged
                                                   # (profiles/r1_bench_s3_sparse_summary.txt: 2445 + 64 MB)
TRAFFIC_SPMM = 8.96e8  # fabric-side bytes per sparse x dense launch (profiles/r1_bench_s3_spx_summary.txt: rsp_mm 831 MB, rsp_mm_dot
                       # 1027 MB, 2 : 1 in the step; 700-900 MB read + 134 MB
                       # written for 134 MB in, 134 MB out: a partner row per nonzero of X, most of them L2 hits)
PEAK_F64_MFMA_TFLOPS = 78.6  # v_mfma_f64_16x16x4_f64: 64 cycles/instr/SIMD measured (tools/mfma_f64_probe.hip)
                             # = 32 flop/clk/SIMD * 1024 SIMDs * 2.4 GHz; equals the datasheet FP64-matrix figure


def cpu_baseline(n, B, seed, dt, u_row):
    """One fwd+adjoint Trotter step of the SAME workload on the host cores, through the oracle: the reference's
    per-step calls (scipy.linalg.expm + np.matmul, sim_plain.py:145-146) for the forward half, and the exact
    adjoint with ONE scipy.linalg.expm_frechet per step (oracle.energy_grad_one_frechet) for the gradient half."""
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
    psi1 = orc.propagate(H0, list(Hs), u_row[None, :], dt, psi)
    t_fwd = time.perf_counter() - t0
    t0 = time.perf_counter()
    loss, grad, _ = orc.energy_grad_one_frechet(H0, list(Hs), M, u_row[None, :], dt, psi)
    t_all = time.perf_counter() - t0
    return {"value": 1.0 / t_all, "unit": "Trotter steps/s (fwd+adjoint)", "cores": int(cores), "kind": "port",
            "forward_only_steps_per_s": 1.0 / t_fwd,
            "sample": "1 fwd+adjoint Trotter step of the same workload (n={}, d={}, B={}) through the oracle: dH assembly, "
                      "scipy.linalg.expm, np.matmul, one scipy.linalg.expm_frechet, 12 contractions: {:.1f} s "
                      "(forward-only step: {:.1f} s)".format(n, 1 << n, B, t_all, t_fwd)}


def other_configs(dev, seed, headline=None):
    """Short measurements of the other BASELINE configs on this GPU (reported as extras, outside the timed region)."""
    import torch
    from diffquantum_amd import problems
    from diffquantum_amd.engine import Engine
    from diffquantum_amd.structured import StructuredEngine, ising_x_terms
    out = {}
    # configs[1]: 8-qubit ring QAOA, 1000 Trotter steps, batch 1024, fp64, forward only
    p = problems.qaoa_ring(8)
    S, B = 1000, 1024
    u = torch.from_numpy(problems.pulse_table(16, S, 1.0, seed=seed)).to(dev)
    psi = torch.from_numpy(problems.random_states(256, B, seed)).to(dev)
    with Engine(256, 16, device=dev.index or 0) as e:
        e.use_torch_stream()
        e.set_hamiltonians(p["H0"].astype(np.complex128), p["Hs"].astype(np.complex128))
        e.propagate(u, 1.0 / S, psi)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fin = e.propagate(u, 1.0 / S, psi)
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        out["config2_qaoa_n8_S1000_B1024_fwd"] = {
            "steps_per_s": S / el, "wall_s": el, "operator_form": bool(e.info("composed")),
            "max_norm_drift": float((torch.linalg.norm(fin, dim=0) - 1).abs().max().item())}
    # configs[4] per-GPU share: 18-qubit Ising + X controls as Pauli strings, complex64, 512/8 = 64 states per GPU
    n, B, S = 18, 64, 20
    rng = np.random.default_rng(seed)
    st = rng.standard_normal((1 << n, B)) + 1j * rng.standard_normal((1 << n, B))
    st = torch.from_numpy((st / np.linalg.norm(st, axis=0)).astype(np.complex64)).to(dev)
    us = problems.pulse_table(n, S, S * 1e-3, seed=seed)
    with StructuredEngine(n, n, device=dev.index or 0, dtype="c64") as e:
        e.use_torch_stream()
        e.set_terms(ising_x_terms(n, seed))
        e.propagate(us[:2], 1e-3, st)
        e.set_option("profile", -1); e.set_option("profile", 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.propagate(us, 1e-3, st)
        torch.cuda.synchronize(); el_f = time.perf_counter() - t0
        ms, by = e.info("prof_ms"), e.info("prof_bytes")
        e.set_option("profile", 0)
        e.energy_grad(us[:2], 1e-3, st, want_final=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        e.energy_grad(us, 1e-3, st, want_final=False)
        torch.cuda.synchronize(); el_g = time.perf_counter() - t0
        out["config5_pergpu_n18_B64_c64_structured"] = {
            "fwd_steps_per_s": S / el_f, "fwd_adjoint_steps_per_s": S / el_g, "taylor_degree": int(e.info("degree")),
            "roofline": {"bound": "hbm", "achieved": by / ms * 1e-6 if ms > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
                         "frac": (by / ms * 1e-6 / 8000.0) if ms > 0 else 0.0,
                         "note": "forward H.psi passes: algorithmic bytes (state-units x 2^n x B x 8 B) / HIP-event time"}}
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--qubits", type=int, default=12)
    ap.add_argument("--batch", type=int, default=4096, help="states per GPU")
    ap.add_argument("--dt", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--skip-other-configs", action="store_true")
    ap.add_argument("--skip-general-path", action="store_true",
                    help="do not run the general-complex-path extra (profiling: only the headline path's kernels)")
    ap.add_argument("--chain", type=int, default=2000,
                    help="length of the full configuration's chain: the timed slice uses ITS checkpoint stride")
    ap.add_argument("--real-path", type=int, default=-1,
                    help="-1 auto (real H -> cos/sin form on real MFMA products), 0 = general complex path only")
    ap.add_argument("--multi-gpu", choices=("auto", "sliced", "replicated"), default="auto",
                    help="N > 1: 'sliced' = chain split over time across the ranks (operator form; three collectives), "
                         "'replicated' = every rank runs the whole chain on its batch shard (one all-reduce); "
                         "auto = sliced if its warm-up gradient equals the replicated one on every rank")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from diffquantum_amd import problems
    from diffquantum_amd.distributed import allreduce_loss_grad, time_sliced_energy_grad
    from diffquantum_amd.engine import Engine

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and rank == 0:
        print("warning: --gpus {} but WORLD_SIZE {}".format(a.gpus, world), file=sys.stderr)
    # one process per GPU. (Rehearsal on a box with fewer GPUs than ranks: DQ_DIST_BACKEND=gloo shares the devices
    # round-robin and stages the collectives through the host -- RCCL refuses two ranks on one device.)
    backend = os.environ.get("DQ_DIST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ or "MASTER_PORT" in os.environ
    if use_dist:  # launched by torch.distributed.run (also exercised at world size 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        import datetime
        tmo = datetime.timedelta(minutes=5)  # a mismatched collective fails the run instead of hanging it
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, timeout=tmo)

    def all_reduce_(t, op):  # in place on a device tensor, whatever the backend
        if backend == "nccl":
            dist.all_reduce(t, op=op)
        else:
            h = t.cpu()
            dist.all_reduce(h, op=op)
            t.copy_(h)

    n, B, K, W = a.qubits, a.batch, a.steps, a.warmup
    d = 1 << n
    # ---- inputs, resident in HBM before the clock starts --------------------------------------
    H0, Hs, M = problems.random_ising_torch(n, a.seed, dev)
    u_all = problems.pulse_table(n, max(K, W, 1), a.dt * max(K, W, 1), seed=a.seed)
    psi0 = torch.from_numpy(problems.random_states(d, B, a.seed, first=rank * B)).to(dev)
    eng = Engine(d, n, device=local)
    eng.use_torch_stream()
    eng.set_hamiltonians(H0, Hs)
    eng.set_observable(M)
    eng.set_option("real_path", a.real_path)
    eng.set_option("plan_steps", max(a.chain, K))  # checkpoint/rebuild ratio of the full 2000-step chain
    u_t = torch.from_numpy(u_all[:K].copy()).to(dev)
    grad_t = torch.zeros((K, n), dtype=torch.float64, device=dev)

    # Propagator cache of the sliced form: a rank's slice of the planned chain is short enough to be cached WHOLE
    # ((q + 3) d^2 fp64 words per step, 2q + 2 with the sparse-generator scheme: 0.8-1.07 GB per step at q = 3), so the budget is
    # what the slice needs, capped at 72 % of the HBM (the engine's auto mode stops at 96 GiB: bigger first-use allocations cost
    # seconds, which a 2000-step single-GPU chain does not win back but a fully cached slice does, evaluation after evaluation)
    slice_steps = (max(a.chain, K) + world - 1) // world
    sliced_cache_bytes = min(8.0 * d * d * 8 * slice_steps + (1 << 20), 0.72 * torch.cuda.get_device_properties(local).total_memory)

    def run_replicated(u_dev, grad_dev):
        eng.set_option("plan_steps", max(a.chain, K))
        eng.set_option("cache_bytes", 0)
        loss, g, _, _ = eng.energy_grad(u_dev, a.dt, psi0, want_final=False, want_energies=False, grad_out=grad_dev)
        return allreduce_loss_grad(loss, g, device=dev)

    def run_sliced(u_dev, grad_dev):
        # chain split over TIME across the ranks: each rank builds and sweeps 1/world of the steps for the
        # WHOLE batch (its own shard enters through W_S only); all-gather of the slice products, all-reduce of W_S,
        # all-reduce of the gradient rows + loss
        eng.set_option("plan_steps", (max(a.chain, K) + world - 1) // world)
        eng.set_option("cache_bytes", sliced_cache_bytes)
        loss, g, _, _ = time_sliced_energy_grad(eng, u_dev, a.dt, psi0, device=dev)
        g = torch.from_numpy(g).to(dev)
        grad_dev.copy_(g)
        return loss, grad_dev

    mode = "replicated"
    sliced_check = None
    if (world > 1 and a.multi_gpu != "replicated") or (use_dist and a.multi_gpu == "sliced"):
        mode = "sliced"  # (asked for explicitly it also runs at world size 1: rehearsal of the collectives)
    if W > 0:
        uw = torch.from_numpy(u_all[:W].copy()).to(dev)
        lr, gr = run_replicated(uw, torch.zeros((W, n), dtype=torch.float64, device=dev))
        if mode == "sliced":
            # the warm-up doubles as the check: same loss and gradient from both forms on every rank, or fall back
            ok = 1.0
            try:
                ls, gs = run_sliced(uw, torch.zeros((W, n), dtype=torch.float64, device=dev))
                err = float((gs - gr).abs().max().item()) if W > 0 else 0.0
                sliced_check = {"max_abs_grad_diff_vs_replicated": err, "abs_loss_diff_vs_replicated": abs(ls - lr)}
                if not (err <= 1e-9 * (1.0 + float(gr.abs().max().item())) and abs(ls - lr) <= 1e-9 * (1.0 + abs(lr))):
                    ok = 0.0
            except Exception as e:  # noqa: BLE001 -- any failure of the sliced form means: use the replicated one
                sliced_check = {"error": repr(e)}
                ok = 0.0
            flag = torch.tensor([ok], dtype=torch.float64, device=dev)
            all_reduce_(flag, dist.ReduceOp.MIN)
            if float(flag.item()) < 1.0 and a.multi_gpu == "auto":
                mode = "replicated"
    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # auto: which form is faster on THIS node is measured, not assumed -- both once on the K-step chain, outside the
    # timed region, max over ranks; the timed region then runs the faster one and the other is reported beside it
    trial = {}
    if mode == "sliced" and a.multi_gpu == "auto" and world > 1:
        for name, fn in (("sliced", run_sliced), ("replicated", run_replicated)):
            try:
                g2 = torch.zeros((K, n), dtype=torch.float64, device=dev)
                fence()
                t1 = time.perf_counter()
                l2, g2 = fn(u_t, g2)
                fence()
                tm = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
                all_reduce_(tm, dist.ReduceOp.MAX)
                trial[name] = (float(tm.item()), l2, g2.clone())
            except Exception as e:  # noqa: BLE001
                trial[name] = (float("inf"), repr(e), None)
        if trial["replicated"][0] < trial["sliced"][0]:
            mode = "replicated"
    run = run_sliced if mode == "sliced" else run_replicated
    eng.profile_enable(True)
    eng.profile_reset()

    fence()
    t0 = time.perf_counter()
    loss, grad = run(u_t, grad_t)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        all_reduce_(tmax, dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    prof = eng.profile()
    eng.profile_enable(False)
    other_mode = None
    if trial:
        oname = "replicated" if mode == "sliced" else "sliced"
        t_o, l_o, g_o = trial[oname]
        if g_o is None:
            other_mode = {"form": oname, "error": l_o}
        else:
            other_mode = {"form": oname, "value": world * K / t_o, "ms_per_step": t_o / max(K, 1) * 1e3,
                          "max_abs_grad_diff_vs_headline_form": float((g_o - grad).abs().max().item()),
                          "abs_loss_diff_vs_headline_form": abs(l_o - loss),
                          "trial_of_the_headline_form_ms_per_step": trial[mode][0] / max(K, 1) * 1e3}
    info = {k: eng.info(k) for k in ("degree", "squarings", "products", "theta", "checkpoint_stride", "time_chunk",
                                     "zgemm_alg", "real_path", "real_hamiltonians", "sparse_controls", "sparse_generator",
                                     "control_nonzeros")}
    general = None
    if rank == 0 and world == 1 and info["real_path"] > 0 and not a.skip_general_path:
        # the same chain on the general complex path (any Hermitian H), outside the timed region, short chain
        Kg = min(K, 4)
        eng.profile_enable(False)
        eng.set_option("real_path", 0)
        ug = torch.from_numpy(u_all[:Kg].copy()).to(dev)
        gg = torch.zeros((Kg, n), dtype=torch.float64, device=dev)
        eng.energy_grad(ug[:1], a.dt, psi0, want_final=False, want_energies=False)
        torch.cuda.synchronize(); tg = time.perf_counter()
        lg, _, _, _ = eng.energy_grad(ug, a.dt, psi0, want_final=False, want_energies=False, grad_out=gg)
        torch.cuda.synchronize(); tg = time.perf_counter() - tg
        eng.set_option("real_path", a.real_path)
        gr = torch.zeros((Kg, n), dtype=torch.float64, device=dev)
        lr, _, _, _ = eng.energy_grad(ug, a.dt, psi0, want_final=False, want_energies=False, grad_out=gr)
        schemes = {}
        for name, opts in (("sparse_controls_dense_drift", {"sparse_generator": 0}),
                           ("dense_real_path", {"sparse_generator": 0, "sparse_controls": 0})):
            # the same chain with the structure of the inputs ignored step by step: what a DENSE drift (then dense controls)
            # of this size costs on the real path
            for k_, v_ in opts.items():
                eng.set_option(k_, v_)
            gs_ = torch.zeros((Kg, n), dtype=torch.float64, device=dev)
            eng.energy_grad(ug[:1], a.dt, psi0, want_final=False, want_energies=False)
            torch.cuda.synchronize(); ts_ = time.perf_counter()
            ls_, _, _, _ = eng.energy_grad(ug, a.dt, psi0, want_final=False, want_energies=False, grad_out=gs_)
            torch.cuda.synchronize(); ts_ = time.perf_counter() - ts_
            schemes[name] = {"steps_per_s": Kg / ts_, "max_abs_grad_diff_vs_headline_scheme": float((gs_ - gr).abs().max().item()),
                             "abs_loss_diff_vs_headline_scheme": abs(ls_ - lr)}
        eng.set_option("sparse_generator", -1)
        eng.set_option("sparse_controls", 1)
        general = {"steps_per_s": Kg / tg, "chain_steps": Kg, "products_per_step": "11 ZGEMMs (3M) of 8 d^3 + 3 (anti-)Hermitian-output ZGEMMs (half the tiles)",
                   "other_real_schemes": schemes,
                   "max_abs_grad_diff_vs_real_path": float((gg - gr).abs().max().item()),
                   "abs_loss_diff_vs_real_path": abs(lg - lr),
                   "note": "same inputs with real_path = 0: what a complex Hermitian H of this size costs; the two paths' "
                           "gradients on the same {}-step chain differ by the amount shown".format(Kg)}

    if rank == 0:
        z = prof["zgemm"]
        rg, rs = prof["rgemm"], prof["rsym"]

        def rate(p):
            return p["flops"] / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0
        per_launch_flops = z["flops"] / max(1, z["launches"])
        per_launch_ms = z["ms"] / max(1, z["launches"])
        achieved = z["flops"] / (z["ms"] * 1e-3) / 1e12 if z["ms"] > 0 else 0.0
        steps_per_s = world * K / elapsed
        W_alg = 8.0 * d * d * B  # one complex (d x d).(d x B) product
        out = {
            "metric": "Trotter steps/sec (fwd+adjoint), n={} qubits batch={}".format(n, B),
            "value": steps_per_s,
            "unit": "Trotter steps/s (each step = fwd+adjoint over one GPU's {}-state batch; aggregate over GPUs)".format(B),
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / max(K, 1) * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: {}-qubit random Ising H stored dense (d={}), {} controls X_k, "
                            "B={} states per GPU, fwd + adjoint gradient, fp64 complex, dt={}; timed chain = {} of the "
                            "{} Trotter steps (batch >= dim: the backward sweep runs in operator form without state checkpoints, "
                            "so the per-step cost is chain-length independent; otherwise the slice uses the checkpoint "
                            "stride of the full chain)".format(n, d, n, B, a.dt, K, max(a.chain, K)),
                "n_qubits": n, "dim": d, "n_controls": n, "batch_per_gpu": B, "global_batch": B * world,
                "chain_steps": K, "dt": a.dt,
                "parallelism": ("dp{} (batch shards, one all-reduce of loss+dL/du)".format(world) if mode == "replicated" else
                                "dp{0} over the batch x {0} time slices: operator form, each rank builds and sweeps 1/{0} of "
                                "the chain for the whole batch; all-gather of the slice products, all-reduce of W_S "
                                "(d x d) and of dL/du".format(world)),
                "multi_gpu_form": mode, "sliced_warmup_check": sliced_check, "other_multi_gpu_form": other_mode,
                "expm": ("cos X + i sin X, Taylor degree {0}: drift and controls sparse, so X_t is sparse -- the powers X^2 .. X^{0} by "
                         "{1} sparse x dense passes, cos/sin as linear combinations of them (no MFMA product in the step "
                         "exponential), theta<={2:.3g}".format(int(info["degree"]), int(info["products"]), info["theta"])
                         if info["sparse_generator"] > 0 else
                         "cos X + i sin X, Taylor degree {} ({} real symmetric-output products{}), theta<={:.3g}".format(
                    int(info["degree"]), int(info["products"]),
                    "; sparse controls: scatter assembly, last reverse-mode level as row dot products" if info["sparse_controls"] > 0 else "",
                    info["theta"]) if info["real_path"] > 0 else
                    "Taylor degree {} ({} complex products, {} squarings), theta<={:.3g}".format(
                        int(info["degree"]), int(info["products"]), int(info["squarings"]), info["theta"])),
                "path": "real-symmetric fast path (H real)" if info["real_path"] > 0 else "general complex path",
                "checkpoint_stride": int(info["checkpoint_stride"]), "operator_backward": bool(eng.info("operator_backward")),
                "trajectory_steps_per_s": steps_per_s * B,
                "loss": loss, "grad_norm": float(torch.linalg.norm(grad).item()) if hasattr(grad, "norm") else float(np.linalg.norm(grad)),
            },
            "roofline": None,  # filled below: the kernel class with the largest share of the timed region
            "roofline_other": None,
            "general_complex_path": general,
            "end_to_end": {
                "algorithmic_flops_per_step": 3 * W_alg,
                "algorithmic_tflops": 3 * W_alg * K / elapsed / 1e12,
                "frac_of_peak": 3 * W_alg * K / elapsed / 1e12 / PEAK_F64_MFMA_TFLOPS,
                "zgemm_flops_per_step": z["flops"] / max(K, 1),
                "rgemm_flops_per_step": (rg["flops"] + rs["flops"]) / max(K, 1),
                "kernel_ms": {k: v["ms"] for k, v in prof.items()},
                "kernel_launches": {k: v["launches"] for k, v in prof.items()},
                "hbm_GBps": {k: (v["bytes"] / v["ms"] * 1e-6 if v["ms"] > 0 else 0.0) for k, v in prof.items()
                             if k not in ("zgemm", "rgemm", "rsym")},
            },
        }
        zg = {
            "bound": "mfma",
            "kernel": "zgemm_kernel<128,128,8,2,4,*,*,{}> (batched complex GEMM, v_mfma_f64_16x16x4_f64, {})".format(
                int(info["zgemm_alg"]), "3M: 3 real MFMAs per complex tile step" if info["zgemm_alg"] else "4M"),
            "achieved": achieved, "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": achieved / PEAK_F64_MFMA_TFLOPS,
            "traffic": TRAFFIC_BYTES_PER_LAUNCH,
            "launches": z["launches"], "flops_per_launch": per_launch_flops, "ms_per_launch": per_launch_ms,
            "ms_in_timed_region": z["ms"],
            "mfma_executed_tflops": achieved * (0.75 if info["zgemm_alg"] else 1.0),
            "frac_executed": achieved * (0.75 if info["zgemm_alg"] else 1.0) / PEAK_F64_MFMA_TFLOPS,
            "note": "achieved = sum over the timed region's full-output ZGEMM launches of the ALGORITHMIC 8*M*N*K flops / sum of "
                    "their HIP-event durations (rank 0). With the 3M complex product the kernel executes 6*M*N*K MFMA "
                    "flops for 8*M*N*K algorithmic ones, so `frac` can exceed 1; `frac_executed` is the matrix-pipe "
                    "utilisation. traffic = PMC FETCH/WRITE per launch (profiles/, separate --pmc passes)",
        }
        rsy = {
            "bound": "mfma",
            "kernel": "rgemm_kernel<128,128,16,4,4,sym> (batched real GEMM, v_mfma_f64_16x16x4_f64, LDS-DMA staged, "
                      "symmetric-output form: tiles on/above the diagonal + mirror; one- and two-product launches)",
            "achieved": rate(rs), "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": rate(rs) / PEAK_F64_MFMA_TFLOPS,
            "traffic": TRAFFIC_RGEMM["rsym"],
            "launches": rs["launches"], "flops_per_launch": rs["flops"] / max(1, rs["launches"]),
            "ms_per_launch": rs["ms"] / max(1, rs["launches"]), "ms_in_timed_region": rs["ms"],
            "note": "achieved = ALGORITHMIC flops of the symmetric-output products (2*K*128*128 per tile on/above the "
                    "diagonal: the product of two commuting symmetric matrices is symmetric, the mirror half is not "
                    "computed) / sum of their HIP-event durations (rank 0); equal to the executed MFMA flops. traffic = PMC "
                    "FETCH/WRITE per launch incl. the fix-up pass (profiles/, separate --pmc passes)",
            "full_output_form": {"achieved": rate(rg), "frac": rate(rg) / PEAK_F64_MFMA_TFLOPS, "launches": rg["launches"],
                                 "ms_per_launch": rg["ms"] / max(1, rg["launches"]), "traffic": TRAFFIC_RGEMM["rgemm"]},
        }
        sp = prof["spmm"]
        spm = {
            "bound": "hbm",
            "kernel": "rsp_mm_kernel (sparse generator x dense matrix: out[i,:] = sum over nnz(X[i,:]) of w X_m[j,:]; one wave per row)",
            "achieved": sp["bytes"] / sp["ms"] * 1e-6 if sp["ms"] > 0 else 0.0, "peak": 8000.0, "unit": "GB/s",
            "frac": (sp["bytes"] / sp["ms"] * 1e-6 / 8000.0) if sp["ms"] > 0 else 0.0,
            "traffic": TRAFFIC_SPMM, "launches": sp["launches"], "ms_per_launch": sp["ms"] / max(1, sp["launches"]),
            "ms_in_timed_region": sp["ms"],
            "note": "achieved = ALGORITHMIC bytes (operand read once, product written once: 2-3 x 8 d^2) / HIP-event time; the "
                    "kernel re-reads a partner row per nonzero of X (13 per row at n = 12), mostly from L2 / Infinity Cache: "
                    "`traffic` is the fabric-side PMC figure per launch",
        }
        ranked = sorted([(z["ms"], zg), (rs["ms"], rsy), (sp["ms"], spm)], key=lambda t: -t[0])
        out["roofline"], out["roofline_other"] = ranked[0][1], ranked[1][1]
        if world == 1 and not a.skip_other_configs:
            try:
                out["other_configs"] = other_configs(dev, a.seed, (n, a.dt, u_all[:K].copy(), psi0, loss))
            except Exception as e:
                out["other_configs"] = {"error": str(e)}
        if world == 1 and not a.skip_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(n, B, a.seed, a.dt, u_all[0])
            except Exception as e:  # the baseline is a reported extra; never lose the GPU line over it
                out["cpu_baseline"] = {"value": None, "unit": "Trotter steps/s (fwd+adjoint)", "cores": os.cpu_count(),
                                       "kind": "port", "sample": "failed: {}".format(e)}
        print(json.dumps(out))
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
