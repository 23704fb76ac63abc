"""bench.py, measuring half: the timed regions of the three BASELINE configurations, the by-input-structure and full-chain extras,
and the CPU baseline (the oracle, as the reported baseline only)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from .report import (CONFIG5_CHAIN, INFO_KEYS, PEAK_F64_MFMA_TFLOPS, PEAK_HBM_GBPS, PEAK_LDS_GBPS, TRAFFIC, describe_expm, emit,
                     executed_tflops, ranked, roofline_blocks)

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
            "roofline_mfma": blocks["zgemm"] if blocks["zgemm"]["launches"] > 0 else None,   # the chain's big complex products
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
        ent["state_columns_quad_nodes"] = e3.info("state_columns_quad_nodes")
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
    # (4) the reference's own control set (demo_maxcut.py:44-79) at the headline's size: ring QAOA, ZZ-edge controls on the diagonal and
    # X-vertex controls, K = 2 n; pulse amplitudes of the same scale (the table's first n columns twice)
    H0, Hs, M = problems.qaoa_ring_torch(n, dev)
    with Engine(d, 2 * n, device=dev.index or 0) as e4:
        e4.use_torch_stream()
        e4.set_hamiltonians(H0, Hs)
        e4.set_observable(M)
        e4.set_option("plan_steps", max(a.chain, K))
        del H0, Hs, M
        u2 = torch.cat([u_dev, u_dev], dim=1).contiguous()
        ent, _ = scheme_entry(e4, torch, u2, a.dt, psi0, d, K,
                              "ring analog QAOA at n = 12 (the reference demo's Hamiltonians, demo_maxcut.py:44-79): drift sum_e (I - Z_i Z_j), "
                              "12 ZZ-edge controls (diagonal) + 12 X-vertex controls -- state columns, quadrature adjoint in the difference basis")
        ent["state_columns"] = e4.info("state_columns")
        ent["state_columns_quad_nodes"] = e4.info("state_columns_quad_nodes")
        out["qaoa_zz_x_n12"] = ent
    torch.cuda.empty_cache()
    return out


def full_chain(eng, a, dev, psi0):
    """BASELINE configs[2] in full: the S = 2000 chain in ONE dq_energy_grad call, with the checks a driver can read:
    * the SAME 2000-step gradient once more on the explicit-propagator GEMM chain with state checkpoints (state_columns = 0: the scheme
      that was soaked against the oracle to 20 000 steps in round 1) -- `max_abs_grad_diff_vs_gemm_chain` at full d = B = 4096: the
      state-column adjoint sweep re-derives psi_t = exp(-A_t) psi_{t+1} over all 2000 steps without a checkpoint, this is what pins it
      at full size (at d = 256 / 512 the oracle itself does: tests/s2000_cases.py);
    * norm drift of Psi(T), sum of energies = loss;
    * a directional derivative of the gradient against central differences of the forward path -- on a 100-STEP chain at full d and B
      (an independent check of the gradient's meaning, not of the chain length)."""
    import torch
    from diffquantum_amd import problems
    n, S = a.qubits, a.chain
    u = torch.from_numpy(problems.pulse_table(n, S, a.dt * S, seed=a.seed)).to(dev)
    eng.set_option("plan_steps", 0)
    eng.set_option("state_columns", -1)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    loss, grad, fin, en = eng.energy_grad(u, a.dt, psi0)
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    sc, nodes = eng.info("state_columns"), eng.info("state_columns_quad_nodes")
    free, tot = torch.cuda.mem_get_info()
    res = {"chain_steps": S, "wall_s": el, "steps_per_s": S / el, "hbm_in_use_GB": (tot - free) / 1e9, "loss": loss,
           "state_columns": bool(sc > 0), "adjoint_form": ("quadrature, {} interior nodes".format(int(nodes)) if nodes > 0 else
                                                           "exact polynomial derivative") if sc > 0 else "explicit propagators",
           "sum_of_energies_minus_loss": float(en.sum()) - loss, "grad_norm": float(np.linalg.norm(grad)),
           "grad_finite": bool(np.isfinite(grad).all()),
           "max_norm_drift": float((torch.linalg.norm(fin, dim=0) - 1).abs().max().item())}
    del fin
    if sc > 0 and not a.skip_gemm_chain_check:
        eng.set_option("state_columns", 0)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss2, grad2, _, _ = eng.energy_grad(u, a.dt, psi0, want_final=False, want_energies=False)
        torch.cuda.synchronize(); el2 = time.perf_counter() - t0
        eng.set_option("state_columns", -1)
        res["gemm_chain"] = {"wall_s": el2, "steps_per_s": S / el2, "loss": loss2, "checkpoint_stride": int(eng.info("checkpoint_stride")),
                             "operator_backward": bool(eng.info("operator_backward"))}
        res["max_abs_grad_diff_vs_gemm_chain"] = float(np.max(np.abs(grad - grad2)))
        res["max_abs_grad"] = float(np.max(np.abs(grad2)))
        res["abs_loss_diff_vs_gemm_chain"] = abs(loss - loss2)
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
        "fwd_steps_per_s": c5["fwd_steps_per_s"], "fwd_adjoint_steps_per_s": c5["steps_per_s"], "polynomial_degree": c5["degree"],
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
                "fwd_adjoint_steps_per_s": len(u) / el, "polynomial_degree": int(e.info("degree")),
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
        # what the checkpoint plan of the FULL chain costs in memory (the timed slice pins only its own share): kept states in pinned
        # host memory when the plan puts them there, device pools otherwise
        two_level, host_ck = bool(e.info("two_level")), bool(e.info("host_checkpoints"))
        sbytes = float((1 << n) * B * 8)
        pinned_full = sbytes * (-(-CONFIG5_CHAIN // max(stride, 1))) / 1e9 if host_ck else 0.0
        pinned_now, pool_now = e.info("pinned_host_bytes") / 1e9, e.info("checkpoint_pool_bytes") / 1e9
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
                "two_level": two_level, "host_checkpoints": host_ck, "pinned_host_GB": pinned_full,
                "pinned_host_GB_this_slice": pinned_now, "checkpoint_pool_GB_this_slice": pool_now,
                "every_step_checkpointed": {"steps_per_s": S / el_ck, "checkpoint_stride": stride_ck},
                "roofline": {"bound": "hbm",
                             "kernel": ("sapply_dma_kernel / schain_dma_kernel (matrix-free passes, LDS-DMA tiles)" if dma_passes
                                        else "sapply_kernel / soverlap_kernel (matrix-free H.psi passes, 2^9-row sub-cubes)"),
                             "passes_per_step": passes / max(S, 1), "dma_passes_per_step": dma_passes / max(S, 1),
                             "achieved": gb, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gb / PEAK_HBM_GBPS,
                             # (the PMC record is of ONE shape: n = 18, B = 512; other shapes carry no measured traffic)
                             "traffic": (by_g / nl * 1.012 if nl and (n, B) == (18, 512) else None),
                             "traffic_over_algorithmic": 1.012 if (n, B) == (18, 512) else None,
                             "traffic_source": ("profiles/r4_bench_config5_summary.txt: PMC bytes per pass = 1.01-1.015 x its state-units x 2^n x B x 8 B "
                                                "(n = 18, B = 512)") if (n, B) == (18, 512) else None,
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


G5_LOSSES = (-2.0000395860025693, -1.9829188702462983, -2.060480806327015, -2.1216781468721955, -2.1870484989606345)   # SURVEY 8-C3 (reference run)


def oracle_demo_epochs(n_epoch, seed=0, lr=2e-2, n_basis=6, per_step=10):
    """The reference's training loop (sim_plain.py:253-296: forward solve, <M>, one-sample MC gradient, Adam) on the demo's problem with
    the ORACLE's arithmetic on the host (scipy expm per step, as sim_plain.py:145): the cpu_baseline leg of --config 1.  Consumes the
    global numpy RNG exactly as the reference does (init draw, then one uniform per epoch).  Returns (seconds, losses - ground)."""
    import torch
    from diffquantum_amd import problems
    from oracle import trotter_oracle as orc
    p = problems.maxcut_problem()
    H0, Hs, M, om, T, psi0 = p["H0"], list(p["Hs"]), p["H_cost"], p["omegas"], float(p["T"]), p["psi0"].reshape(-1, 1).astype(np.complex128)
    np.random.seed(seed)
    coeff = torch.tensor(np.random.normal(0, 1e-3, [len(Hs), n_basis]), requires_grad=True)
    opt = torch.optim.Adam([coeff], lr=lr)
    ground = float(np.linalg.eigvalsh(M)[0])
    losses = []
    t0 = time.perf_counter()
    for _ in range(n_epoch):
        c = coeff.detach().numpy().copy()
        fin, _, _ = orc.trotter_sim_plain(H0, Hs, c, om, T, "BSpline", per_step, psi0, 0, T)
        loss = float(np.real(fin[:, 0].conj() @ (M @ fin[:, 0])))
        s = np.random.uniform() * T
        g, _, _ = orc.mc_gradient(H0, Hs, M, c, om, T, "BSpline", per_step, psi0, s)
        opt.zero_grad()
        coeff.grad = torch.from_numpy(g)
        opt.step()
        losses.append(loss - ground)
    return time.perf_counter() - t0, losses


def run_config1(a, D):
    """BASELINE configs[0]: demo_maxcut.py's 4-vertex analog QAOA (d = 16, 8 controls, batch 1, 30 Trotter steps per solve) -- training
    epochs per second through SimulatorPlain.train_energy on the engine (one epoch = forward solve + the reference's one-sample MC
    gradient estimate, 17 chains as one device-resident batch, + Adam; demo_maxcut.py:86, sim_plain.py:253-296), the first five losses
    against the reference's recorded trace, and the same loop through the oracle on the host beside it."""
    import torch
    from diffquantum_amd import problems
    from diffquantum_amd.qobj import Qobj
    from diffquantum_amd.sim_plain import SimulatorPlain
    K = a.steps if a.steps is not None else 60
    p = problems.maxcut_problem()
    log_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "logs")
    os.makedirs(os.path.join(log_dir, "text"), exist_ok=True)

    def make(n_epoch):
        sim = SimulatorPlain(lr=2e-2, n_basis=6, n_epoch=n_epoch, log_dir=log_dir)
        sim.T, sim.omegas = p["T"], list(p["omegas"])
        return sim
    args = (Qobj(p["H_cost"]), Qobj(p["H0"]), [Qobj(h) for h in p["Hs"]], Qobj(p["psi0"]))
    np.random.seed(0)
    sim = make(max(5, a.warmup))
    sim.train_energy(*args)                                   # warm-up; its first five epochs are the reference's trace G5
    trace = [float(x) for x in sim.losses_energy[:5]]
    ground = float(np.linalg.eigvalsh(p["H_cost"])[0])
    trace_err = float(np.max(np.abs(np.array(trace) + ground - np.array(G5_LOSSES))))   # (losses_energy = loss - ground; G5 holds the losses)
    times = []
    for _ in range(max(1, a.trials)):
        sim = make(K)
        np.random.seed(1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        sim.train_energy(*args)
        torch.cuda.synchronize(); times.append(time.perf_counter() - t0)
    el = float(np.median(times))
    cpu = None
    if not a.skip_cpu_baseline:
        n_cpu = 12
        sec, closs = oracle_demo_epochs(n_cpu)
        cpu = {"value": n_cpu / sec, "unit": "training epochs/s", "cores": 1, "kind": "port",
               "sample": "{} epochs of the same loop through the oracle on the host (scipy expm per Trotter step, 17 chains per epoch one "
                         "after the other, as the reference runs them): {:.2f} s; first losses {}".format(
                             n_cpu, sec, ", ".join("{:.6f}".format(x + ground) for x in closs[:3]))}
    out = {"metric": "training epochs/sec, demo_maxcut.py 4-vertex analog QAOA (batch 1)", "value": K / el, "unit": "training epochs/s",
           "n_gpus": 1, "steps": K, "warmup": max(5, a.warmup), "ms_per_step": el / K * 1e3, "higher_is_better": True, "scaling": "weak",
           "vs_baseline": None, "dtype": "f64", "data": "the demo's own problem (no dataset)", "backend": D.name(),
           "trials": {"n": len(times), "ms_per_step_median": el / K * 1e3, "ms_per_step_min": min(times) / K * 1e3,
                      "ms_per_step_max": max(times) / K * 1e3},
           "config": {"workload": "BASELINE configs[0]: demo_maxcut.py 4-vertex analog QAOA (d=16, 8 controls, batch 1, 30 Trotter steps per "
                                  "solve), SimulatorPlain.train_energy: one step = one epoch (forward solve + one-sample MC gradient: 17 chains "
                                  "as one device batch + Adam)",
                      "n_qubits": 4, "dim": 16, "n_controls": 8, "batch_per_gpu": 1, "global_batch": 1, "chain_steps": 30, "parallelism": "dp1",
                      "first_five_losses": [x + ground for x in trace], "max_abs_diff_vs_reference_trace_G5": trace_err,
                      "note": "latency-bound: d = 16, batch 1 -- a handful of microsecond kernels per epoch; host time (pulse tables, "
                              "Python, Adam) is most of an epoch"},
           "roofline": None, "cpu_baseline": cpu}
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
                          "polynomial_degree": m["degree"], "loss": m["loss"], "grad_norm": m["grad_norm"],
                          "forward_only_steps_per_s_rank0": m["fwd_steps_per_s"],
                          "checkpoint_stride": m["checkpoint_stride"], "every_step_checkpointed_steps_per_s_rank0": m["every_step_checkpointed"]["steps_per_s"],
                          # the checkpoint plan of the full 4000-step chain per rank: two levels? kept states in pinned HOST memory (GB)?
                          "two_level": m["two_level"], "host_checkpoints": m["host_checkpoints"], "pinned_host_GB": m["pinned_host_GB"],
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
    from diffquantum_amd.distributed import (allreduce_loss_grad, column_sharded_energy_grad, propagator_sharded_energy_grad,
                                             time_sliced_energy_grad)
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
    elif a.workload == "xy-drives":     # I/Q drives: X_k on half of the qubits, Y_k on the others (by_input_structure's inputs)
        H0, Hs, M = problems.xy_drives_torch(n, a.seed, dev, n_x=n // 2, n_y=n - n // 2)
    elif a.workload == "qaoa-ring":     # the reference demo's Hamiltonians at n qubits: ZZ-edge + X-vertex controls (K = 2 n)
        H0, Hs, M = problems.qaoa_ring_torch(n, dev)
    else:
        H0, Hs, M = problems.random_dense_torch(n, n, a.seed, dev, complex_h=(a.workload == "dense-complex"))
    nc = int(Hs.shape[0])               # controls
    u_all = problems.pulse_table(n, max(K, W, 1), a.dt * max(K, W, 1), seed=a.seed)
    if nc != n:
        u_all = np.concatenate([u_all] * (nc // n), axis=1)
    psi0 = torch.from_numpy(problems.random_states(d, B, a.seed, first=rank * B)).to(dev)
    eng = Engine(d, nc, device=D.local)
    eng.use_torch_stream()
    eng.set_hamiltonians(H0, Hs)
    eng.set_observable(M)
    M_dev = M if a.multi_gpu == "propagators" else None
    del H0, Hs, M
    eng.set_option("real_path", a.real_path)
    eng.set_option("state_columns", a.state_columns)
    eng.set_option("plan_steps", max(a.chain, K))  # checkpoint/rebuild ratio of the full 2000-step chain
    u_t = torch.from_numpy(u_all[:K].copy()).to(dev)
    grad_t = torch.zeros((K, nc), dtype=torch.float64, device=dev)

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

    def run_columns(u_dev, grad_dev):
        # the WHOLE global batch's energy gradient with the adjoint sweep split over the dim COLUMNS of (P R, 2 M P) (round 5): Gram
        # matrix of this rank's states, all-reduce of R (16 d^2 bytes), P built by every rank, d / world columns swept, one all-reduce
        # of (loss, dL/du).  Strong scaling of the job that one GPU runs on dim columns -- batch shards only add states
        loss, g = column_sharded_energy_grad(eng, u_dev, a.dt, psi0, device=dev)
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
    mode = "sliced" if (a.multi_gpu == "sliced" and D.on) or both else (a.multi_gpu if a.multi_gpu in ("propagators", "columns") else "replicated")
    want_beside = multi and a.multi_gpu in ("replicated", "auto") and not a.no_beside
    forms = {}
    sliced_check = None
    columns_check = None
    columns_ok = a.workload in ("ising", "xy-drives", "qaoa-ring") and a.state_columns != 0   # (needs a Pauli-sum generator on the state-column kernels)
    if mode == "columns" and not columns_ok:
        raise SystemExit("--multi-gpu columns needs the Ising workload with the state-column form")
    if W > 0:
        uw = torch.from_numpy(u_all[:W].copy()).to(dev)
        lr, gr = run_replicated(uw, torch.zeros((W, nc), dtype=torch.float64, device=dev))
        if columns_ok and (mode == "columns" or want_beside):
            lc, gc = run_columns(uw, torch.zeros((W, nc), dtype=torch.float64, device=dev))
            errc = float((gc - gr).abs().max().item())
            columns_check = {"max_abs_grad_diff_vs_replicated": errc, "abs_loss_diff_vs_replicated": abs(lc - lr)}
            okc = errc <= 1e-9 * (1.0 + float(gr.abs().max().item())) and abs(lc - lr) <= 1e-9 * (1.0 + abs(lr))
            flagc = torch.tensor([1.0 if okc else 0.0], dtype=torch.float64, device=dev)
            D.all_reduce_(flagc, dist.ReduceOp.MIN)
            columns_check["agreed_on_every_rank"] = bool(flagc.item() >= 1.0)
            if flagc.item() < 1.0 and mode == "columns":
                raise SystemExit("column-sharded form disagrees with the replicated form: {}".format(columns_check))
        if mode == "sliced" or want_beside:
            # the warm-up doubles as the check: same loss and gradient from both forms on every rank
            ls, gs = run_sliced(uw, torch.zeros((W, nc), dtype=torch.float64, device=dev))
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
            g2 = torch.zeros((K, nc), dtype=torch.float64, device=dev)
            D.fence()
            t1 = time.perf_counter()
            l2, g2 = fn(u_t, g2)
            D.fence()
            forms[name] = {"seconds": D.max_over_ranks(time.perf_counter() - t1), "loss": l2, "grad": g2.clone()}
    if want_beside and columns_ok and columns_check and columns_check["agreed_on_every_rank"]:
        g2 = torch.zeros((K, nc), dtype=torch.float64, device=dev)
        run_columns(u_t[:1].contiguous(), torch.zeros((1, nc), dtype=torch.float64, device=dev))   # (allocations)
        D.fence()
        t1 = time.perf_counter()
        l2, g2 = run_columns(u_t, g2)
        D.fence()
        forms["columns"] = {"seconds": D.max_over_ranks(time.perf_counter() - t1), "loss": l2, "grad": g2.clone()}
    if both:
        mode = "replicated" if forms["replicated"]["seconds"] < forms["sliced"]["seconds"] else "sliced"
    whole = None
    if want_beside and rank == 0:
        psi_all = torch.from_numpy(problems.random_states(d, B * world, a.seed, first=0)).to(dev) if world > 1 else psi0
        eng.set_option("plan_steps", max(a.chain, K))
        eng.set_option("cache_bytes", 0)
        g2 = torch.zeros((K, nc), dtype=torch.float64, device=dev)
        eng.energy_grad(u_t[:1].contiguous(), a.dt, psi_all, want_final=False, want_energies=False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        eng.energy_grad(u_t, a.dt, psi_all, want_final=False, want_energies=False, grad_out=g2)
        torch.cuda.synchronize(); whole = time.perf_counter() - t1
        del psi_all
    prop_check = None
    if mode == "propagators" and W > 0:   # the warm-up chain doubles as the check against the replicated form
        lp, gp = run_propagators(uw, torch.zeros((W, nc), dtype=torch.float64, device=dev))
        prop_check = {"max_abs_grad_diff_vs_replicated": float((gp - gr).abs().max().item()), "abs_loss_diff_vs_replicated": abs(lp - lr)}
    run = {"sliced": run_sliced, "propagators": run_propagators, "columns": run_columns}.get(mode, run_replicated)
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
        if "replicated" in forms and "sliced" in forms:
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
        "xy-drives": "Ising drift + X / Y drives: a complex Hermitian sum of Pauli strings on the state-column kernels",
        "qaoa-ring": "drift sum_e (I - Z_i Z_j), ZZ-edge controls (diagonal) + X-vertex controls: a sum of Pauli strings on the state-column kernels",
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
        if "columns" in forms:   # the whole global batch's gradient, adjoint sweep split over the dim columns across the ranks
            beside["columns_form"] = [world * K / forms["columns"]["seconds"], forms["columns"]["seconds"] / K * 1e3]
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
                                                               "dense-complex": "dense random complex Hermitian H",
                                                               "xy-drives": "random Ising drift + X_k / Y_k drives (not a BASELINE config: by_input_structure's inputs as the timed workload)",
                                                               "qaoa-ring": "ring QAOA, ZZ-edge + X-vertex controls (the reference demo's Hamiltonians at n = 12; not a BASELINE config)"}[a.workload], d, nc, B, K,
                                                              max(a.chain, K)),
            "workload_detail": "stored dense; dt={}; batch >= dim: operator-form backward sweep, no state checkpoints, per-step cost "
                               "independent of the chain length; otherwise the slice uses the checkpoint stride of the full chain".format(a.dt),
            "input_structure": structure_note,
            "n_qubits": n, "dim": d, "n_controls": nc, "batch_per_gpu": B, "global_batch": B * world,
            "chain_steps": K, "dt": a.dt,
            # (SURVEY D1 "pulse-table upload included": the table is 8 K S bytes -- 2 KB here -- and is handed over as a DEVICE tensor that is
            # resident before the clock starts; inside the timed call the library copies it to the host once for the step bound and back)
            "pulse_table": "device-resident before the timed region ({} B); read back by the call for the step bound".format(8 * K * n),
            "parallelism": "dp{}".format(world),
            "parallelism_detail": ("dp{} (batch shards, one all-reduce of loss+dL/du)".format(world) if mode == "replicated" else
                                   "dp{0} over the batch, the propagators of every window of {0} steps built one per rank and all-gathered; "
                                   "state-form chain, reduce-scatter of the propagators' cotangents (SURVEY 8-F2 as written)".format(world)
                                   if mode == "propagators" else
                                   "the whole {1}-state batch's energy gradient on the dim columns of (P R, 2 M P): Gram matrices all-reduced "
                                   "(16 d^2 bytes), P built by every rank, d / {0} columns swept per rank, one all-reduce of loss + dL/du "
                                   "(strong scaling of the job one GPU runs on dim columns)".format(world, B * world)
                                   if mode == "columns" else
                                   "dp{0} over the batch x {0} time slices: operator form, each rank builds and sweeps 1/{0} of "
                                   "the chain for the whole batch; all-gather of the slice products, all-reduce of W_S "
                                   "(d x d) and of dL/du".format(world)),
            "multi_gpu_form": mode, "sliced_warmup_check": sliced_check, "propagators_warmup_check": prop_check,
            "columns_warmup_check": columns_check,
            "multi_gpu_forms": forms_out,
            "collectives_forced_at_world_1": force and world == 1,
            "expm": describe_expm(info),
            "path": ("state-column form (Pauli-string H)" if info.get("state_columns", 0) > 0 else
                     "real-symmetric fast path (H real)" if info["real_path"] > 0 else "general complex path"),
            "state_columns": bool(info.get("state_columns", 0) > 0),
            "polynomial_degree": int(info["degree"]), "taylor_degree_at_same_bound": int(info.get("state_columns_taylor_degree", 0)) or None,
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
        # north_star's quantity (>= 40 % of the fp64 MFMA peak on the batched propagator GEMM): the chain's complex GEMM on the SAME
        # inputs with the state-column form switched off (the by_input_structure run of this process)
        "roofline_mfma": (structure or {}).get("ising_inputs_gemm_chain", {}).get("roofline_mfma"),
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
            fc = full_chain(eng, a, dev, psi0)
            oc["full_chain_S{}".format(a.chain)] = fc
            # [steps, steps/s of the whole chain in one call, max |dL/du - the checkpointed GEMM chain's| over all 2000 x 12 entries]
            out["full_chain"] = [fc["chain_steps"], fc["steps_per_s"], fc.get("max_abs_grad_diff_vs_gemm_chain")]
        out["other_configs"] = oc
    if world == 1 and not D.on and not a.skip_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(n, B, a.seed, a.dt, u_all[0])
        except Exception as e:  # the baseline is a reported extra; never lose the GPU line over it
            out["cpu_baseline"] = {"value": None, "unit": "Trotter steps/s (fwd+adjoint)", "cores": os.cpu_count(),
                                   "kind": "port", "sample": "failed: {}".format(e)}
    emit(out)
    eng.close()


