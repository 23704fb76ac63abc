"""The BASELINE.json configurations themselves under `-m gpu`:

  configs[2]  n = 12 (d = 4096): the ORACLE (the reference's expm + matmul arithmetic, sim_plain.py:145-146 /
              diffqc.cc:192-198, plus one scipy expm_frechet per step) at the headline dimension, against every
              evaluation scheme of the engine -- the 128-tile 3M complex GEMM, the (anti-)Hermitian-output K-split
              tails, the real symmetric GEMM and the sparse x dense passes are all exercised at d = 4096 here.
              Host cost: about a minute per oracle evaluation (2 steps, 32 states); the answers are also a committed
              fixture (tests/d4096_cases.py, tests/golden/make_oracle_d4096_goldens.py) used when made for identical inputs.
  configs[1]  8-qubit ring QAOA (d = 256, K = 16): against the oracle at S = 20 / B = 48 (amplitudes + gradient), at the
              full S = 1000 for a few columns, and the full S = 1000 / B = 1024 forward through size-independent properties.
  configs[4]  n = 18 (d = 262144), 64 states per GPU, complex64, matrix-free engine: size-independent properties.

Tolerances: BASELINE.json north_star -- 1e-10 in fp64 (amplitudes and gradients), 1e-4 in fp32.
"""
import numpy as np
import pytest

from diffquantum_amd import problems

pytestmark = pytest.mark.gpu
TOL = 1e-10


# ---------------------------------------------------------------------------------------------------------------
# configs[2]: the headline dimension against the oracle
# ---------------------------------------------------------------------------------------------------------------

N12 = dict(n=12, S=2, B=32, dt=1e-3)


@pytest.fixture(scope="module")
def ising12(oracle):
    """BASELINE's workload (random Ising drift stored dense, X_k controls, M = H0) + the oracle's answer (tests/d4096_cases.py: from
    the committed fixture when it was made for these very inputs, else computed here -- about a minute)."""
    import d4096_cases
    P, a = d4096_cases.answer("ising12", oracle)
    return dict(P, loss=a["loss"], grad=a["grad"], fin=a["fin"], energies=a["energies"])


SCHEMES = {
    # name: (options, expected info)
    "sparse_generator": ({}, {"real_path": True, "sparse_generator": True, "xor_regular": True}),
    "sparse_generator_row_kernels": ({"xor_regular": 0}, {"real_path": True, "sparse_generator": True, "xor_regular": False}),
    "sparse_controls_dense_drift": ({"sparse_generator": 0}, {"real_path": True, "sparse_generator": False, "sparse_controls": True}),
    "dense_real": ({"sparse_generator": 0, "sparse_controls": 0}, {"real_path": True, "sparse_generator": False, "sparse_controls": False}),
    "complex_tape": ({"real_path": 0}, {"real_path": False}),
    # the state-column form (statecol.hip): no propagator at all; `compose` is left on auto (an explicit value pins the other forms)
    "state_columns": ({"state_columns": 1}, {"real_path": True, "sparse_generator": True, "xor_regular": True}),
}


@pytest.mark.parametrize("compose", [1, 0], ids=["operator_form", "state_form"])
@pytest.mark.parametrize("scheme", list(SCHEMES))
def test_headline_dimension_matches_oracle(ising12, scheme, compose):
    """n = 12, d = 4096: final amplitudes, loss and dL/du of a 2-step chain against the oracle at 1e-10, for every scheme
    (operator form = what the headline shape B >= 3/4 d runs; state form = what narrower batches run)."""
    from diffquantum_amd.engine import Engine
    P = ising12
    d, n, dt = len(P["H0"]), N12["n"], N12["dt"]
    opts, want = SCHEMES[scheme]
    with Engine(d, n) as e:
        e.set_hamiltonians(P["H0"], P["Hs"])
        e.set_observable(P["M"])
        for k, v in opts.items():
            e.set_option(k, v)
        if scheme == "state_columns":
            # ("state_form" id: the 32 columns on 5 workgroups -- 7, 7, 6, 6, 6 columns each: the several-columns-per-workgroup loop of
            # the headline's kernels (sums accumulated over a workgroup's columns, images reused) against the ORACLE at this dimension;
            # "operator_form" id: one column per workgroup)
            if not compose:
                e.set_option("state_columns_workgroups", 5)
        else:
            e.set_option("compose", compose)
            e.set_option("state_columns", 0)
        fwd = e.propagate(P["u"], dt, P["psi0"])
        assert e.info("state_columns") == (1 if scheme == "state_columns" else 0)
        loss, grad, fin, en = e.energy_grad(P["u"], dt, P["psi0"])
        assert e.info("state_columns") == (1 if scheme == "state_columns" else 0)
        assert (e.info("real_path") >= 1) == want["real_path"]
        if "sparse_generator" in want:
            assert (e.info("sparse_generator") >= 1) == want["sparse_generator"]
        if "sparse_controls" in want:
            assert (e.info("sparse_controls") >= 1) == want["sparse_controls"]
        if "xor_regular" in want:
            assert (e.info("xor_regular") >= 1) == want["xor_regular"]
        assert e.info("operator_backward") == (1 if compose and scheme != "state_columns" else 0)
    nc = P["fin"].shape[1]                                                # amplitudes of the stored columns, energies of all
    assert np.max(np.abs(fwd[:, :nc] - P["fin"])) < TOL and np.max(np.abs(fwd - fin)) < TOL
    assert np.max(np.abs(fin[:, :nc] - P["fin"])) < TOL and np.max(np.abs(en - P["energies"])) < TOL
    assert abs(loss - P["loss"]) < TOL * N12["B"] and abs(float(en.sum()) - P["loss"]) < TOL * N12["B"]
    assert np.max(np.abs(grad - P["grad"])) < TOL, np.max(np.abs(grad - P["grad"]))
    assert np.max(np.abs(P["grad"])) > 1e-4                             # (the comparison is not of zeros)


@pytest.fixture(scope="module")
def xyz12(oracle):
    """d = 4096 with X, Y and Z drives (a COMPLEX Hermitian Pauli sum: I/Q control plus a detuning; tests/d4096_cases.py) and the
    oracle's answer."""
    import d4096_cases
    P, a = d4096_cases.answer("xyz12", oracle)
    return dict(P, loss=a["loss"], grad=a["grad"], fin=a["fin"], energies=a["energies"])


@pytest.mark.parametrize("form", ["state_columns", "column_form_operator", "column_form_state", "complex_tape"])
def test_headline_dimension_complex_pauli_sum_matches_oracle(xyz12, form):
    """the COMPLEX column form (zxr_* kernels, xorexp.hip) directly against the oracle at d = 4096 (it was
    compared with the oracle at d <= 512 and with the engine's own tape above), in operator and in state form; the
    state-column form (statecol.hip) on the same complex generator; and the tape."""
    from diffquantum_amd.engine import Engine
    P = xyz12
    with Engine(4096, 12) as e:
        e.set_hamiltonians(P["H0"], P["Hs"])
        e.set_observable(P["M"])
        assert e.info("real_hamiltonians") == 0 and e.info("xor_regular_found") == 11
        if form == "state_columns":
            e.set_option("state_columns", 1)
            e.set_option("state_columns_workgroups", 3)      # 16 columns on 3 workgroups (6 + 5 + 5): the several-columns loop, signed kernels
        else:
            e.set_option("state_columns", 0)
            e.set_option("compose", 1 if form == "column_form_operator" else 0)
            if form == "complex_tape":
                e.set_option("xor_regular", 0)
        fwd = e.propagate(P["u"], P["dt"], P["psi0"])
        loss, grad, fin, en = e.energy_grad(P["u"], P["dt"], P["psi0"])
        assert e.info("state_columns") == (1 if form == "state_columns" else 0)
        assert e.info("xor_regular") == (0 if form == "complex_tape" else 2)
        if form != "state_columns":
            assert e.info("operator_backward") == (1 if form == "column_form_operator" else 0)
    nc = P["fin"].shape[1]
    assert np.max(np.abs(fwd[:, :nc] - P["fin"])) < TOL and np.max(np.abs(fin[:, :nc] - P["fin"])) < TOL
    assert np.max(np.abs(fwd - fin)) < TOL and np.max(np.abs(en - P["energies"])) < TOL
    assert abs(loss - P["loss"]) < TOL * P["B"] and abs(float(en.sum()) - P["loss"]) < TOL * P["B"]
    assert np.max(np.abs(grad - P["grad"])) < TOL, np.max(np.abs(grad - P["grad"]))
    assert np.max(np.abs(P["grad"])) > 1e-4


def test_headline_dimension_complex_hermitian_matches_oracle(oracle):
    """d = 4096 with a dense COMPLEX Hermitian drift and controls (what `set_H`, diffqc.cc:43-67, accepts in general):
    the complex tape with scaling-and-squaring (theta ~ 1) and the Hermitian-output products on truly complex data."""
    from diffquantum_amd.engine import Engine
    import d4096_cases
    P, a = d4096_cases.answer("herm4096", oracle)
    H0, Hs, M, u, psi0, dt, B = P["H0"], P["Hs"], P["M"], P["u"], P["psi0"], P["dt"], P["B"]
    d, K = 4096, len(Hs)
    rl, rg, rf, ren = a["loss"], a["grad"], a["fin"], a["energies"]
    for compose in (1, 0):
        with Engine(d, K) as e:
            e.set_hamiltonians(H0, Hs)
            e.set_observable(M)
            e.set_option("compose", compose)
            loss, grad, fin, en = e.energy_grad(u, dt, psi0)
            assert e.info("real_path") == 0 and e.info("operator_backward") == compose
        assert np.max(np.abs(fin[:, :rf.shape[1]] - rf)) < TOL and np.max(np.abs(en - ren)) < TOL and abs(loss - rl) < TOL * B
        assert np.max(np.abs(grad - rg)) < TOL, (compose, np.max(np.abs(grad - rg)))
    assert np.max(np.abs(rg)) > 1e-4


# ---------------------------------------------------------------------------------------------------------------
# configs[1]: 8-qubit ring QAOA, 1000 Trotter steps, batch 1024, forward
# ---------------------------------------------------------------------------------------------------------------

def _qaoa8():
    p = problems.qaoa_ring(8)
    return p["H0"].astype(np.complex128), p["Hs"].astype(np.complex128), p["H_cost"].astype(np.complex128)


def test_config2_qaoa_ring_matches_oracle(oracle):
    """qaoa_ring(8) (d = 256, K = 16 = 8 edge ZZ + 8 vertex X), S = 20, B = 48, default engine settings: amplitudes and
    the adjoint gradient against the oracle."""
    from diffquantum_amd.engine import Engine
    H0, Hs, M = _qaoa8()
    S, B = 20, 48
    T = 1.0
    u = problems.pulse_table(16, S, T, seed=0)
    psi0 = problems.random_states(256, B, seed=0)
    ref = oracle.propagate(H0, list(Hs), u, T / S, psi0)
    rl, rg, rf = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, T / S, psi0)
    with Engine(256, 16) as e:
        e.set_hamiltonians(H0, Hs)
        e.set_observable(M)
        got = e.propagate(u, T / S, psi0)
        loss, grad, fin, en = e.energy_grad(u, T / S, psi0)
    assert np.max(np.abs(got - ref)) < TOL and np.max(np.abs(fin - rf)) < TOL
    assert abs(loss - rl) < TOL * B and np.max(np.abs(grad - rg)) < TOL


def test_config2_full_size_forward(oracle):
    """The configuration as BASELINE states it: S = 1000, B = 1024, fp64, forward only.  Four columns against the oracle's
    1000-step loop, the whole batch through norm conservation, linearity and time reversal."""
    import torch
    from diffquantum_amd.engine import Engine
    H0, Hs, M = _qaoa8()
    S, B = 1000, 1024
    T = 1.0
    u = problems.pulse_table(16, S, T, seed=0)
    psi0 = problems.random_states(256, B, seed=0)
    dev = torch.device("cuda:0")
    p0 = torch.from_numpy(psi0).to(dev)
    with Engine(256, 16) as e:
        e.use_torch_stream()
        e.set_hamiltonians(H0, Hs)
        out = e.propagate(u, T / S, p0)
        assert float((torch.linalg.norm(out, dim=0) - 1).abs().max()) < 1e-12
        mix = (0.6 * p0[:, :512] + (0.3 - 0.2j) * p0[:, 512:]).contiguous()
        lin = e.propagate(u, T / S, mix) - (0.6 * out[:, :512] + (0.3 - 0.2j) * out[:, 512:])
        assert float(lin.abs().max()) < 1e-12
        back = e.propagate(u[::-1].copy(), -T / S, out)
        assert float((back - p0).abs().max()) < 1e-11
        cols = [0, 1, 517, 1023]
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=1):     # d = 256: one BLAS thread is several times faster than an oversubscribed pool
            ref = oracle.propagate(H0, list(Hs), u, T / S, psi0[:, cols])
        assert np.max(np.abs(out[:, cols].cpu().numpy() - ref)) < TOL


# ---------------------------------------------------------------------------------------------------------------
# configs[4]: 18 qubits, complex64, 64 states per GPU (512 over 8), matrix-free engine
# ---------------------------------------------------------------------------------------------------------------

def test_config5_n18_properties():
    """n = 18 (d = 262144; a dense complex64 H would be 550 GB), B = 64, dt = 1e-3, complex64: norm conservation,
    linearity, time reversal, loss consistency and a directional derivative of the adjoint gradient (fp32: 1e-4 class)."""
    import torch
    from diffquantum_amd.structured import StructuredEngine, ising_x_terms
    n, B, S, dt = 18, 64, 8, 1e-3
    d = 1 << n
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(18)
    st = rng.standard_normal((d, B)) + 1j * rng.standard_normal((d, B))
    psi0 = torch.from_numpy((st / np.linalg.norm(st, axis=0)).astype(np.complex64)).to(dev)
    u = problems.pulse_table(n, S, S * dt, seed=1)
    with StructuredEngine(n, n, dtype="c64") as e:
        e.use_torch_stream()
        e.set_terms(ising_x_terms(n, seed=0))
        out = e.propagate(u, dt, psi0)
        nrm = torch.linalg.norm(out.to(torch.complex128), dim=0)
        assert float((nrm - 1).abs().max()) < 1e-5
        mix = (0.6 * psi0[:, :32] + (0.3 - 0.2j) * psi0[:, 32:]).contiguous()
        lin = e.propagate(u, dt, mix) - (0.6 * out[:, :32] + (0.3 - 0.2j) * out[:, 32:])
        assert float(lin.abs().max()) < 1e-5 * float(out.abs().max()) * 50      # amplitudes are ~ 2e-3 at d = 262144
        back = e.propagate(u[::-1].copy(), -dt, out)
        assert float((back - psi0).abs().max()) < 1e-4 * float(psi0.abs().max())
        loss, grad, fin, en = e.energy_grad(u, dt, psi0)
        assert float((fin - out).abs().max()) < 1e-6

        def L(uu):
            f = e.propagate(uu, dt, psi0)
            return float(torch.real(torch.sum(torch.conj(f.to(torch.complex128)) * e.apply(f, observable=True).to(torch.complex128))))

        assert abs(L(u) - loss) < 1e-4 * max(1.0, abs(loss))
        v = np.random.default_rng(0).standard_normal(u.shape)
        eps = 0.25
        fd, an = (L(u + eps * v) - L(u - eps * v)) / (2 * eps), float(np.sum(grad * v))
        assert abs(fd - an) < 2e-3 * max(1.0, abs(an)), (fd, an)
        assert e.info("degree") >= 4


# ---------------------------------------------------------------------------------------------------------------
# a dimension that is NOT a power of two, large enough for the 16-wave linear-form GEMM (1536 = 12 x 128) but not for the
# XCD-slab sparse passes (1024-multiples only): the kernel-selection boundaries against the oracle
# ---------------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("kind", ["complex", "real_sparse"])
def test_dimension_1536_matches_oracle(oracle, kind):
    from diffquantum_amd.engine import Engine
    d, K, B, S, dt = 1536, 2, 24, 2, 0.004
    rng = np.random.default_rng(1536)
    if kind == "complex":
        def herm():
            a = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
            return (a + a.conj().T) / (4.0 * np.sqrt(d))
        H0, Hs = herm(), np.stack([herm() for _ in range(K)])
    else:
        H0 = np.diag(rng.standard_normal(d)).astype(complex)
        Hs = np.zeros((K, d, d), dtype=complex)
        for k in range(K):
            for i in range(0, d - (k + 1), 2 * (k + 1)):
                Hs[k, i, i + k + 1] = Hs[k, i + k + 1, i] = rng.standard_normal()
    M = np.diag(rng.standard_normal(d)).astype(complex)
    u = rng.uniform(-1, 1, (S, K))
    psi0 = problems.random_states(d, B, seed=7)
    rl, rg, rf = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, dt, psi0)
    for compose in (1, 0):
        with Engine(d, K) as e:
            e.set_hamiltonians(H0, Hs)
            e.set_observable(M)
            e.set_option("compose", compose)
            loss, grad, fin, en = e.energy_grad(u, dt, psi0)
            if kind == "real_sparse":
                assert e.info("sparse_generator") == 1
        assert np.max(np.abs(fin - rf)) < TOL and abs(loss - rl) < TOL * B
        assert np.max(np.abs(grad - rg)) < TOL, (kind, compose, np.max(np.abs(grad - rg)))


# ---------------------------------------------------------------------------------------------------------------
# configs[3]'s shape on one GPU: n = 12, the WHOLE 32768-state batch (8 GPUs x 4096 in BASELINE; one GPU holds it)
# ---------------------------------------------------------------------------------------------------------------

def test_config4_shape_whole_batch_on_one_gpu(oracle):
    """d = 4096, B = 32768 (configs[3]'s global batch), S = 3, fp64, fwd + adjoint: the wide-batch form (adjoint sweep on the 4096
    columns of (P R, 2 M P), engine.hip do_energy_grad_xs_wide -- the form one GPU runs that job in, and one rank's program of the
    column-sharded multi-GPU form) against the sweep over the 32768 states themselves -- loss, dL/du, per-state energies -- and four
    columns of Psi(T) with their energies against the oracle's chain (diffqc.cc:190-200 on the same inputs)."""
    import torch
    from diffquantum_amd.engine import Engine
    n, B, S, dt = 12, 32768, 3, 1e-3
    d = 1 << n
    dev = torch.device("cuda:0")
    H0t, Hst, Mt = problems.random_ising_torch(n, 0, dev)
    u = problems.pulse_table(n, 2000, 2000 * dt, seed=0)[:S].copy()
    u[1] += 0.37 * np.random.default_rng(5).uniform(-1, 1, n)
    u[2] -= 0.21 * np.random.default_rng(6).uniform(-1, 1, n)
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234)
    psi0 = torch.randn((d, B), dtype=torch.float64, device=dev, generator=gen) + 1j * torch.randn((d, B), dtype=torch.float64, device=dev, generator=gen)
    psi0 = (psi0 / torch.linalg.vector_norm(psi0, dim=0, keepdim=True)).contiguous()
    with Engine(d, n) as e:
        e.use_torch_stream()
        e.set_hamiltonians(H0t, Hst)
        e.set_observable(Mt)
        e.set_option("state_columns_wide_adjoint", 1)
        l1, g1, f1, e1 = e.energy_grad(u, dt, psi0)
        assert e.info("state_columns_wide_adjoint") == 1 and e.info("state_columns_compose_segments") >= 1
        l2, g2, _, _ = e.energy_grad(u, dt, psi0, want_final=False, want_energies=False)
        assert e.info("state_columns_wide_adjoint") == 1
        e.set_option("state_columns", 1)
        e.set_option("state_columns_wide_adjoint", 0)
        l0, g0, f0, e0 = e.energy_grad(u, dt, psi0)
        assert e.info("state_columns") == 1 and e.info("state_columns_wide_adjoint") == 0
        assert e.info("state_columns_quad_nodes") == 2     # (the swept states: the headline's kernels)
    scale = max(1.0, float(np.max(np.abs(g0))))
    assert abs(l1 - l0) < TOL * B and abs(l2 - l0) < TOL * B, (l1 - l0, l2 - l0)
    assert np.max(np.abs(g1 - g0)) < TOL * scale and np.max(np.abs(g2 - g0)) < TOL * scale, (np.max(np.abs(g1 - g0)), scale)
    assert np.max(np.abs(e1 - e0)) < TOL * 10 and float((f1 - f0).abs().max().item()) < TOL * 10
    assert abs(float(np.sum(e1)) - l1) < TOL * B and np.max(np.abs(g0)) > 1e-3
    # four columns against the oracle
    H0, Hs, M = problems.random_ising(n, seed=0)
    assert float((H0t - torch.from_numpy(H0).to(dev)).abs().max().item()) == 0.0
    cols = psi0[:, :4].cpu().numpy()
    ref = oracle.propagate(H0, list(Hs), u, dt, cols)
    ren = oracle.energies(M, ref)
    assert np.max(np.abs(f1[:, :4].cpu().numpy() - ref)) < TOL and np.max(np.abs(f0[:, :4].cpu().numpy() - ref)) < TOL
    assert np.max(np.abs(e1[:4] - ren)) < TOL and np.max(np.abs(e0[:4] - ren)) < TOL
