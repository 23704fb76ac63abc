"""CPU tests: pin the oracle (and the host-side pulse model / problem builders) to golden vectors
captured from the reference itself (tests/golden/make_goldens.py imports /root/reference/sim_plain.py
and runs /root/reference/demo_maxcut.py in place).  No GPU needed."""
import os
import re

import numpy as np
import pytest

from diffquantum_amd import problems, pulse


@pytest.fixture(scope="module")
def G(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "reference_goldens.npz")))


# ---- G1 / G2: basis functions and generate_u -------------------------------------------------------

def test_g1_bump_basis(oracle, G):
    for nb, key in ((6, "g1_bspline6"), (4, "g1_bspline4")):
        ref = G[key]
        mine = np.array([[oracle.bspline_py(b, nb, float(x)) for x in G["g1_x"]] for b in range(nb)])
        assert np.array_equal(mine, ref)
        cc = np.array([[oracle.bspline_cc(b, nb, float(x)) for x in G["g1_x"]] for b in range(nb)])
        assert np.max(np.abs(cc - ref)) < 1e-15
        assert np.max(np.abs(pulse.bump_table(nb, G["g1_x"]) - ref)) < 1e-15
        # quirk: bump 0 has support (-3 tau, 0) and vanishes on [0, 1]
        assert np.all(ref[0][G["g1_x"] >= 0] == 0)


def test_g1_legendre(oracle, G):
    ref = G["g1_legendre5"]
    assert np.max(np.abs(pulse.legendre_table(5, G["g1_xl"]) - ref)) < 1e-14
    mine = np.array([[oracle.legendre_cc(j, float(x)) for x in G["g1_xl"]] for j in range(5)])
    assert np.max(np.abs(mine - ref)) < 1e-14


@pytest.mark.parametrize("basis", ["BSpline", "Legendre", "poly", "Fourier"])
def test_g2_generate_u(oracle, G, basis):
    ref = G["g2_u_" + basis]
    T = float(G["g2_T"])
    mine = oracle.generate_u_table(G["g2_coeff"], G["g2_omegas"], T, G["g2_ts"], basis)
    assert np.max(np.abs(mine - ref)) < 1e-14
    prod = pulse.u_table(G["g2_coeff"], G["g2_omegas"], T, G["g2_ts"], basis)
    assert np.max(np.abs(prod - ref)) < 1e-14


def test_du_dcoeff_is_the_derivative_of_u_table(oracle, G):
    c, w, T, ts = G["g2_coeff"], G["g2_omegas"], float(G["g2_T"]), G["g2_ts"]
    for basis in ("BSpline", "Legendre"):
        an = pulse.du_dcoeff(c, w, T, ts, basis)
        assert np.max(np.abs(an - oracle.du_dcoeff(c, w, T, ts, basis))) < 1e-13
        eps = 1e-6
        for (i, j) in ((0, 1), (2, 4), (1, 5)):
            cp, cm = c.copy(), c.copy()
            cp[i, j] += eps
            cm[i, j] -= eps
            fd = (pulse.u_table(cp, w, T, ts, basis) - pulse.u_table(cm, w, T, ts, basis)) / (2 * eps)
            assert np.max(np.abs(fd[:, i] - an[:, i, j])) < 1e-8


def test_f_u_equals_generate_u_for_one_plain_channel(oracle, G):
    """diffqc.cc f_u with one channel, w = 0, vv[1] = 0 is generate_u up to the 1e-6 cutoff."""
    c, w, T = G["g2_coeff"], G["g2_omegas"], float(G["g2_T"])
    ts = np.linspace(0.05, T, 23, endpoint=False)
    vv = np.stack([c, np.zeros_like(c)])
    channels = [[[0.0, w[h], 0.0, float(h)]] for h in range(3)]
    for func_type, basis in ((1, "BSpline"), (0, "Legendre")):
        a = oracle.f_u_table(vv, channels, T, func_type, ts)
        b = oracle.generate_u_table(c, w, T, ts, basis)
        assert np.max(np.abs(a - b)) < 1e-13
        assert np.max(np.abs(pulse.f_u_table(vv, channels, T, func_type, ts) - a)) < 1e-14


def test_f_u_carrier_and_cutoff(oracle):
    rng = np.random.default_rng(5)
    vv = rng.normal(0, 1, (2, 3, 6))
    vv[:, 2, :] = 1e-9  # below the |N| < 1e-6 cutoff -> contributes exactly 0
    channels = [[[0, 1.5, 2.0, 0], [0, 0.7, -3.0, 1]], [[0, 2.0, 0.5, 2]]]
    ts = np.linspace(0, 2.0, 31)
    a = oracle.f_u_table(vv, channels, 2.0, 1, ts)
    assert np.all(a[:, 1] == 0.0)
    assert np.max(np.abs(pulse.f_u_table(vv, channels, 2.0, 1, ts) - a)) < 1e-14
    with pytest.raises(ValueError):
        pulse.f_u_table(vv, [[[0, 1.0, 0.0, 7]]], 2.0, 1, ts)


def test_step_counts(oracle):
    assert oracle.n_steps_py(10, 0, 2.0) == 30 and pulse.n_steps_sim_plain(10, 0, 2.0) == 30
    assert oracle.n_steps_cc(10, 2.0, 0.5) == 25 and pulse.n_steps_diffqc(10, 2.0, 0.5) == 25
    assert pulse.n_steps_sim_plain(10, 2.0, 0.5) == -5  # python twin has no abs (sim_plain.py:123)
    with pytest.raises(ValueError):
        pulse.time_grid(2.0, 0.5, -5)
    ts, dt = pulse.time_grid(0.0, 2.0, 30)
    ots, odt = oracle.time_grid(0.0, 2.0, 30)
    assert np.array_equal(ts, ots) and dt == odt


# ---- the demo problem ----------------------------------------------------------------------------------

def test_demo_problem_matches_reference_construction(G):
    p = problems.maxcut_problem()
    assert np.array_equal(p["H0"], G["demo_H0"].real) and np.all(G["demo_H0"].imag == 0)
    assert np.array_equal(p["H_cost"], G["demo_H_cost"].real)
    assert np.array_equal(p["Hs"], G["demo_Hs"].real)
    assert np.allclose(p["psi0"], G["demo_psi0"].real, atol=0, rtol=1e-16)
    assert p["T"] == float(G["demo_T"]) and np.array_equal(p["omegas"], G["demo_omegas"])
    # the aliasing quirk: H0 = sum_e (I - Z_i Z_j), not zero
    assert np.array_equal(np.diag(p["H0"]), [0, 4, 4, 4, 4, 8, 4, 4, 4, 4, 8, 4, 4, 4, 4, 0])


# ---- G3 / G4 / G6 / G7: the propagator chain -------------------------------------------------------

def test_g3_trotter_demo(oracle, G):
    fin, u, dt = oracle.trotter_sim_plain(G["demo_H0"], G["demo_Hs"], G["g3_coeff"], G["demo_omegas"],
                                          float(G["demo_T"]), "BSpline", 10, G["demo_psi0"], 0, float(G["demo_T"]))
    assert u.shape == (30, 8)
    assert np.max(np.abs(u - G["g3_u"])) < 1e-14
    assert np.max(np.abs(fin[:, 0] - G["g3_final"])) < 1e-13
    e = oracle.energies(G["demo_H_cost"], fin)[0]
    assert abs(e - float(G["g3_energy"])) < 1e-13
    assert abs(e - (-1.6146970246860874)) < 1e-12  # value recorded in SURVEY.md 8-C3


def test_g4_partial_intervals_and_mc_gradient(oracle, G):
    args = (G["demo_H0"], G["demo_Hs"], G["demo_H_cost"], G["g3_coeff"], G["demo_omegas"], float(G["demo_T"]),
            "BSpline", 10, G["demo_psi0"])
    grad, ps, phi = oracle.mc_gradient(*args, s=float(G["g4_s"]))
    assert np.max(np.abs(phi[:, 0] - G["g4_phi"])) < 1e-13
    assert np.max(np.abs(ps - G["g4_ps"])) < 1e-12
    grad, ps, phi = oracle.mc_gradient(*args, s=float(G["g4_mc_s"]))
    assert np.max(np.abs(grad - G["g4_mc_grad"])) < 1e-12
    assert np.all(G["g4_mc_grad"][:, 0] == 0)  # bump 0 never gets gradient


def test_g6_adjoint_oracle_matches_reference_fd(oracle, G):
    T = float(G["demo_T"])
    ts, dt = oracle.time_grid(0, T, 30)
    u = oracle.generate_u_table(G["g3_coeff"], G["demo_omegas"], T, ts, "BSpline")
    loss, grad_u, fin = oracle.energy_grad(G["demo_H0"], list(G["demo_Hs"]), G["demo_H_cost"], u, dt, G["demo_psi0"])
    assert abs(loss - float(G["g3_energy"])) < 1e-12
    g = pulse.grad_coeff_from_grad_u(grad_u, G["g3_coeff"], G["demo_omegas"], T, ts, "BSpline")
    assert np.max(np.abs(g - G["g6_fd_grad"])) < 2e-8  # FD-limited (eps = 1e-6)


def test_g7_batched_oracle_equals_columnwise_reference(oracle, G):
    fin, u, dt = oracle.trotter_sim_plain(G["g7_H0"], G["g7_Hs"], G["g7_coeff"], G["g7_omegas"], float(G["g7_T"]),
                                          "Legendre", int(G["g7_per_step"]), G["g7_psi0"], float(G["g7_T0"]),
                                          float(G["g7_T"]))
    assert u.shape[0] == int(7 * ((1.3 - 0.2) + 1))
    assert np.max(np.abs(fin - G["g7_final"])) < 1e-13


def test_g5_reference_trace_sanity(G):
    # recorded from the reference run itself; the GPU test replays it through the engine
    assert np.allclose(G["g5_losses_energy"] - 4.0, [-2.0000395860025693, -1.9829188702462983, -2.060480806327015,
                                                     -2.1216781468721955, -2.1870484989606345], atol=1e-12)
    assert bin(int(G["g5_cut"]))[2:] in ("1010", "101")
    assert bin(int(G["g5b_cut"]))[2:] in ("1010", "101")
    assert G["g5b_losses_energy"][-1] < 0.2  # 202 epochs reach the -4 ground energy to within 0.2


def test_oracle_energy_grad_vs_own_fd(oracle):
    H0, Hs, M = problems.random_hermitian_problem(2, 2, seed=3)
    u = np.random.default_rng(0).uniform(-1, 1, (4, 2))
    psi0 = problems.random_states(4, 2, seed=1)
    _, g, _ = oracle.energy_grad(H0, list(Hs), M, u, 0.2, psi0)
    assert np.max(np.abs(g - oracle.energy_grad_fd(H0, list(Hs), M, u, 0.2, psi0))) < 1e-8


# ---- C-ABI surface -------------------------------------------------------------------------------------

def test_library_exports_every_declared_symbol():
    """The shared library loads (no GPU needed for that) and exports what include/diffqc_hip.h declares."""
    import ctypes
    from diffquantum_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "include", "diffqc_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = set(re.findall(r"\b(dq_[a-z_0-9]+)\s*\(", text))
    assert len(declared) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().dq_version().startswith(b"diffqc_hip")


def test_no_device_fails_loudly():
    """Without a GPU the product refuses to run instead of falling back to a CPU path."""
    import torch
    from diffquantum_amd import _lib
    from diffquantum_amd.engine import Engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.DiffqcNoDevice):
        Engine(4, 1)


# ---- the C restatement of diffqc.cc -----------------------------------------------------------------

def test_c_oracle_expm_matches_scipy():
    import scipy.linalg
    from oracle import c_oracle
    rng = np.random.default_rng(0)
    for d, nrm in ((2, 1e-3), (4, 0.1), (8, 0.5), (16, 1.5), (16, 4.0), (8, 30.0)):
        H = rng.standard_normal((d, d)) + 1j * rng.standard_normal((d, d))
        A = -1j * (H + H.conj().T)
        A *= nrm / np.linalg.norm(A, 1)
        assert np.max(np.abs(c_oracle.expm(A) - scipy.linalg.expm(A))) < 2e-13 * max(1.0, nrm)


def test_c_oracle_pulse_functions(oracle, G):
    from oracle import c_oracle
    L = c_oracle.lib()
    for x in G["g1_x"]:
        for b in range(6):
            assert L.dqo_bspline(b, 6, float(x)) == oracle.bspline_cc(b, 6, float(x))
    for x in (-40.0, -32.0, -1.0, 0.0, 0.3, 32.0, 33.0):
        assert L.dqo_expit(x) == oracle.my_expit(x)
    for j in range(6):
        for x in (-1.0, -0.3, 0.0, 0.77, 1.0):
            assert abs(L.dqo_legendre(j, x) - oracle.legendre_cc(j, x)) < 1e-15


@pytest.mark.parametrize("func_type", [0, 1])
def test_c_oracle_trotter_matches_numpy_oracle(oracle, func_type):
    from oracle import c_oracle
    H0, Hs, _ = problems.random_hermitian_problem(3, 2, seed=12)
    vv = np.random.default_rng(func_type).normal(0, 1.0, (2, 3, 6))
    channels = [[[0.0, 2.0, 1.5, 0.0], [0.0, 0.8, -2.5, 1.0]], [[0.0, 3.0, 0.0, 2.0]]]
    psi0 = problems.random_states(8, 1, seed=3)[:, 0]
    for (T0, T) in ((0.0, 1.6), (1.6, 0.0)):
        a = c_oracle.trotter(H0, Hs, channels, 1.6, func_type, psi0, T0, T, 12, vv)
        b, _, _ = oracle.trotter_diffqc(H0, list(Hs), channels, 1.6, func_type, psi0, T0, T, 12, vv.tolist())
        assert np.max(np.abs(a - b)) < 1e-12


def test_c_oracle_g3_demo_forward(G):
    """The C stepper with the pinned sim_plain pulse table reproduces the reference's G3 final state."""
    from oracle import c_oracle
    out = c_oracle.propagate(G["demo_H0"], G["demo_Hs"], G["g3_u"], float(G["demo_T"]) / 30, G["demo_psi0"])
    assert np.max(np.abs(out[:, 0] - G["g3_final"])) < 1e-12


# ---- host logic of the structured engine (no GPU) ------------------------------------------------------

def test_pauli_masks_and_dense_strings():
    from diffquantum_amd.structured import pauli_dense, pauli_masks, terms_to_dense, DRIFT, OBSERVABLE
    assert pauli_masks("XIZY") == (0b1001, 0b0011)
    assert pauli_masks("iiii") == (0, 0)
    with pytest.raises(ValueError):
        pauli_masks("XA")
    # the mask action (P psi)[i] = i^{nY} (-1)^{popc((i^x)&z)} psi[i^x] reproduces the kron matrix
    rng = np.random.default_rng(0)
    for s in ("X", "Y", "Z", "XY", "ZYX", "IYIZ", "YYXZ"):
        n = len(s)
        x, z = pauli_masks(s)
        psi = rng.standard_normal(1 << n) + 1j * rng.standard_normal(1 << n)
        idx = np.arange(1 << n)
        sign = np.array([(-1) ** bin(((i ^ x) & z)).count("1") for i in idx])
        ny = bin(x & z).count("1")
        mine = (1j ** ny) * sign * psi[idx ^ x]
        assert np.allclose(mine, pauli_dense(s) @ psi, atol=1e-14), s
    H0, Hs, M = terms_to_dense([(DRIFT, "ZI", 0.5), (0, "XX", 2.0), (OBSERVABLE, "IZ", -1.0)], 2, 1)
    assert np.allclose(H0, 0.5 * np.kron(np.diag([1, -1]), np.eye(2)))
    assert np.allclose(Hs[0], 2.0 * np.kron([[0, 1], [1, 0]], [[0, 1], [1, 0]]))
    assert np.allclose(M, -np.kron(np.eye(2), np.diag([1, -1])))


def test_ising_string_form_matches_dense_generator():
    from diffquantum_amd.structured import ising_x_terms, terms_to_dense
    for n, seed in ((3, 0), (5, 3)):
        H0, Hs, M = terms_to_dense(ising_x_terms(n, seed), n, n)
        D0, Ds, DM = problems.random_ising(n, seed)
        assert np.allclose(H0, D0, atol=1e-14) and np.allclose(Hs, Ds) and np.allclose(M, DM, atol=1e-14)


def test_one_frechet_adjoint_equals_per_control_oracle(oracle):
    """The large-d organisation of the adjoint oracle (one expm_frechet per step; bench.py's CPU baseline)."""
    for n, K, B, S in ((2, 2, 3, 5), (3, 3, 9, 4)):
        H0, Hs, M = problems.random_hermitian_problem(n, K, seed=n + S)
        u = np.random.default_rng(S).uniform(-2, 2, (S, K))
        psi0 = problems.random_states(1 << n, B, seed=2)
        l0, g0, f0 = oracle.energy_grad(H0, list(Hs), M, u, 0.1, psi0)
        l1, g1, f1 = oracle.energy_grad_one_frechet(H0, list(Hs), M, u, 0.1, psi0)
        assert abs(l0 - l1) < 1e-12 and np.max(np.abs(g0 - g1)) < 1e-12 and np.max(np.abs(f0 - f1)) < 1e-13


def test_golden_qutip_stub_is_independent_of_the_product(golden_dir):
    """The goldens were captured by running the reference against tests/golden/qutip_stub: that stub must not route
    through the product's own Qobj (a bug there would be baked into G3-G6 and then matched by the product)."""
    import subprocess
    import sys
    stub = os.path.join(golden_dir, "qutip_stub")
    code = ("import sys; sys.path.insert(0, {!r}); import qutip, numpy as np\n"
            "assert not [m for m in sys.modules if m.split('.')[0] in ('diffquantum_amd', 'oracle')]\n"
            "k = qutip.Qobj([1, 1j]) / np.sqrt(2.0); Z = qutip.Qobj([[1, 0], [0, -1]]); X = qutip.Qobj([[0, 1], [1, 0]])\n"
            "assert k.shape == (2, 1) and abs(k.norm() - 1) < 1e-15 and abs(Z.matrix_element(k, k)) < 1e-15\n"
            "Y = qutip.Qobj([[0, -1j], [1j, 0]]); assert abs(Y.matrix_element(k, k) - 1) < 1e-15\n"
            "assert np.allclose(((qutip.qeye(2) + 0.5j * X) * k).full()[:, 0], (np.array([1, 1j]) + 0.5j * np.array([1j, 1])) / np.sqrt(2))\n"
            "w, v = Z.eigenstates(); assert list(w) == [-1.0, 1.0] and abs(abs(v[0].full()[1, 0]) - 1) < 1e-15\n"
            "assert abs((k.dag() * v[0]).norm() ** 2 - 0.5) < 1e-15 and X.eigenenergies()[0] == -1.0\n").format(stub)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


# ---------------------------------------------------------------------------------------------------------------
# A4: the channel model f_u against the REFERENCE'S OWN ARITHMETIC (tests/golden/fu_goldens.npz: my_expit, bspline and f_u
# sliced out of /root/reference/diffqc.cc:75-135 at generation time and compiled with g++, make_fu_goldens.py) -- the carrier
# branch cos(w t) A + sin(w t) B, both quadratures, several channels per Hamiltonian, the |N| < 1e-6 cutoff, the my_expit clamp
# ---------------------------------------------------------------------------------------------------------------

def _fu_cases(golden_dir):
    g = np.load(os.path.join(golden_dir, "fu_goldens.npz"))
    for name in g["names"]:
        name = str(name)
        ch = g[name + "/channels"]
        channels = [[list(map(float, row)) for row in h if not np.isnan(row[0])] for h in ch]
        yield name, g[name + "/vv"], channels, float(g[name + "/duration"]), int(g[name + "/func_type"]), g[name + "/ts"], g[name + "/f_u"]


def test_f_u_oracle_matches_reference_arithmetic(oracle, golden_dir):
    """The oracle's restatement of diffqc.cc:95-135 (Legendre values by the three-term recurrence instead of std::legendre:
    last-bit differences only)."""
    n = 0
    for name, vv, channels, duration, ft, ts, ref in _fu_cases(golden_dir):
        got = oracle.f_u_table(vv, channels, duration, ft, ts)
        assert got.shape == ref.shape and np.max(np.abs(got - ref)) < 2e-15, name
        n += 1
    assert n == 7


def test_f_u_product_twins_match_reference_arithmetic(golden_dir):
    """The product's two builders of the pulse table for `diffqc.trotter` (diffquantum_amd/pulse.py, vectorised numpy, and the
    compiled module's C++ f_u_table, csrc/diffqc_module.cpp) against the same reference tables."""
    from diffquantum_amd import pulse
    try:
        from diffquantum_amd import _diffqc_native as native
    except ImportError:
        native = None
    for name, vv, channels, duration, ft, ts, ref in _fu_cases(golden_dir):
        got = pulse.f_u_table(vv, channels, duration, ft, ts)
        assert np.max(np.abs(got - ref)) < 1e-14, name
        if native is not None:
            got = native.f_u_table(vv.tolist(), ts.tolist(), channels, duration, ft)
            assert np.max(np.abs(got - ref)) < 1e-14, name
    # what the cases cover (so that a regenerated file cannot silently lose a branch)
    g = np.load(os.path.join(golden_dir, "fu_goldens.npz"))
    cut = g["cutoff_and_clamp_ft0/f_u"]
    assert np.all(cut[:, 0] == 0.0) and np.all(np.abs(cut[:, 1]) > 0.0)          # below / above the 1e-6 cutoff (diffqc.cc:128)
    assert np.all(g["multi_channel_zero_row_ft1/f_u"][:, 2] == 0.0)               # a channel with A = B = 0
    assert np.any(g["bump_window_ends_nb4/f_u"][0] != 0.0)                        # t < 0: bumps 0 and 1 reach below the window
