"""Structured (matrix-free) engine vs the dense engine / the oracle at small qubit counts (SURVEY.md 8-F1:
"parity only checkable at small n against the dense path")."""
import numpy as np
import pytest

from diffquantum_amd import problems
from diffquantum_amd.structured import DRIFT, OBSERVABLE, StructuredEngine, ising_x_terms, terms_to_dense

pytestmark = pytest.mark.gpu


def random_terms(n, K, seed, n_drift=6, per_control=2, n_obs=4, max_weight=3):
    rng = np.random.default_rng(seed)

    def rstring():
        s = ["I"] * n
        for q in rng.choice(n, size=rng.integers(1, min(max_weight, n) + 1), replace=False):
            s[q] = "XYZ"[rng.integers(3)]
        return "".join(s)

    terms = [(DRIFT, rstring(), float(rng.uniform(-1, 1))) for _ in range(n_drift)]
    for k in range(K):
        terms += [(k, rstring(), float(rng.uniform(-1, 1))) for _ in range(per_control)]
    terms += [(OBSERVABLE, rstring(), float(rng.uniform(-1, 1))) for _ in range(n_obs)]
    return terms


@pytest.mark.parametrize("dtype,tol", [("c128", 1e-13), ("c64", 2e-5)])
@pytest.mark.parametrize("n,K,B", [(1, 1, 1), (3, 2, 5), (6, 3, 17), (10, 4, 20), (11, 2, 3)])
def test_apply_matches_dense(n, K, B, dtype, tol):
    terms = random_terms(n, K, seed=n * 7 + K)
    H0, Hs, M = terms_to_dense(terms, n, K)
    u = np.random.default_rng(1).uniform(-2, 2, K)
    psi = problems.random_states(1 << n, B, seed=2)
    with StructuredEngine(n, K, dtype=dtype) as e:
        e.set_terms(terms)
        got = e.apply(psi, u)
        gotM = e.apply(psi, observable=True)
    H = H0 + np.tensordot(u, Hs, axes=1)
    assert np.max(np.abs(got - H @ psi)) < tol * 10
    assert np.max(np.abs(gotM - M @ psi)) < tol * 10


@pytest.mark.parametrize("dtype,tol", [("c128", 1e-10), ("c64", 1e-4)])
@pytest.mark.parametrize("n,K,B,S,dt", [(2, 1, 3, 6, 0.1), (4, 3, 9, 12, 0.07), (6, 2, 33, 8, -0.05), (10, 3, 4, 5, 0.04)])
def test_chain_and_gradient_match_oracle(oracle, n, K, B, S, dt, dtype, tol):
    terms = random_terms(n, K, seed=n + K + S)
    H0, Hs, M = terms_to_dense(terms, n, K)
    u = np.random.default_rng(S).uniform(-1.5, 1.5, (S, K))
    psi0 = problems.random_states(1 << n, B, seed=5)
    with StructuredEngine(n, K, dtype=dtype) as e:
        e.set_terms(terms)
        fwd = e.propagate(u, dt, psi0)
        loss, grad, fin, en = e.energy_grad(u, dt, psi0)
        deg = e.info("degree")
    if n <= 6:
        rl, rg, rf = oracle.energy_grad(H0, list(Hs), M, u, dt, psi0)
    else:  # the oracle's expm_frechet is too slow at d = 1024: compare with the dense GPU engine (itself oracle-checked)
        from diffquantum_amd.engine import Engine
        with Engine(1 << n, K) as de:
            de.set_hamiltonians(H0, Hs)
            de.set_observable(M)
            rl, rg, rf, _ = de.energy_grad(u, dt, psi0)
    assert np.max(np.abs(fwd - rf)) < tol and np.max(np.abs(fin - rf)) < tol, deg
    assert abs(loss - rl) < tol * B * 4
    assert np.max(np.abs(grad - rg)) < tol * B * 4
    assert abs(en.sum() - loss) < 1e-9 * B


def test_ising_problem_equals_dense_builder():
    """The string form of the D2 synthetic problem is the same operator as problems.random_ising."""
    n = 5
    H0, Hs, M = terms_to_dense(ising_x_terms(n, seed=3), n, n)
    D0, Ds, DM = problems.random_ising(n, seed=3)
    assert np.allclose(H0, D0, atol=1e-14) and np.allclose(Hs, Ds) and np.allclose(M, DM, atol=1e-14)


def test_two_groups_and_checkpoint_stride(oracle):
    """n = 10 in complex128 uses 2 local-bit groups (p = 8); small checkpoint budget forces psi_t rebuilds."""
    n, K, B, S, dt = 10, 10, 6, 9, 0.03
    terms = ising_x_terms(n, seed=1)
    u = problems.pulse_table(K, S, S * dt, seed=2)
    psi0 = problems.random_states(1 << n, B, seed=3)
    with StructuredEngine(n, K, dtype="c128") as e:
        e.set_terms(terms)
        assert e.info("groups") == 2
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("checkpoint_stride") == 1
        e.set_option("checkpoint_bytes", (1 << n) * B * 16 * 2)
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("checkpoint_stride") == 5
    assert abs(l0 - l1) < 1e-12 and np.max(np.abs(g0 - g1)) < 1e-11 and np.max(np.abs(f0 - f1)) < 1e-13
    H0, Hs, M = terms_to_dense(terms, n, K)
    from diffquantum_amd.engine import Engine
    with Engine(1 << n, K) as de:
        de.set_hamiltonians(H0, Hs)
        de.set_observable(M)
        rl, rg, rf, _ = de.energy_grad(u, dt, psi0)
    assert np.max(np.abs(f0 - rf)) < 1e-10 and np.max(np.abs(g0 - rg)) < 1e-10 * B and abs(l0 - rl) < 1e-10 * B


@pytest.mark.parametrize("dtype,tol", [("c128", 1e-11), ("c64", 2e-5)])
@pytest.mark.parametrize("kind,B", [("ising", 6), ("ising", 5), ("random", 8)])
def test_fused_overlap_passes_equal_separate_ones(kind, B, dtype, tol):
    """Option `fuse_overlaps = 1`: the reverse-mode apply pass of a group also forms that group's control overlaps; the default
    (0) runs the separate overlap passes.  Same gradients: two groups (n = 10 / 11), odd and even batches, X / Y / Z strings of weight <= 3
    and diagonal control strings (their overlaps ride in group 0)."""
    n, K, S, dt = (10, 10, 5, 0.02) if kind == "ising" else (11, 4, 4, 0.02)
    terms = ising_x_terms(n, seed=4) if kind == "ising" else random_terms(n, K, seed=9) + [(0, "Z" + "I" * (n - 1), 0.7), (1, "IZZ" + "I" * (n - 3), -0.4)]
    u = problems.pulse_table(K, S, S * dt, seed=1)
    psi0 = problems.random_states(1 << n, B, seed=2)
    if dtype == "c64":
        psi0 = psi0.astype(np.complex64)
    with StructuredEngine(n, K, dtype=dtype) as e:
        e.set_terms(terms)
        assert e.info("groups") >= (2 if kind == "ising" else 1)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        p0 = e.info("passes")
        e.set_option("fuse_overlaps", 1)
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        p1 = e.info("passes")
    assert p1 < p0                                   # fewer streaming passes
    assert abs(l1 - l0) <= tol * B and np.max(np.abs(g1 - g0)) < tol * B and np.array_equal(f1, f0)
    assert np.max(np.abs(g0)) > 1e-3


def test_properties_at_n16():
    """n = 16 (d = 65536, dense H would be 64 GB): norm conservation, linearity, reversibility, directional derivative."""
    n, B, S, dt = 16, 8, 6, 0.01
    terms = ising_x_terms(n, seed=0)
    u = problems.pulse_table(n, S, S * dt, seed=1)
    psi0 = problems.random_states(1 << n, B, seed=2).astype(np.complex64)
    with StructuredEngine(n, n, dtype="c64") as e:
        e.set_terms(terms)
        out = e.propagate(u, dt, psi0)
        assert np.max(np.abs(np.linalg.norm(out.astype(np.complex128), axis=0) - 1)) < 1e-5
        mix = (0.6 * psi0[:, :2] + (0.3 - 0.2j) * psi0[:, 2:4]).astype(np.complex64)
        assert np.max(np.abs(e.propagate(u, dt, mix) - (0.6 * out[:, :2] + (0.3 - 0.2j) * out[:, 2:4]))) < 1e-5
        back = e.propagate(u[::-1].copy(), -dt, out)
        assert np.max(np.abs(back - psi0)) < 1e-5
        loss, grad, fin, en = e.energy_grad(u, dt, psi0)
        v = np.random.default_rng(0).standard_normal(u.shape)
        eps = 1e-2

        def L(uu):
            f = e.propagate(uu, dt, psi0)
            return float(np.real(np.vdot(f.astype(np.complex128), e.apply(f, observable=True).astype(np.complex128))))

        fd = (L(u + eps * v) - L(u - eps * v)) / (2 * eps)
        assert abs(fd - float(np.sum(grad * v))) < 2e-3 * max(1.0, abs(fd))
        assert abs(L(u) - loss) < 1e-3 * max(1.0, abs(loss))


def test_structured_errors():
    with StructuredEngine(3, 1) as e:
        with pytest.raises(ValueError):
            e.set_terms([(0, "XX", 1.0)])  # wrong string length
        with pytest.raises(ValueError):
            e.set_terms([(5, "XII", 1.0)])  # owner out of range
        with pytest.raises(ValueError):
            e.set_terms([(0, "XQI", 1.0)])
        from diffquantum_amd._lib import DiffqcError
        with pytest.raises(DiffqcError):
            e.propagate(np.zeros((1, 1)), 0.1, np.ones((8, 1)))  # terms not set
        e.set_terms([(DRIFT, "ZII", 1.0), (0, "IXI", 1.0)])
        with pytest.raises(DiffqcError):
            e.energy_grad(np.zeros((1, 1)), 0.1, np.ones((8, 1)))  # no observable


def _dense_reference(terms, n, K, u, dt, psi0):
    from diffquantum_amd.engine import Engine
    H0, Hs, M = terms_to_dense(terms, n, K)
    with Engine(1 << n, K) as de:
        de.set_hamiltonians(H0, Hs)
        de.set_observable(M)
        return de.energy_grad(u, dt, psi0.astype(np.complex128))


@pytest.mark.parametrize("dtype", ["c64", "c128"])
@pytest.mark.parametrize("workgroups", [0, 3])
@pytest.mark.parametrize("kind,n,B", [("ising", 9, 16), ("ising", 10, 16), ("ising", 11, 48), ("random", 9, 32), ("random", 10, 32), ("random", 12, 16)])
def test_dma_forms_match_dense_and_one_tile_kernels(kind, n, B, workgroups, dtype):
    """The persistent LDS-DMA passes (complex64: 2^9-row tiles; complex128, round 4: 2^8-row tiles; batch % 16 == 0) against the dense engine in complex128 and against
    the one-tile-per-workgroup kernels.  `ising`: pure X-strings -> scalar string tables, the chain form (one pass finishes an
    application and starts the next) and overlaps fused into the reverse apply; `random`: X / Y / Z strings -> the general DMA loop,
    separate DMA overlap passes.  dma_workgroups = 3 makes every workgroup walk several tiles (uneven counts) at these small sizes,
    which is what the kernels do at n = 18."""
    K = n if kind == "ising" else 3
    S, dt = 5, 0.02
    terms = ising_x_terms(n, seed=6) if kind == "ising" else random_terms(n, K, seed=12) + [(0, "Z" + "I" * (n - 1), 0.7)]
    u = problems.pulse_table(K, S, S * dt, seed=3)
    psi0 = problems.random_states(1 << n, B, seed=4).astype(np.complex64 if dtype == "c64" else np.complex128)
    rl, rg, rf, _ = _dense_reference(terms, n, K, u, dt, psi0)
    with StructuredEngine(n, K, dtype=dtype) as e:
        e.set_terms(terms)
        if workgroups:
            e.set_option("dma_workgroups", workgroups)
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        # (complex128: the separate overlap passes of groups that are not pure X-strings, and one general apply form, keep the one-tile kernels)
        assert (e.info("dma_passes") == e.info("passes") > 0) if (dtype == "c64" or kind == "ising") else (e.info("dma_passes") > 0)
        p1 = e.info("passes")
        o1 = e.propagate(u, dt, psi0)
        e.set_option("reverse_chain", 0)          # the reverse sweep as one fused pass per group and stage
        l3, g3, f3, _ = e.energy_grad(u, dt, psi0)
        p3 = e.info("passes")
        e.set_option("chain", 0)
        e.set_option("fuse_overlaps", -1)
        l2, g2, f2, _ = e.energy_grad(u, dt, psi0)
        p2 = e.info("passes")
        e.set_option("dma", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("dma_passes") == 0
        o0 = e.propagate(u, dt, psi0)
    if kind == "ising":
        assert p1 < p2                    # (chain form at two groups) + fused overlaps: fewer walks over the state
        two_groups = n > (9 if dtype == "c64" else 8)       # (a tile holds 2^9 rows in complex64, 2^8 in complex128)
        assert (p1 < p3 < p2) if two_groups else (p1 == p3)     # two groups: the reverse sweep is a chain as well
    tol = 2e-5 if dtype == "c64" else 1e-11
    for l, g, f in ((l1, g1, f1), (l3, g3, f3), (l2, g2, f2), (l0, g0, f0)):
        assert np.max(np.abs(f - rf)) < tol and abs(l - rl) < tol * B * 4 and np.max(np.abs(g - rg)) < tol * B * 4
    assert np.max(np.abs(o1 - f1)) < (1e-6 if dtype == "c64" else 1e-13) and np.max(np.abs(o1 - o0)) < tol
    assert np.max(np.abs(g1 - g0)) < tol * B and np.max(np.abs(rg)) > 1e-3


def test_dma_forms_many_tiles_per_workgroup():
    """n = 14, B = 128: 32 row tiles x 8 column chunks = 256 tile units over 40 workgroups (six or seven tiles each, both LDS buffers
    in turn); against the one-tile kernels, which the tests above tie to the dense engine."""
    n, B, S, dt = 14, 128, 3, 0.01
    terms = ising_x_terms(n, seed=2)
    u = problems.pulse_table(n, S, S * dt, seed=5)
    psi0 = problems.random_states(1 << n, B, seed=6).astype(np.complex64)
    with StructuredEngine(n, n, dtype="c64") as e:
        e.set_terms(terms)
        e.set_option("dma_workgroups", 40)
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("dma_passes") == e.info("passes")
        e.set_option("dma", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
    assert np.max(np.abs(f1 - f0)) < 2e-6 and abs(l1 - l0) < 2e-5 * B and np.max(np.abs(g1 - g0)) < 2e-5 * B
    assert np.max(np.abs(np.linalg.norm(f1.astype(np.complex128), axis=0) - 1)) < 1e-5 and np.max(np.abs(g1)) > 1e-3


def test_plan_steps_sets_the_stride_of_a_longer_chain():
    """`plan_steps`: a slice of a long chain is differentiated with the long chain's checkpoint stride (bench.py --config 5 times 20 of
    4000 steps this way); the stride changes, the numbers do not."""
    n, K, B, S, dt = 10, 10, 16, 7, 0.02
    terms = ising_x_terms(n, seed=8)
    u = problems.pulse_table(K, S, S * dt, seed=1)
    psi0 = problems.random_states(1 << n, B, seed=2).astype(np.complex64)
    with StructuredEngine(n, K, dtype="c64") as e:
        e.set_terms(terms)
        e.set_option("checkpoint_bytes", (1 << n) * B * 8 * 10)          # room for 10 states
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("checkpoint_stride") == 1
        p0 = e.info("passes")
        e.set_option("plan_steps", 40)                                    # 40 steps over 10 checkpoints: every 4th state is kept
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("checkpoint_stride") == 4 and e.info("passes") > p0  # the states in between are rebuilt by reversal
    assert np.array_equal(f0, f1) and abs(l0 - l1) < 2e-5 * B and np.max(np.abs(g0 - g1)) < 2e-5 * B


@pytest.mark.parametrize("dtype,n,B,tol", [("c128", 10, 6, 1e-11), ("c64", 11, 32, 2e-5)])
def test_group_interleaving_changes_the_tiling_not_the_numbers(dtype, n, B, tol):
    """`interleave_groups` (set before set_terms): which qubits are local to which of the two string groups is a tiling decision."""
    K, S, dt = n, 4, 0.02
    terms = ising_x_terms(n, seed=5)
    u = problems.pulse_table(K, S, S * dt, seed=1)
    psi0 = problems.random_states(1 << n, B, seed=2)
    if dtype == "c64":
        psi0 = psi0.astype(np.complex64)
    res = []
    for v in (0, 1, 2, 3):
        with StructuredEngine(n, K, dtype=dtype) as e:
            e.set_option("interleave_groups", v)
            e.set_terms(terms)
            assert e.info("groups") == 2
            res.append(e.energy_grad(u, dt, psi0))
    l0, g0, f0, _ = res[0]
    for l, g, f, _ in res[1:]:
        assert abs(l - l0) <= tol * B and np.max(np.abs(g - g0)) < tol * B and np.max(np.abs(f - f0)) < tol
    assert np.max(np.abs(g0)) > 1e-3


@pytest.mark.parametrize("dtype,B,tol", [("c128", 6, 1e-11), ("c64", 16, 2e-5)])
def test_two_level_checkpoints(dtype, B, tol):
    """When the states of a chain do not all fit but steps / s + m s of them do, every s-th state is kept and a segment at a time is
    recomputed forward WITH the Horner iterates of its steps stored: no reversal chain, no separate iterates chain."""
    n, K, S, dt = 10, 10, 120, 0.01
    terms = ising_x_terms(n, seed=11)
    u = problems.pulse_table(K, S, S * dt, seed=1)
    psi0 = problems.random_states(1 << n, B, seed=2)
    if dtype == "c64":
        psi0 = psi0.astype(np.complex64)
    sb = (1 << n) * B * (16 if dtype == "c128" else 8)
    with StructuredEngine(n, K, dtype=dtype) as e:
        e.set_terms(terms)
        la, ga, fa, _ = e.energy_grad(u, dt, psi0)                 # everything fits: a checkpoint per step
        assert e.info("checkpoint_stride") == 1 and e.info("two_level") == 0
        m = int(e.info("degree"))
        s2 = max(2, round((S / m) ** 0.5))
        need = -(-S // s2) + m * s2                                 # kept states + one segment with its iterates
        assert need + 1 < S
        e.set_option("checkpoint_bytes", sb * (need + 1))
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("two_level") == 1 and e.info("checkpoint_stride") == s2
        p1 = e.info("passes")
        assert e.info("host_checkpoints") == 0
        # the pools hold what this plan uses and nothing of the plan before it (S per-step checkpoints were released)
        assert e.info("checkpoint_pool_bytes") <= sb * (need + 1)
        e.set_option("checkpoint_bytes", sb * (m * s2 + 2))          # room for one segment, not for the kept states: those go to the host
        l2, g2, f2, _ = e.energy_grad(u, dt, psi0)
        assert e.info("two_level") == 1 and e.info("host_checkpoints") == 1 and e.info("checkpoint_stride") == s2 and e.info("passes") == p1
        assert e.info("checkpoint_pool_bytes") <= sb * (m * s2 + 2)   # the device-side kept states are gone: segment + staging buffer only
        assert e.info("pinned_host_bytes") == sb * -(-S // s2)       # the kept states, in pinned host memory
        e.set_option("host_checkpoint_bytes", sb * 2)                 # a host that will not pin that much: one level, no failure
        l4, g4, f4, _ = e.energy_grad(u, dt, psi0)
        assert e.info("two_level") == 0 and e.info("host_checkpoints") == 0
        assert abs(l4 - la) <= tol * B and np.max(np.abs(g4 - ga)) < tol * B
        e.set_option("host_checkpoint_bytes", 0)
        e.set_option("host_checkpoints", 0)
        l3, g3, f3, _ = e.energy_grad(u, dt, psi0)
        assert e.info("two_level") == 0 and e.info("checkpoint_stride") == -(-S // (m * s2 + 2))
        assert e.info("checkpoint_pool_bytes") <= sb * (m * s2 + 2)   # one level: the segment pools and the staging buffer were released
        e.set_option("checkpoint_bytes", 0)                          # default budget (70 % of free + held): every state fits again
        ld, gd, fd, _ = e.energy_grad(u, dt, psi0)
        assert e.info("checkpoint_stride") == 1 and e.info("checkpoint_pool_bytes") == sb * S
        assert abs(ld - la) <= tol * B and np.max(np.abs(gd - ga)) < tol * B
        e.set_option("checkpoint_bytes", sb * (need + 1))
        e.set_option("two_level", 0)
        l0, g0, f0, _ = e.energy_grad(u, dt, psi0)
        assert e.info("two_level") == 0 and e.info("checkpoint_stride") == -(-S // (need + 1)) and e.info("passes") > p1
    for l, g, f in ((l1, g1, f1), (l2, g2, f2), (l3, g3, f3), (l0, g0, f0)):
        assert abs(l - la) <= tol * B and np.max(np.abs(g - ga)) < tol * B and np.max(np.abs(f - fa)) < tol
    assert np.max(np.abs(ga)) > 1e-3


@pytest.mark.parametrize("dtype,n,B,tol", [("c128", 10, 16, 1e-12), ("c64", 11, 32, 2e-6), ("c64", 10, 16, 2e-6)])
def test_step_handover_changes_the_passes_not_the_numbers(dtype, n, B, tol):
    """`step_handover` (round 5, default on): the pass that finishes step t's Horner chain also starts step t + 1's, with step t + 1's
    coefficients and diagonal from a second set -- a link where a finish and a start pass ran.  One walk less per step in the forward
    pass and in a segment's recomputation; the same states, loss and gradient under every checkpoint plan (all states kept / two
    levels / one level with reversals), in `propagate` too, and against the dense engine."""
    K, S, dt = n, 60, 0.01
    terms = ising_x_terms(n, seed=41)
    u = problems.pulse_table(K, S, S * dt, seed=42)
    psi0 = problems.random_states(1 << n, B, seed=43)
    if dtype == "c64":
        psi0 = psi0.astype(np.complex64)
    sb = (1 << n) * B * (16 if dtype == "c128" else 8)
    rl, rg, rf, _ = _dense_reference(terms, n, K, u, dt, psi0)
    out = {}
    for hand in (1, 0):
        with StructuredEngine(n, K, dtype=dtype) as e:
            e.set_terms(terms)
            assert e.info("groups") == 2
            e.set_option("step_handover", hand)
            res = [e.energy_grad(u, dt, psi0)]                       # every state kept
            passes = [e.info("passes")]
            assert e.info("checkpoint_stride") == 1
            m = int(e.info("degree"))
            s2 = max(2, round((S / m) ** 0.5))
            e.set_option("checkpoint_bytes", sb * (-(-S // s2) + m * s2 + 1))
            res.append(e.energy_grad(u, dt, psi0))                   # two levels: segments recomputed with their iterates
            passes.append(e.info("passes"))
            assert e.info("two_level") == 1 and e.info("checkpoint_stride") == s2
            e.set_option("two_level", 0)
            e.set_option("checkpoint_bytes", sb * 6)
            res.append(e.energy_grad(u, dt, psi0))                   # one level: reversal chains in the backward pass
            passes.append(e.info("passes"))
            assert e.info("two_level") == 0 and e.info("checkpoint_stride") > 1
            fwd = e.propagate(u, dt, psi0)
            passes.append(e.info("passes"))
            assert e.info("dma_passes") == e.info("passes")
            if hand:   # the reverse chain's last stage with its (dead) store of zbar_m: the same gradient
                e.set_option("reverse_keep_last", 1)
                lk, gk, fk, _ = e.energy_grad(u, dt, psi0)
                l1, g1, f1_, _ = res[2]
                assert lk == l1 and np.array_equal(gk, g1) and np.array_equal(fk, f1_)
        out[hand] = (res, passes, fwd)
    (r1, p1, f1), (r0, p0, f0) = out[1], out[0]
    assert p1[0] == p0[0] - (S - 1) and p1[3] == p0[3] - (S - 1) and p1[2] == p0[2] - (S - 1)   # forward pass: one walk less per step but the last
    assert p1[1] < p0[1] - (S - 1)                                                             # and in the segments' recomputation
    for (l, g, f, _), (lo, go, fo, _) in zip(r1, r0):
        assert abs(l - lo) <= tol * B and np.max(np.abs(g - go)) < tol * B and np.max(np.abs(f - fo)) < tol
    assert np.max(np.abs(f1 - f0)) < tol and np.max(np.abs(f1 - r1[0][2])) < tol
    big = 1e-10 if dtype == "c128" else 1e-4
    l, g, f, _ = r1[1]
    assert np.max(np.abs(f - rf)) < big and abs(l - rl) < big * B and np.max(np.abs(g - rg)) < big * B and np.max(np.abs(rg)) > 1e-3


@pytest.mark.parametrize("n", [10, 12])
def test_fp32_gradient_over_the_full_4000_step_chain(n):
    """configs[4]'s kernels over configs[4]'s chain length: 4000 steps, 32 states, complex64, the LDS-DMA forms with both checkpoint
    levels in use -- against the complex128 dense engine on the same inputs (north_star's fp32 tolerance: 1e-4).  What this
    separates: fp32 accumulation over 4000 steps (loss sums, overlaps, lambda) from the finite-difference noise of the n = 18
    directional-derivative check (profiles/r3_config5_full_chain.txt: 1e-2 at B = 512)."""
    K, S, B, dt = n, 4000, 32, 1e-3
    terms = ising_x_terms(n, seed=21)
    u = problems.pulse_table(K, S, S * dt, seed=22)
    psi0 = problems.random_states(1 << n, B, seed=23).astype(np.complex64)
    rl, rg, rf, _ = _dense_reference(terms, n, K, u, dt, psi0)
    sb = (1 << n) * B * 8
    with StructuredEngine(n, K, dtype="c64") as e:
        e.set_terms(terms)
        e.propagate(u, dt, psi0)           # (the degree of the 4000-step chain: the polynomial's per-step bound is chain_tol / steps there)
        m = int(e.info("degree"))
        s2 = max(2, round((S / m) ** 0.5))
        e.set_option("checkpoint_bytes", sb * (-(-S // s2) + m * s2 + 1))     # kept states + one segment with its iterates: two levels
        l1, g1, f1, _ = e.energy_grad(u, dt, psi0)
        assert e.info("two_level") == 1 and e.info("checkpoint_stride") == s2 and e.info("dma_passes") == e.info("passes") > 0
    gmax = float(np.max(np.abs(rg)))
    rel = float(np.max(np.abs(g1 - rg))) / gmax
    assert gmax > 1e-4
    assert np.max(np.abs(f1 - rf)) < 1e-4, np.max(np.abs(f1 - rf))
    assert abs(l1 - rl) < 1e-4 * B, abs(l1 - rl)
    assert rel < 1e-4, rel


@pytest.mark.parametrize("dtype,n,B,tol", [("c128", 10, 6, 1e-10), ("c128", 11, 16, 1e-10), ("c64", 11, 32, 1e-4)])
def test_economized_step_polynomial_matches_taylor_and_the_dense_engine(dtype, n, B, tol):
    """The matrix-free step in Horner form with the economized polynomial's ratios (option polynomial = 1, the default; csrc/structured.hip
    plan_degree): fewer stages than Taylor at the same bound, the chain, its reversals (checkpoint stride 3) and the reverse mode of the
    recursion against the dense engine's result, and against the Taylor form of the same engine."""
    K, S, dt = n, 12, 0.01
    terms = ising_x_terms(n, seed=31)
    u = problems.pulse_table(K, S, S * dt, seed=32)
    psi0 = problems.random_states(1 << n, B, seed=33)
    if dtype == "c64":
        psi0 = psi0.astype(np.complex64)
    rl, rg, rf, _ = _dense_reference(terms, n, K, u, dt, psi0)
    out = {}
    with StructuredEngine(n, K, dtype=dtype) as e:
        e.set_terms(terms)
        e.set_option("checkpoint_bytes", (1 << n) * B * (16 if dtype == "c128" else 8) * 4)      # 4 states: stride 3, reversals in between
        for poly in (1, 0):
            e.set_option("polynomial", poly)
            loss, grad, fin, _ = e.energy_grad(u, dt, psi0)
            assert e.info("checkpoint_stride") >= 2
            out[poly] = (loss, grad, fin, e.info("degree"), e.info("taylor_degree"), e.propagate(u, dt, psi0))
    assert out[0][3] == out[0][4] == out[1][4] and out[1][3] < out[0][3], (out[1][3], out[0][3])
    for poly in (1, 0):
        loss, grad, fin, _, _, fwd = out[poly]
        assert np.max(np.abs(fin - rf)) < tol and np.max(np.abs(fwd - rf)) < tol
        assert abs(loss - rl) < tol * B and np.max(np.abs(grad - rg)) < tol * max(1.0, float(np.max(np.abs(rg)))) * 4
    assert np.max(np.abs(rg)) > 1e-4
