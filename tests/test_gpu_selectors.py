"""Which path a call takes under DEFAULT options, frozen in a table.

engine.hip picks among the state-column kernels (swept states / the composed product / the wide-batch adjoint on dim columns), the
explicit-propagator chain in state or operator form, and the matrix-free route, by dimension, batch width, objective and what the
caller asks back -- with cost rules whose constants were fitted on one box.  A re-fitted constant must not move a shape onto a path no
test ran: for d in {256, 512, 1024, 4096} x batch / d in {1/4, 1, 4, 9} x objective in {energy, fidelity} x {final states asked for
or not} this sweep asserts (i) the route (dq_get_info) against the table below and (ii) loss, gradient and, where asked for, the final
states against the SAME context with state_columns = 0 (the explicit-propagator chain: oracle-checked at every one of these dimensions in
test_gpu_chain.py / test_gpu_baseline_configs.py) at 1e-10.  Reference loop: diffqc.cc:190-200; loss sim_plain.py:281.

To re-record the table after a deliberate change of the rules: DQ_SELECTOR_RECORD=1 python -m pytest tests/test_gpu_selectors.py -s
prints it (and skips the route assertion)."""
import os

import numpy as np
import pytest

from diffquantum_amd import problems

pytestmark = pytest.mark.gpu
TOL = 1e-10
DT = 1.5e-3
CHAINS = {8: (3, 60), 9: (3, 60), 10: (3, 60), 12: (3, 24)}   # chain lengths per n: the cost rules weigh sweeps over the chain against products per call
KEYS = ("state_columns", "state_columns_wide_adjoint", "state_columns_compose_segments", "state_columns_quad_nodes", "composed",
        "operator_backward", "matrix_free")

# (n, steps, batch / d numerator over 4, objective, want_final) -> the route: values of KEYS ("state_columns_compose_segments" as 0 / 1)
ROUTES = {
    (8, 3, 1, 'energy', False): (0, 0, 0, 0, 1, 1, 0),
    (8, 3, 1, 'energy', True): (0, 0, 0, 0, 1, 1, 0),
    (8, 3, 1, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (8, 3, 1, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
    (8, 3, 4, 'energy', False): (0, 0, 0, 0, 1, 1, 0),
    (8, 3, 4, 'energy', True): (0, 0, 0, 0, 1, 1, 0),
    (8, 3, 4, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (8, 3, 4, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
    (8, 3, 16, 'energy', False): (0, 1, 1, 2, 1, 0, 0),
    (8, 3, 16, 'energy', True): (0, 1, 1, 2, 1, 0, 0),
    (8, 3, 16, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (8, 3, 16, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
    (8, 3, 36, 'energy', False): (0, 1, 1, 2, 1, 0, 0),
    (8, 3, 36, 'energy', True): (0, 1, 1, 2, 1, 0, 0),
    (8, 3, 36, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (8, 3, 36, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
    (8, 60, 1, 'energy', False): (0, 0, 0, 0, 1, 1, 0),
    (8, 60, 1, 'energy', True): (0, 0, 0, 0, 1, 1, 0),
    (8, 60, 1, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (8, 60, 1, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
    (8, 60, 4, 'energy', False): (0, 0, 0, 0, 1, 1, 0),
    (8, 60, 4, 'energy', True): (0, 0, 0, 0, 1, 1, 0),
    (8, 60, 4, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (8, 60, 4, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
    (8, 60, 16, 'energy', False): (0, 1, 1, 2, 1, 0, 0),
    (8, 60, 16, 'energy', True): (0, 1, 1, 2, 1, 0, 0),
    (8, 60, 16, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (8, 60, 16, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
    (8, 60, 36, 'energy', False): (0, 1, 1, 2, 1, 0, 0),
    (8, 60, 36, 'energy', True): (0, 1, 1, 2, 1, 0, 0),
    (8, 60, 36, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (8, 60, 36, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
    (9, 3, 1, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (9, 3, 1, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (9, 3, 1, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (9, 3, 1, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (9, 3, 4, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (9, 3, 4, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (9, 3, 4, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (9, 3, 4, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (9, 3, 16, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (9, 3, 16, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (9, 3, 16, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (9, 3, 16, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (9, 3, 36, 'energy', False): (0, 1, 1, 2, 1, 0, 0),
    (9, 3, 36, 'energy', True): (0, 0, 0, 0, 1, 1, 0),
    (9, 3, 36, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (9, 3, 36, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
    (9, 60, 1, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (9, 60, 1, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (9, 60, 1, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (9, 60, 1, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (9, 60, 4, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (9, 60, 4, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (9, 60, 4, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (9, 60, 4, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (9, 60, 16, 'energy', False): (0, 1, 1, 2, 1, 0, 0),
    (9, 60, 16, 'energy', True): (0, 1, 1, 2, 1, 0, 0),
    (9, 60, 16, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (9, 60, 16, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (9, 60, 36, 'energy', False): (0, 1, 1, 2, 1, 0, 0),
    (9, 60, 36, 'energy', True): (0, 1, 1, 2, 1, 0, 0),
    (9, 60, 36, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (9, 60, 36, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
    (10, 3, 1, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (10, 3, 1, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (10, 3, 1, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (10, 3, 1, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (10, 3, 4, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (10, 3, 4, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (10, 3, 4, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (10, 3, 4, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (10, 3, 16, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (10, 3, 16, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (10, 3, 16, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (10, 3, 16, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (10, 3, 36, 'energy', False): (0, 0, 0, 0, 1, 1, 0),
    (10, 3, 36, 'energy', True): (0, 0, 0, 0, 1, 1, 0),
    (10, 3, 36, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (10, 3, 36, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
    (10, 60, 1, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (10, 60, 1, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (10, 60, 1, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (10, 60, 1, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (10, 60, 4, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (10, 60, 4, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (10, 60, 4, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (10, 60, 4, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (10, 60, 16, 'energy', False): (0, 1, 1, 2, 1, 0, 0),
    (10, 60, 16, 'energy', True): (0, 1, 1, 2, 1, 0, 0),
    (10, 60, 16, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (10, 60, 16, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (10, 60, 36, 'energy', False): (0, 1, 1, 2, 1, 0, 0),
    (10, 60, 36, 'energy', True): (0, 1, 1, 2, 1, 0, 0),
    (10, 60, 36, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (10, 60, 36, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
    (12, 3, 1, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (12, 3, 1, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (12, 3, 1, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (12, 3, 1, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (12, 3, 4, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (12, 3, 4, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (12, 3, 4, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (12, 3, 4, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (12, 3, 16, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (12, 3, 16, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (12, 3, 16, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (12, 3, 16, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (12, 3, 36, 'energy', False): (0, 0, 0, 0, 1, 1, 0),
    (12, 3, 36, 'energy', True): (0, 0, 0, 0, 1, 1, 0),
    (12, 3, 36, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (12, 3, 36, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
    (12, 24, 1, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (12, 24, 1, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (12, 24, 1, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (12, 24, 1, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (12, 24, 4, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (12, 24, 4, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (12, 24, 4, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (12, 24, 4, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (12, 24, 16, 'energy', False): (1, 0, 0, 2, 0, 0, 0),
    (12, 24, 16, 'energy', True): (1, 0, 0, 2, 0, 0, 0),
    (12, 24, 16, 'fidelity', False): (1, 0, 0, 2, 0, 0, 0),
    (12, 24, 16, 'fidelity', True): (1, 0, 0, 2, 0, 0, 0),
    (12, 24, 36, 'energy', False): (0, 1, 1, 2, 1, 0, 0),
    (12, 24, 36, 'energy', True): (0, 0, 0, 0, 1, 1, 0),
    (12, 24, 36, 'fidelity', False): (0, 0, 0, 0, 1, 1, 0),
    (12, 24, 36, 'fidelity', True): (0, 0, 0, 0, 1, 1, 0),
}


def _route(e):
    r = [int(e.info(k)) for k in KEYS]
    r[2] = 1 if r[2] > 0 else 0
    return tuple(r)


def _inputs(n, B, S):
    import torch
    dev = torch.device("cuda:0")
    d = 1 << n
    H0, Hs, M = problems.random_ising_torch(n, 30 + n, dev)
    gen = torch.Generator(device=dev)
    gen.manual_seed(100 + n)
    mk = lambda: torch.randn((d, B), dtype=torch.float64, device=dev, generator=gen) + 1j * torch.randn((d, B), dtype=torch.float64, device=dev, generator=gen)
    nrm = lambda x: (x / torch.linalg.vector_norm(x, dim=0, keepdim=True)).contiguous()
    u = np.random.default_rng(n).uniform(-1, 1, (S, n))
    return H0, Hs, M, u, nrm(mk()), nrm(mk())


@pytest.mark.parametrize("n,S", [(n, S) for n in (8, 9, 10, 12) for S in CHAINS[n]])
def test_default_routes_and_their_answers(n, S):
    from diffquantum_amd.engine import Engine
    d = 1 << n
    record = os.environ.get("DQ_SELECTOR_RECORD", "0") not in ("", "0")
    seen = {}
    for q4 in (1, 4, 16, 36):                      # batch = q4 d / 4
        B = q4 * d // 4
        H0, Hs, M, u, psi0, tgt = _inputs(n, B, S)
        with Engine(d, n) as e:
            e.use_torch_stream()
            e.set_hamiltonians(H0, Hs)
            e.set_observable(M)
            for objective in ("energy", "fidelity"):
                for want_final in (True, False):
                    if objective == "energy":
                        call = lambda: e.energy_grad(u, DT, psi0, want_final=want_final, want_energies=want_final)
                    else:
                        call = lambda: e.fidelity_grad(u, DT, psi0, tgt, want_final=want_final)
                    e.set_option("state_columns", -1)
                    l1, g1, f1, _ = call()
                    route = _route(e)
                    e.set_option("state_columns", 0)
                    l0, g0, f0, _ = call()
                    assert int(e.info("state_columns")) == 0 and int(e.info("state_columns_wide_adjoint")) == 0
                    key = (n, S, q4, objective, want_final)
                    seen[key] = route
                    scale = max(1.0, float(np.max(np.abs(g0))))
                    assert abs(l1 - l0) < TOL * B, (key, route, l1 - l0)
                    assert np.max(np.abs(g1 - g0)) < TOL * scale, (key, route, float(np.max(np.abs(g1 - g0))))
                    if want_final:
                        assert float((f1 - f0).abs().max().item()) < TOL * 10, (key, route)
                    if not record:
                        assert route == ROUTES[key], "{}: route {} (keys {}), the table says {}".format(key, route, KEYS, ROUTES[key])
        del H0, Hs, M, psi0, tgt
    if record:
        for k in sorted(seen):
            print("    {}: {},".format(k, seen[k]))
