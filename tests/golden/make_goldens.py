"""Generate tests/golden/reference_goldens.npz by IMPORTING the reference in place.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_goldens.py            # regenerate and write the .npz
    python tests/golden/make_goldens.py --check    # regenerate, compare with the committed .npz, write nothing

The reference's sim_plain.py / demo_maxcut.py are imported / executed unmodified from
/root/reference with (i) tests/golden/qutip_stub on sys.path (qutip is not installed; the stub is a
self-contained numpy Qobj that imports nothing from diffquantum_amd, so the goldens are independent of the
product's own Qobj) and (ii) a subclass that only redirects the log directory
(the reference tree is read-only) and optionally shortens n_epoch.  Only inputs and outputs
(arrays) are stored -- no reference source.

Golden sets:
  G1 basis tables, G2 generate_u for the four bases, G3 trotter on the demo problem,
  G4 partial-interval / gate runs + compute_energy_grad_MC, G5 5-epoch and 202-epoch training traces,
  G6 finite-difference gradient of G3's loss through the reference trotter,
  G7 reference trotter column-by-column on a random complex 3-qubit problem (pins the batched path).
"""
import os
import runpy
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, os.path.join(HERE, "qutip_stub"))
sys.path.insert(0, REF)

import qutip as qp  # noqa: E402  (the stub)
import sim_plain as ref_sim  # noqa: E402  (the reference, in place)
import torch  # noqa: E402

TMP = tempfile.mkdtemp(prefix="dq_goldens_")
os.makedirs(os.path.join(TMP, "text"), exist_ok=True)
RefSim = ref_sim.SimulatorPlain
CAPTURE = {}


class PatchedSim(RefSim):
    """Reference class with the log directory redirected (reference tree is read-only)."""

    n_epoch_override = None

    def __init__(self, *a, **k):
        k.setdefault("log_dir", TMP)
        if PatchedSim.n_epoch_override is not None:
            k["n_epoch"] = PatchedSim.n_epoch_override
        super().__init__(*a, **k)
        CAPTURE["sim"] = self

    def train_energy(self, M, H0, Hs, initial_state):
        CAPTURE["args"] = (M, H0, Hs, initial_state)
        return super().train_energy(M, H0, Hs, initial_state)


def run_demo(n_epoch, seed):
    PatchedSim.n_epoch_override = n_epoch
    ref_sim.SimulatorPlain = PatchedSim
    np.random.seed(seed)
    torch.manual_seed(seed)
    try:
        runpy.run_path(os.path.join(REF, "demo_maxcut.py"), run_name="__main__")
    finally:
        ref_sim.SimulatorPlain = RefSim
    return CAPTURE["sim"], CAPTURE["args"]


out = {}

# ---------------- G1: basis tables -------------------------------------------------------
xs = np.concatenate([np.linspace(-0.25, 1.25, 61), [0.0, 0.125, 0.25, 0.375, 0.5, 0.625, 0.75, 0.875, 1.0]])
s6 = PatchedSim(n_basis=6, basis="BSpline")
out["g1_x"] = xs
out["g1_bspline6"] = np.array([[f(float(x)) for x in xs] for f in s6.func_bsplines])
s4 = PatchedSim(n_basis=4, basis="BSpline")
out["g1_bspline4"] = np.array([[f(float(x)) for x in xs] for f in s4.func_bsplines])
sl = PatchedSim(n_basis=5, basis="Legendre")
xl = np.linspace(-1, 1, 41)
out["g1_xl"] = xl
out["g1_legendre5"] = np.array([[float(p(x)) for x in xl] for p in sl.legendre_ps])

# ---------------- G2: generate_u ---------------------------------------------------------
rng = np.random.default_rng(11)
g2_coeff = rng.normal(0, 0.8, (3, 6))
g2_omegas = np.array([np.pi, 1.5, 0.25])
g2_T = 1.7
g2_ts = np.concatenate([np.linspace(0, g2_T, 35), [0.3, 1.0]])
out.update(g2_coeff=g2_coeff, g2_omegas=g2_omegas, g2_T=g2_T, g2_ts=g2_ts)
for basis in ["BSpline", "Legendre", "poly", "Fourier"]:
    s = PatchedSim(n_basis=6, basis=basis)
    s.T = g2_T
    s.omegas = list(g2_omegas)
    fs = [s.generate_u(i, g2_coeff) for i in range(3)]
    out["g2_u_" + basis] = np.array([[f(float(t), None) for f in fs] for t in g2_ts])

# ---------------- G5 (first: it gives us the demo problem) -------------------------------
sim5, (Mq, H0q, Hsq, psi0q) = run_demo(5, 0)
out["demo_H_cost"] = Mq.full()
out["demo_H0"] = H0q.full()
out["demo_Hs"] = np.stack([h.full() for h in Hsq])
out["demo_psi0"] = psi0q.full()[:, 0]
out["demo_T"] = float(sim5.T)
out["demo_omegas"] = np.array(sim5.omegas, dtype=float)
out["g5_losses_energy"] = np.array(sim5.losses_energy, dtype=float)
out["g5_coeff"] = sim5.spectral_coeff.detach().numpy().copy()
out["g5_final_state"] = sim5.final_state.full()[:, 0]
st, prob = sim5.find_state(sim5.final_state)
out["g5_cut"] = np.array(st)
out["g5_prob"] = np.asarray(prob, dtype=float)

sim202, _ = run_demo(202, 0)
out["g5b_losses_energy"] = np.array(sim202.losses_energy, dtype=float)
out["g5b_coeff"] = sim202.spectral_coeff.detach().numpy().copy()
st, prob = sim202.find_state(sim202.final_state)
out["g5b_cut"] = np.array(st)
out["g5b_prob"] = np.asarray(prob, dtype=float)

# ---------------- G3: trotter on the demo problem ----------------------------------------
sim = PatchedSim(lr=2e-2, n_basis=6, n_epoch=1)
sim.T = float(sim5.T)
sim.omegas = list(sim5.omegas)
g3_coeff = np.random.default_rng(7).normal(0, 0.5, (8, 6))
out["g3_coeff"] = g3_coeff


def build_H(coeff):
    H = [H0q]
    for i in range(8):
        H.append([Hsq[i], sim.generate_u(i, coeff)])
    return H


H = build_H(g3_coeff)
fin = sim.trotter(H, psi0q, 0, sim.T)
out["g3_final"] = fin.full()[:, 0]
out["g3_energy"] = np.array(Mq.matrix_element(fin, fin).real)
n_steps = int(sim.per_step * ((sim.T - 0) + 1))
dt = sim.T / n_steps
ts = []
t = 0
for k in range(n_steps):
    ts.append(t)
    t += dt
out["g3_ts"] = np.array(ts)
out["g3_u"] = np.array([[H[i + 1][1](tt, None) for i in range(8)] for tt in ts])

# ---------------- G4: partial intervals, gates, MC gradient -------------------------------
s_fix = 0.7
phi = sim.trotter(H, psi0q, 0, s_fix)
out["g4_s"] = np.array(s_fix)
out["g4_phi"] = phi.full()[:, 0]
r = 0.5
ps_list, ketp, ketm = [], [], []
for i in range(8):
    gate_p = (qp.qeye(16) + r * 1.0j * H[i + 1][0]) / np.sqrt(1.0 + r ** 2)
    gate_m = (qp.qeye(16) - r * 1.0j * H[i + 1][0]) / np.sqrt(1.0 + r ** 2)
    kp = sim.trotter(H, gate_p * phi, s_fix, sim.T)
    km = sim.trotter(H, gate_m * phi, s_fix, sim.T)
    ketp.append(kp.full()[:, 0])
    ketm.append(km.full()[:, 0])
    ps_p = Mq.matrix_element(kp, kp)
    ps_m = Mq.matrix_element(km, km)
    ps_list.append(((1 + r ** 2) / 2 / r * (ps_m - ps_p)).real)
out["g4_ket_p"] = np.array(ketp)
out["g4_ket_m"] = np.array(ketm)
out["g4_ps"] = np.array(ps_list)

sim.n_Hs = 8
sim.spectral_coeff = torch.tensor(g3_coeff, requires_grad=True)
H_mc = build_H(sim.spectral_coeff.detach().numpy().copy())
np.random.seed(123)
s_drawn = np.random.uniform() * sim.T
np.random.seed(123)
g_mc = sim.compute_energy_grad_MC(Mq, H_mc, psi0q)
out["g4_mc_s"] = np.array(s_drawn)
out["g4_mc_grad"] = g_mc.numpy().copy()

# ---------------- G6: FD gradient of the G3 loss ------------------------------------------


def loss_of(coeff):
    f = sim.trotter(build_H(coeff), psi0q, 0, sim.T)
    return Mq.matrix_element(f, f).real


eps = 1e-6
fd = np.zeros_like(g3_coeff)
for i in range(8):
    for j in range(6):
        cp = g3_coeff.copy(); cp[i, j] += eps
        cm = g3_coeff.copy(); cm[i, j] -= eps
        fd[i, j] = (loss_of(cp) - loss_of(cm)) / (2 * eps)
out["g6_fd_grad"] = fd

# ---------------- G7: random complex problem, Legendre basis, several columns --------------
rng = np.random.default_rng(77)
d7 = 8


def herm():
    a = rng.standard_normal((d7, d7)) + 1j * rng.standard_normal((d7, d7))
    a = a + a.conj().T
    return a / np.linalg.norm(a, 2)


H0_7 = herm()
Hs_7 = np.stack([herm() for _ in range(3)])
coeff7 = rng.normal(0, 0.7, (3, 5))
s7 = PatchedSim(n_basis=5, basis="Legendre", per_step=7)
s7.T = 1.3
s7.omegas = [2.0, np.pi, 0.5]
H7 = [qp.Qobj(H0_7)] + [[qp.Qobj(Hs_7[i]), s7.generate_u(i, coeff7)] for i in range(3)]
psi7 = rng.standard_normal((d7, 6)) + 1j * rng.standard_normal((d7, 6))
psi7 /= np.linalg.norm(psi7, axis=0)
fin7 = np.stack([s7.trotter(H7, qp.Qobj(psi7[:, b]), 0.2, 1.3).full()[:, 0] for b in range(6)], axis=1)
out.update(g7_H0=H0_7, g7_Hs=Hs_7, g7_coeff=coeff7, g7_T=1.3, g7_T0=0.2, g7_omegas=np.array(s7.omegas),
           g7_per_step=np.array(7), g7_psi0=psi7, g7_final=fin7)

path = os.path.join(HERE, "reference_goldens.npz")
if "--check" in sys.argv[1:]:
    old = np.load(path)
    assert sorted(old.files) == sorted(out), (sorted(set(old.files) ^ set(out)))
    worst, same = 0.0, 0
    for k in sorted(out):
        a, b = np.asarray(out[k]), old[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        if np.array_equal(a, b):
            same += 1
        else:
            dlt = float(np.max(np.abs(a - b)))
            worst = max(worst, dlt)
            print("  differs: {:24s} max |new - committed| = {:.3e}".format(k, dlt))
    print("check: {} of {} arrays bit-identical to the committed fixture, worst difference {:.3e}".format(same, len(out), worst))
    sys.exit(0 if worst < 1e-12 else 1)
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes;", len(out), "arrays")
print("G3 energy", float(out["g3_energy"]), "G5 losses", out["g5_losses_energy"], "cut", bin(int(out["g5_cut"])),
      "cut202", bin(int(out["g5b_cut"])))
