"""``SimulatorPlain`` -- host-side mirror of the reference simulator/optimiser API
(/root/reference/sim_plain.py:14-505) for the propagator hot path, running on the MI355X engine.

Kept from the reference (same names, argument meaning, RNG consumption order, return types):
  constructor kwargs (sim_plain.py:20-22); caller-set attributes ``T``, ``omegas``, ``Pauli_M``;
  ``generate_u`` (:73-99), ``trotter`` (:119-153), ``compute_energy_grad_MC`` (:156-231),
  ``train_energy`` (:245-305), ``stochastic_measure`` (:101-117), ``multi_kron`` / ``multi_dot`` /
  ``find_state`` (:477-505), attributes ``logger``, ``final_state``, ``losses_energy``, ``spectral_coeff``.
Changed on purpose:
  * ``trotter`` evaluates the pulse closures on the whole time grid and runs ONE engine call
    (assembly + exp + chained matmul in HIP) instead of a Python loop of scipy.linalg.expm;
    ``psi0_`` may carry a batch of columns.
  * ``compute_energy_grad_MC`` evolves its 2K shifted states as one batch (same time grid, same
    propagators) instead of 2K separate solves.
  * ``gradient='adjoint'`` (new) replaces the one-sample Monte-Carlo estimator by the exact
    adjoint gradient of the discrete chain (``compute_energy_grad_adjoint``).
  * ``train_fidelity`` (sim_plain.py:414-475 integrates with qutip.mesolve) is re-based on the Trotter chain
    (SURVEY.md section 8-F4); with ``gradient='adjoint'`` all (initial, target) pairs run as one batch.
Out of scope: ``train_energy_FD`` and ``compute_energy_grad_FD`` integrate with
``qutip.mesolve`` -- not the Trotter path -- and raise NotImplementedError.
"""
from __future__ import annotations

import hashlib
import math

import numpy as np

from . import pulse
from .engine import Engine
from .logger import Logger
from .qobj import Qobj as _OwnQobj
from .qobj import as_array


_QOBJ_TYPE = None


def _qobj_type():
    """Real qutip.Qobj when the user has qutip, this package's Qobj otherwise (resolved once)."""
    global _QOBJ_TYPE
    if _QOBJ_TYPE is None:
        _QOBJ_TYPE = _OwnQobj
        try:
            import qutip
            if hasattr(qutip, "Qobj") and getattr(qutip.Qobj, "__module__", "").startswith("qutip"):
                _QOBJ_TYPE = qutip.Qobj
        except Exception:
            pass
    return _QOBJ_TYPE


def _dense(x):
    return np.ascontiguousarray(x.full() if hasattr(x, "full") else as_array(x), dtype=np.complex128)


class _Pulse:
    """u_i(t) closure with the reference call signature ``u(t, args)`` plus a vectorised form."""

    def __init__(self, sim, i, coeff):
        self.sim, self.i = sim, i
        self.coeff_i = np.array(coeff[i], dtype=np.float64)

    def table(self, ts):
        s = self.sim
        return pulse.u_table(self.coeff_i[None, :], [s.omegas[self.i]], s.T, ts, s.basis)[:, 0]

    def __call__(self, t, args=None):
        return float(self.table(np.array([t], dtype=np.float64))[0])


class SimulatorPlain(object):
    def __init__(self, n_basis=5, basis='BSpline', n_epoch=200, log_dir=None, n_step=100, lr=2e-2,
                 is_noisy=False, measure_sample_times=1000, method_name='Ours', sampling_measure=False,
                 per_step=10, gradient='mc', device=0):
        if basis not in pulse.BASES:
            raise ValueError("basis must be one of {}".format(pulse.BASES))
        if gradient not in ('mc', 'adjoint'):
            raise ValueError("gradient must be 'mc' or 'adjoint'")
        self.n_basis = n_basis
        self.basis = basis
        self.log_name = basis
        self.n_epoch = n_epoch
        self.n_step = n_step
        self.lr = lr
        self.is_noisy = is_noisy
        self.sampling_measure = sampling_measure
        self.measure_sample_times = measure_sample_times
        self.per_step = per_step
        self.gradient = gradient
        self.device = device
        self.logger = Logger(name=method_name, path=log_dir)
        self.logger.write_text("SimulatorPlain on diffquantum_amd (MI355X engine) ========")
        for key in ("n_basis", "basis", "n_epoch", "n_step", "lr", "is_noisy", "measure_sample_times",
                    "method_name", "sampling_measure", "per_step", "gradient"):
            self.logger.write_text("{}: {}".format(key, locals()[key]))
        self.my_solver = self.trotter
        self._engine = None
        self._engine_key = None
        if basis == 'BSpline':
            self.func_bsplines = [self._bump_fn(b) for b in range(n_basis)]
        if basis == 'Legendre':
            self.legendre_ps = [self._legendre_fn(j) for j in range(n_basis)]

    # ---- pulse model ---------------------------------------------------------------------
    def _bump_fn(self, b):
        n = self.n_basis
        return lambda t: float(pulse.bump_table(n, np.array([t], dtype=np.float64))[b, 0])

    def _legendre_fn(self, j):
        n = self.n_basis
        return lambda x: float(pulse.legendre_table(n, np.array([x], dtype=np.float64))[j, 0])

    def sigmoid(self, x):
        return 1 / (1 + math.exp(-x))

    def generate_u(self, i, spectral_coeff):
        """u_i(t) = omegas[i] * (2 sigmoid(sum_j c_ij phi_j(t)) - 1) as a callable ``u(t, args)``."""
        return _Pulse(self, i, np.asarray(spectral_coeff, dtype=np.float64))

    # ---- the hot path ----------------------------------------------------------------------
    def _get_engine(self, H0, Hs):
        key = hashlib.blake2b(H0.tobytes() + Hs.tobytes(), digest_size=16).digest()
        if self._engine is None or self._engine_key != key:
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(H0.shape[0], Hs.shape[0], self.device)
            self._engine.set_hamiltonians(H0, Hs)
            self._engine_key = key
            self._engine_M = None
        return self._engine

    def _split_H(self, H_):
        if not H_ or isinstance(H_[0], list):
            raise ValueError("H must be [H0, [H_1, u_1], ...] with the drift H0 first (sim_plain.py:141-143)")
        H0 = _dense(H_[0])
        mats, fns = [], []
        for h in H_[1:]:
            if not isinstance(h, list) or len(h) != 2:
                raise ValueError("every entry after H0 must be a [H_i, u_i] pair")
            mats.append(_dense(h[0]))
            fns.append(h[1])
        Hs = np.stack(mats) if mats else np.zeros((0,) + H0.shape, np.complex128)
        return H0, Hs, fns

    @staticmethod
    def _pulse_table(fns, ts):
        u = np.empty((len(ts), len(fns)), dtype=np.float64)
        ours = [k for k, f in enumerate(fns) if isinstance(f, _Pulse)]
        if ours and all(fns[k].sim is fns[ours[0]].sim for k in ours):
            # all of this simulator's closures in one vectorised evaluation (one basis table for every control)
            sim = fns[ours[0]].sim
            coeff = np.stack([fns[k].coeff_i for k in ours])
            om = [sim.omegas[fns[k].i] for k in ours]
            u[:, ours] = pulse.u_table(coeff, om, sim.T, ts, sim.basis)
        else:
            for k in ours:
                u[:, k] = fns[k].table(ts)
        for k, f in enumerate(fns):
            if not isinstance(f, _Pulse):
                u[:, k] = [f(float(t), None) for t in ts]
        return u

    def trotter(self, H_, psi0_, T0, T, **args):
        """psi(T) = prod_t exp(-i dt H(t_t)) psi(T0), n_steps = int(per_step*((T-T0)+1)), left endpoints."""
        H0, Hs, fns = self._split_H(H_)
        psi = _dense(psi0_)
        n_steps = pulse.n_steps_sim_plain(self.per_step, T0, T)
        ts, dt = pulse.time_grid(T0, T, n_steps)
        u = self._pulse_table(fns, ts)
        out = self._get_engine(H0, Hs).propagate(u, dt, psi)
        return _qobj_type()(out)

    # ---- measurements ----------------------------------------------------------------------
    def stochastic_measure(self, psi, per_Pauli=100):
        v = _dense(psi)[:, 0]
        ans = 0
        for entry in self.Pauli_M:
            weight = entry[1]
            evals, estates = entry[2]
            distr = [abs(np.vdot(_dense(es)[:, 0], v)) ** 2 for es in estates]
            res = np.random.choice(len(evals), per_Pauli, p=distr)
            for j in range(len(evals)):
                ans += weight * evals[j] * np.count_nonzero(res == j) / per_Pauli
        return ans

    def _measure(self, M, ket):
        val = self.stochastic_measure(ket) if self.sampling_measure else M.matrix_element(ket, ket)
        if self.is_noisy:
            val += np.random.normal(scale=np.abs(val.real) / 5)
        return val

    # ---- gradients -------------------------------------------------------------------------
    def _coeff_numpy(self):
        c = self.spectral_coeff
        return c.detach().numpy().copy() if hasattr(c, "detach") else np.array(c, dtype=np.float64)

    def compute_energy_grad_MC(self, M, H, initial_state, coeff=1.0):
        """One-sample estimator of (1/T) dL/dcoeff: sample s ~ U(0,T), evolve 0->s, apply
        (I +- i r H_i)/sqrt(1+r^2), evolve s->T, difference of <M>, times du_i(s)/dcoeff."""
        import torch
        s = np.random.uniform() * self.T
        if self.basis not in ('Legendre', 'BSpline'):
            raise NotImplementedError("compute_energy_grad_MC supports Legendre and BSpline (as the reference does)")
        cnp = self._coeff_numpy()
        dDdv = pulse.du_dcoeff(cnp, self.omegas, self.T, np.array([s]), self.basis)[0]  # [K][n_basis]
        H0, Hs, fns = self._split_H(H)
        K = Hs.shape[0]
        d = H0.shape[0]
        r = 1 / 2
        if self.my_solver == self.trotter and not self.sampling_measure and not self.is_noisy:
            # the whole estimator core in one device-resident call (dq_mc_shift_grad): 0 -> s, the 2K gates, s -> T as one
            # batch, <M> -- same time grids and arithmetic as the per-call path below
            eng = self._get_engine(H0, Hs)
            self._set_engine_observable(eng, M)
            na, nb = pulse.n_steps_sim_plain(self.per_step, 0, s), pulse.n_steps_sim_plain(self.per_step, s, self.T)
            (tsa, dta), (tsb, dtb) = pulse.time_grid(0, s, na), pulse.time_grid(s, self.T, nb)
            ps = eng.mc_shift_grad(self._pulse_table(fns, tsa), dta, self._pulse_table(fns, tsb), dtb,
                                   _dense(initial_state), r)[0]
            return torch.from_numpy(coeff * ps[:, None] * dDdv)
        phi = _dense(self.my_solver(H, initial_state, 0, s))
        shifted = np.empty((d, 2 * K), dtype=np.complex128)
        for i in range(K):
            hp = Hs[i] @ phi[:, 0]
            shifted[:, 2 * i] = (phi[:, 0] + r * 1.j * hp) / np.sqrt(1. + r ** 2)
            shifted[:, 2 * i + 1] = (phi[:, 0] - r * 1.j * hp) / np.sqrt(1. + r ** 2)
        kets = _dense(self.my_solver(H, _OwnQobj(shifted), s, self.T))
        Q = _qobj_type()
        grad = np.zeros((K, self.n_basis))
        for i in range(K):
            ps_p = self._measure(M, Q(kets[:, 2 * i].reshape(-1, 1)))
            ps_m = self._measure(M, Q(kets[:, 2 * i + 1].reshape(-1, 1)))
            ps = coeff * ((1 + r ** 2) / 2 / r * (ps_m - ps_p)).real
            grad[i] = ps * dDdv[i]
        return torch.from_numpy(grad)

    def _set_engine_observable(self, eng, M):
        Mn = _dense(M)
        mkey = hashlib.blake2b(Mn.tobytes(), digest_size=16).digest()
        if getattr(self, "_engine_M", None) != mkey:
            eng.set_observable(Mn)
            self._engine_M = mkey

    def compute_energy_grad_adjoint(self, M, H, initial_state, coeff=1.0):
        """Exact d<M>/dcoeff of the discrete chain by the engine's adjoint sweep (not in the reference)."""
        import torch
        H0, Hs, fns = self._split_H(H)
        eng = self._get_engine(H0, Hs)
        self._set_engine_observable(eng, M)
        n_steps = pulse.n_steps_sim_plain(self.per_step, 0, self.T)
        ts, dt = pulse.time_grid(0, self.T, n_steps)
        u = self._pulse_table(fns, ts)
        loss, grad_u, fin, _ = eng.energy_grad(u, dt, _dense(initial_state))
        g = pulse.grad_coeff_from_grad_u(grad_u, self._coeff_numpy(), self.omegas, self.T, ts, self.basis)
        self._last_adjoint = (loss, fin)
        return torch.from_numpy(coeff * g)

    # ---- training ----------------------------------------------------------------------------
    def save_plot(self, plot_name):
        return

    def train_energy(self, M, H0, Hs, initial_state):
        """Minimise <psi(T)|M|psi(T)> over the spectral coefficients with Adam."""
        import torch
        self.logger.write_text("!!!! train_energy ========")
        self.n_Hs = len(Hs)
        init = np.random.normal(0, 1e-3, [self.n_Hs, self.n_basis])
        self.spectral_coeff = torch.tensor(init, requires_grad=True)
        optimizer = torch.optim.Adam([self.spectral_coeff], lr=self.lr)
        ground = None
        self.losses_energy = []
        for epoch in range(1, self.n_epoch + 1):
            frozen = self.spectral_coeff.detach().numpy().copy()
            H = [H0] + [[Hs[i], self.generate_u(i, frozen)] for i in range(self.n_Hs)]
            final_state = self.my_solver(H, initial_state, 0, self.T)
            if self.sampling_measure:
                loss_energy = self.stochastic_measure(final_state).real
            else:
                loss_energy = M.matrix_element(final_state, final_state).real
            if self.is_noisy:
                loss_energy += np.random.normal(scale=np.abs(loss_energy.real) / 5)
            loss = loss_energy
            optimizer.zero_grad()
            if self.gradient == 'adjoint':
                grad_coeff = self.compute_energy_grad_adjoint(M, H, initial_state)
            else:
                grad_coeff = self.compute_energy_grad_MC(M, H, initial_state)
            self.spectral_coeff.grad = grad_coeff
            optimizer.step()
            if ground is None:
                ground = M.eigenenergies()[0]
            loss_energy = loss_energy - ground
            self.logger.write_text("epoch: {:04d}, loss: {}, loss_energy: {}".format(epoch, loss, loss_energy))
            self.losses_energy.append(loss_energy)
            self.final_state = final_state
        return self.spectral_coeff

    def compute_energy_grad_FD(self, *a, **k):
        raise NotImplementedError("finite differences through qutip.mesolve are outside the Trotter hot path "
                                  "(SURVEY.md section 2 row 6)")

    def train_energy_FD(self, *a, **k):
        raise NotImplementedError("train_energy_FD integrates with qutip.mesolve, not the Trotter path "
                                  "(SURVEY.md section 2 row 5)")

    def train_fidelity(self, H0, Hs, initial_states, target_states):
        """Maximise |<target_i|psi_i(T)>|^2 over the spectral coefficients.

        Engine-based counterpart of the reference's train_fidelity (sim_plain.py:414-475), which integrates with
        qutip.mesolve: here the forward solve is the Trotter chain (`my_solver`), like train_energy.
          gradient='mc'      -- the reference's loop: for every pair, M = |target><target|, loss 1 - <M>, one-sample
                                compute_energy_grad_MC(M, H, psi0, coeff=-1.0), one Adam step per pair (:436-460).
          gradient='adjoint' -- all pairs as ONE batch: exact gradient of sum_i (1 - F_i) from the engine's adjoint
                                sweep (dq_fidelity_grad), one Adam step per epoch.
        Appends the mean infidelity per epoch to `losses_energy` and returns the coefficients."""
        import torch
        self.n_Hs = len(Hs)
        init = np.random.normal(0, 1, [self.n_Hs, self.n_basis])
        self.spectral_coeff = torch.tensor(init, requires_grad=True)
        optimizer = torch.optim.Adam([self.spectral_coeff], lr=self.lr)
        Q = _qobj_type()
        self.losses_energy = []
        for epoch in range(1, self.n_epoch + 1):
            if self.gradient == 'adjoint':
                frozen = self.spectral_coeff.detach().numpy().copy()
                H = [H0] + [[Hs[j], self.generate_u(j, frozen)] for j in range(self.n_Hs)]
                H0n, Hsn, fns = self._split_H(H)
                eng = self._get_engine(H0n, Hsn)
                n_steps = pulse.n_steps_sim_plain(self.per_step, 0, self.T)
                ts, dt = pulse.time_grid(0, self.T, n_steps)
                u = self._pulse_table(fns, ts)
                psi0 = np.concatenate([_dense(p) for p in initial_states], axis=1)
                tgt = np.concatenate([_dense(p) for p in target_states], axis=1)
                loss, grad_u, fin, fid = eng.fidelity_grad(u, dt, psi0, tgt)
                g = pulse.grad_coeff_from_grad_u(grad_u, frozen, self.omegas, self.T, ts, self.basis)
                optimizer.zero_grad()
                self.spectral_coeff.grad = torch.from_numpy(g / len(initial_states))
                optimizer.step()
                mean_loss = float(loss) / len(initial_states)
                self.final_state = Q(fin)
            else:
                batch_losses = []
                for i in range(len(initial_states)):
                    frozen = self.spectral_coeff.detach().numpy().copy()
                    H = [H0] + [[Hs[j], self.generate_u(j, frozen)] for j in range(self.n_Hs)]
                    psi0, psi1 = initial_states[i], target_states[i]
                    M = psi1 * psi1.dag()
                    final_state = self.my_solver(H, psi0, 0, self.T)
                    inner_product_norm = M.matrix_element(final_state, final_state)
                    if self.is_noisy:
                        inner_product_norm += np.random.normal(scale=np.abs(inner_product_norm.real) / 5)
                    loss_fidelity = 1 - inner_product_norm
                    optimizer.zero_grad()
                    self.spectral_coeff.grad = self.compute_energy_grad_MC(M, H, psi0, coeff=-1.0)
                    optimizer.step()
                    batch_losses.append(loss_fidelity.real)
                    self.final_state = final_state
                mean_loss = float(np.array(batch_losses).mean())
            print("epoch: {:04d}, loss: {:.4f}, loss_fidelity: {:.4f}".format(epoch, mean_loss, mean_loss))
            self.losses_energy.append(mean_loss)
        return self.spectral_coeff

    # ---- helpers -------------------------------------------------------------------------------
    @staticmethod
    def multi_kron(*args):
        ret = np.array([[1.0]])
        for q in args:
            ret = np.kron(ret, q)
        return ret

    @staticmethod
    def multi_dot(*args):
        ret = args[0]
        for q in args[1:]:
            ret = np.dot(ret, q)
        return ret

    @staticmethod
    def find_state(final_state):
        amp = _dense(final_state).reshape(-1)
        prob = amp.real ** 2 + amp.imag ** 2
        return int(np.argmax(prob)), prob
