// Native `diffqc` extension module (pybind11) over the C-ABI of libdiffqc_hip.so.
//
// The reference ships a compiled pybind11 module named `diffqc` (/root/reference/diffqc.cc:210-228) with
//     print_test(), complex_test(list[complex]), test_eigen(list[list[float]]), __version__ = "dev",
//     set_H(H0, Hs, channels, duration, func_type)            diffqc.cc:43-73
//     trotter(psi0, T0, T, per_step, vv) -> list[complex]     diffqc.cc:173-205
// This file is that module for the MI355X engine: same names, argument meaning, copy-in / fresh-list-out ownership and
// process-global state (one Hamiltonian set per process), GIL held for the call -- but the time loop (dH assembly,
// dH.exp(), exp * psi) is one dq_propagate on the GPU, and the pulse envelope f_u (diffqc.cc:95-135) is evaluated here for
// the whole time grid first.  Own code throughout; the reference's behaviour is restated, not copied:
//   * step count  n = int(per_step * (|T - T0| + 1)), dt = (T - T0) / n, samples at the LEFT end points built by repeated
//     t += dt (diffqc.cc:182-184,195,199);
//   * basis functions: Legendre P_j(2 t / duration - 1) for func_type 0, otherwise the "bspline" bump with centre
//     tau (b - 1.5), tau = 1 / (n_basis - 2), open support of half-width 1.5 tau (diffqc.cc:82-93);
//   * per channel row (unused, omega, w, idx): A = vv[0][idx] . phi, B = vv[1][idx] . phi, N = sqrt(A^2 + B^2); the row adds
//     omega (2 expit(N) - 1) / N (cos(w t) A + sin(w t) B), nothing when |N| < 1e-6; expit saturates beyond +-32
//     (diffqc.cc:75-80,108-131).
// Differences, deliberate: malformed shapes raise ValueError (the reference indexes out of bounds, diffqc.cc:101-103), and the
// extras `trotter_batch` / `f_u_table` exist.  diffquantum_amd/diffqc.py is the pure-Python twin (same engine underneath).
#include <pybind11/complex.h>
#include <pybind11/numpy.h>
#include <cstdint>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cmath>
#include <complex>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "diffqc_hip.h"

namespace py = pybind11;
typedef std::complex<double> cplx;

namespace {

struct Row { double omega, w; int idx; };

struct Global {
  dq_ctx* ctx = nullptr;
  int dim = 0, K = 0, func_type = 0;
  double duration = 1.0;
  std::vector<std::vector<Row>> channels;  // [K][rows]
} G;

[[noreturn]] void raise_status(int st) {
  const std::string msg = dq_last_error() ? dq_last_error() : "libdiffqc_hip error";
  if (st == DQ_ERR_INVALID) throw py::value_error(msg);
  if (st == DQ_ERR_NOMEM) throw std::bad_alloc();
  throw std::runtime_error(msg);
}
void check(int st) { if (st != DQ_OK) raise_status(st); }

double expit32(double x) {
  if (x > 32.0) return 1.0;
  if (x < -32.0) return 0.0;
  return 1.0 / (1.0 + std::exp(-x));
}

// phi[j] for j < n_basis at time t
void basis_at(double t, int n_basis, std::vector<double>& phi) {
  phi.assign(n_basis, 0.0);
  if (G.func_type == 0) {
    const double x = 2.0 * t / G.duration - 1.0;
    if (n_basis > 0) phi[0] = 1.0;
    if (n_basis > 1) phi[1] = x;
    for (int k = 1; k + 1 < n_basis; ++k) phi[k + 1] = ((2 * k + 1) * x * phi[k] - k * phi[k - 1]) / (k + 1);
  } else {
    const double x = t / G.duration, tau = 1.0 / (n_basis - 2), half = 1.5 * tau;
    for (int b = 0; b < n_basis; ++b) {
      const double c = tau * (b - 1.5), l = c - half, r = c + half;
      if (x < r && x > l) phi[b] = (x - l) * (x - r) / (-(half * half));
    }
  }
}

// u[a][h] for the sample times ts
std::vector<double> pulse_table(const std::vector<std::vector<std::vector<double>>>& vv, const std::vector<double>& ts) {
  if (vv.size() != 2 || vv[0].empty() || vv[0].size() != vv[1].size())
    throw py::value_error("vv must have shape [2][n_param][n_basis]");
  const int n_param = (int)vv[0].size(), n_basis = (int)vv[0][0].size();
  for (int c = 0; c < 2; ++c)
    for (const auto& row : vv[c])
      if ((int)row.size() != n_basis) throw py::value_error("vv must have shape [2][n_param][n_basis]");
  if (G.func_type != 0 && n_basis < 3) throw py::value_error("the bump basis needs n_basis >= 3");
  std::vector<double> u(ts.size() * (size_t)G.K, 0.0), phi;
  for (size_t a = 0; a < ts.size(); ++a) {
    const double t = ts[a];
    basis_at(t, n_basis, phi);
    for (int h = 0; h < G.K; ++h) {
      double acc = 0.0;
      for (const Row& row : G.channels[h]) {
        if (row.idx < 0 || row.idx >= n_param) throw py::value_error("channel parameter index out of range");
        double A = 0.0, B = 0.0;
        for (int j = 0; j < n_basis; ++j) { A += vv[0][row.idx][j] * phi[j]; B += vv[1][row.idx][j] * phi[j]; }
        const double N = std::sqrt(A * A + B * B);
        if (std::fabs(N) < 0.000001) continue;
        acc += row.omega * (2.0 * expit32(N) - 1.0) / N * (std::cos(row.w * t) * A + std::sin(row.w * t) * B);
      }
      u[a * G.K + h] = acc;
    }
  }
  return u;
}

void time_grid(double T0, double T, int per_step, std::vector<double>& ts, double& dt) {
  const int n = (int)(per_step * (std::fabs(T - T0) + 1.0));
  if (n <= 0) throw py::value_error("per_step * (|T - T0| + 1) must give at least one step");
  dt = (T - T0) / n;
  ts.resize(n);
  double t = T0;
  for (int k = 0; k < n; ++k) { ts[k] = t; t += dt; }
}

// H0: [d][d], Hs: [K][d][d] -- numpy buffers (nested lists convert on the way in: forcecast), read in place: no per-element
// Python objects, one copy to the device.  The module owns ONE engine context; the Python side borrows it (ctx_handle).
void set_H(py::array_t<cplx, py::array::c_style | py::array::forcecast> H0,
           py::array_t<cplx, py::array::c_style | py::array::forcecast> Hs,
           const std::vector<std::vector<std::vector<double>>>& channels, double duration, int func_type) {
  if (H0.ndim() != 2 || H0.shape(0) != H0.shape(1) || H0.shape(0) == 0) throw py::value_error("H0 must be a non-empty square matrix");
  const int d = (int)H0.shape(0);
  int K = 0;
  if (Hs.size() > 0) {
    if (Hs.ndim() != 3 || Hs.shape(1) != d || Hs.shape(2) != d)
      throw py::value_error("every control Hamiltonian must have the shape of H0");
    K = (int)Hs.shape(0);
  }
  if ((int)channels.size() != K) throw py::value_error("channels must have one entry per control Hamiltonian");
  std::vector<std::vector<Row>> ch(K);
  for (int k = 0; k < K; ++k)
    for (const auto& row : channels[k]) {
      if (row.size() < 4) throw py::value_error("each channel row needs at least 4 entries (unused, omega, w, idx)");
      ch[k].push_back(Row{row[1], row[2], (int)std::lround(row[3])});
    }
  if (!(duration != 0.0) || !std::isfinite(duration)) throw py::value_error("duration must be finite and non-zero");
  dq_ctx* ctx = nullptr;
  check(dq_create(&ctx, d, K, 0));
  const int st = dq_set_hamiltonians(ctx, H0.data(), K ? Hs.data() : nullptr);
  if (st != DQ_OK) { dq_destroy(ctx); raise_status(st); }
  if (G.ctx) dq_destroy(G.ctx);
  G.ctx = ctx; G.dim = d; G.K = K; G.func_type = func_type; G.duration = duration; G.channels = std::move(ch);
}

// destroy the engine context (set_H must be called again before trotter)
void close_ctx() {
  if (G.ctx) dq_destroy(G.ctx);
  G.ctx = nullptr;
}

void need_H() { if (!G.ctx) throw std::runtime_error("set_H must be called before trotter"); }

std::vector<cplx> trotter(const std::vector<cplx>& psi0, double T0, double T, int per_step,
                          const std::vector<std::vector<std::vector<double>>>& vv) {
  need_H();
  if ((int)psi0.size() != G.dim) throw py::value_error("psi0 has the wrong length for the dimension given to set_H");
  std::vector<double> ts;
  double dt;
  time_grid(T0, T, per_step, ts, dt);
  const std::vector<double> u = pulse_table(vv, ts);
  std::vector<cplx> out(psi0.size());
  check(dq_propagate(G.ctx, u.data(), (int)ts.size(), dt, psi0.data(), out.data(), 1));
  return out;
}

// extras -------------------------------------------------------------------------------------------------------------

py::array_t<cplx> trotter_batch(py::array_t<cplx, py::array::c_style | py::array::forcecast> psi0, double T0, double T,
                                int per_step, const std::vector<std::vector<std::vector<double>>>& vv) {
  need_H();
  if (psi0.ndim() != 2 || psi0.shape(0) != G.dim) throw py::value_error("psi0 must have shape [dim][batch]");
  std::vector<double> ts;
  double dt;
  time_grid(T0, T, per_step, ts, dt);
  const std::vector<double> u = pulse_table(vv, ts);
  py::array_t<cplx> out({psi0.shape(0), psi0.shape(1)});
  check(dq_propagate(G.ctx, u.data(), (int)ts.size(), dt, psi0.data(), out.mutable_data(), (int)psi0.shape(1)));
  return out;
}

// the pulse table alone (no GPU call): [n_steps][K] for the channels given to the last set_H, or for `channels` if given
py::array_t<double> f_u_table(const std::vector<std::vector<std::vector<double>>>& vv, const std::vector<double>& ts,
                              py::object channels, double duration, int func_type) {
  Global saved = G;
  struct Restore { Global& g; Global s; ~Restore() { g = s; } } restore{G, saved};
  if (!channels.is_none()) {
    const auto ch = channels.cast<std::vector<std::vector<std::vector<double>>>>();
    G.K = (int)ch.size();
    G.channels.assign(G.K, {});
    for (int k = 0; k < G.K; ++k)
      for (const auto& row : ch[k]) {
        if (row.size() < 4) throw py::value_error("each channel row needs at least 4 entries (unused, omega, w, idx)");
        G.channels[k].push_back(Row{row[1], row[2], (int)std::lround(row[3])});
      }
    G.duration = duration;
    G.func_type = func_type;
  } else if (!G.ctx) {
    throw std::runtime_error("set_H must be called first (or pass channels, duration, func_type)");
  }
  const std::vector<double> u = pulse_table(vv, ts);
  py::array_t<double> out({(py::ssize_t)ts.size(), (py::ssize_t)G.K});
  std::copy(u.begin(), u.end(), out.mutable_data());
  return out;
}

}  // namespace

PYBIND11_MODULE(_diffqc_native, m) {
  m.doc() = "diffqc on MI355X: set_H / trotter of the reference's pybind11 module over libdiffqc_hip.so";
  m.def("print_test", [] { py::print("hello"); });
  m.def("complex_test", [](const std::vector<cplx>& v) { return v; });
  m.def("test_eigen", [](const std::vector<std::vector<double>>& v) { return v; });
  m.def("set_H", &set_H, py::arg("H0"), py::arg("Hs"), py::arg("channels"), py::arg("duration"), py::arg("func_type"));
  m.def("trotter", &trotter, py::arg("psi0"), py::arg("T0"), py::arg("T"), py::arg("per_step"), py::arg("vv"));
  m.def("trotter_batch", &trotter_batch, py::arg("psi0"), py::arg("T0"), py::arg("T"), py::arg("per_step"), py::arg("vv"));
  m.def("f_u_table", &f_u_table, py::arg("vv"), py::arg("ts"), py::arg("channels") = py::none(), py::arg("duration") = 1.0,
        py::arg("func_type") = 1);
  m.def("engine_version", [] { return std::string(dq_version()); });
  // the module's engine context as an integer (0: none): the Python side wraps it (Engine.borrow) instead of opening a second one
  m.def("ctx_handle", [] { return (std::uintptr_t)G.ctx; });
  m.def("close", &close_ctx);
  m.attr("__version__") = "dev";
}
