// Probe of the complex column form (zxr_* kernels of diffquantum_amd/csrc/xorexp.hip): n-qubit Ising diagonal drift, X_k controls on
// the first qubits and Y_k controls on the last ones (complex Hermitian, XOR-regular).  Checks the rows of U = exp(-i dt H) and the
// control gradients of Re tr(G^H U) against a host evaluation of the same series, and (small n) against central differences.
#include <hip/hip_runtime.h>
#include <complex>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "../diffquantum_amd/csrc/xorexp.hip"

typedef std::complex<double> cd;

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 12;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int L = argc > 3 ? atoi(argv[3]) : 7;
  const int d = 1 << n, nx = n < 8 ? n : 8, ny = n < 8 ? n : 8, K = nx + ny, R = K;
  const double dt = 1e-3;
  dq::XrTab tab;
  memset(&tab, 0, sizeof(tab));
  tab.R = R;
  tab.nd = 1;
  tab.dkid[0] = -1;
  std::vector<double> diag(d), u(K);
  std::vector<unsigned> sgn(d, 0u);
  for (int i = 0; i < d; ++i) diag[i] = sin(0.37 * i) * 30.0;
  for (int k = 0; k < K; ++k) u[k] = 3.0 + 1.0 * k;
  for (int k = 0; k < nx; ++k) { tab.xoff[k] = 1 << (n - 1 - k); tab.kid[k] = k; tab.cmag[k] = 1.0; tab.phase[k] = 0; }
  for (int k = 0; k < ny; ++k) {  // Y on qubit q = n - ny + k: H[i, i ^ e] = -i if bit(i) = 0 else +i
    const int j = nx + k, q = n - ny + k, e = 1 << (n - 1 - q);
    tab.xoff[j] = e; tab.kid[j] = j; tab.cmag[j] = 0.5 + 0.125 * k; tab.phase[j] = 1;
    for (int i = 0; i < d; ++i) if (!(i & e)) sgn[i] |= 1u << j;
  }
  const size_t nn = (size_t)d * d;
  double2 *U, *G, *xpow;
  double *du, *dgen, *part, *grad;
  unsigned* dsgn;
  (void)hipMalloc(&U, nn * 16); (void)hipMalloc(&G, nn * 16);
  (void)hipMalloc(&du, K * 8); (void)hipMalloc(&dgen, d * 8); (void)hipMalloc(&dsgn, d * 4);
  (void)hipMalloc(&part, (size_t)dq::xr_blocks(d, 1) * dq::XR_SLOTS * 8); (void)hipMalloc(&grad, K * 8);
  (void)hipMalloc(&xpow, dq::zxr_pow_elems(d, L, 1) * 16);
  (void)hipMemcpy(du, u.data(), K * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dgen, diag.data(), d * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dsgn, sgn.data(), d * 4, hipMemcpyHostToDevice);
  tab.dval = dgen;
  tab.sgn = dsgn;
  std::vector<double2> hg(nn);
  for (size_t i = 0; i < nn; ++i) { hg[i].x = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5; hg[i].y = (double)((i * 40503u) % 997) / 997.0 - 0.5; }
  (void)hipMemcpy(G, hg.data(), nn * 16, hipMemcpyHostToDevice);
  if (!dq::zxr_supported(d, L, R, 1)) { printf("unsupported shape\n"); return 1; }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto timeit = [&](const char* name, auto&& launch) {
    launch(); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    for (int r = 0; r < reps; ++r) launch();
    (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    hipError_t err = hipGetLastError();
    printf("%-44s %.4f ms/launch %s\n", name, ms / reps, err == hipSuccess ? "" : hipGetErrorString(err));
  };
  timeit("zxr_forward (rows of U = exp(-i dt H))", [&] { (void)dq::zxr_forward_launch(U, (long)nn, d, tab, du, K, dt, L, 1, 0); });
  timeit("zxr_reverse (control gradients from G)", [&] { (void)dq::zxr_reverse_launch(G, (long)nn, d, tab, du, K, dt, L, 1, part, dsgn, xpow, grad, 0, 0); });

  // host: H[i, i ^ x_j] = u c (phase ? i : 1) sigma;  rows of U via A~ = A^T = -i dt conj(H) (H Hermitian)
  auto hent = [&](const std::vector<double>& uu, int i, int j) {
    const double s = (sgn[i] >> j) & 1 ? -1.0 : 1.0;
    return uu[tab.kid[j]] * tab.cmag[j] * s * (tab.phase[j] ? cd(0, 1) : cd(1, 0));
  };
  auto applyT = [&](const std::vector<double>& uu, const std::vector<cd>& v, std::vector<cd>& o) {  // o = A~ v
    for (int i = 0; i < d; ++i) {
      cd s = cd(0, -dt) * diag[i] * v[i];
      for (int j = 0; j < R; ++j) s += cd(0, -dt) * std::conj(hent(uu, i, j)) * v[i ^ tab.xoff[j]];
      o[i] = s;
    }
  };
  auto urow = [&](const std::vector<double>& uu, int c, std::vector<cd>& row, std::vector<std::vector<cd>>* xs) {
    std::vector<cd> x(d, 0.0), y(d);
    x[c] = 1.0;
    row = x;
    if (xs) (*xs)[0] = x;
    for (int m = 1; m <= L; ++m) {
      applyT(uu, x, y);
      for (int i = 0; i < d; ++i) { x[i] = y[i] / (double)m; row[i] += x[i]; }
      if (xs && m < L) (*xs)[m] = x;
    }
  };
  std::vector<double2> hu(nn);
  (void)hipMemcpy(hu.data(), U, nn * 16, hipMemcpyDeviceToHost);
  std::vector<double> hgrad(K), rgrad(K, 0.0);
  (void)hipMemcpy(hgrad.data(), grad, K * 8, hipMemcpyDeviceToHost);
  double eu = 0, su = 0;
  std::vector<cd> row, y(d), t2(d);
  std::vector<std::vector<cd>> xs(L, std::vector<cd>(d));
  for (int c = 0; c < d; ++c) {
    urow(u, c, row, &xs);
    if (c % 97 == 0 || c == d - 1)
      for (int i = 0; i < d; ++i) {
        eu = fmax(eu, std::abs(row[i] - cd(hu[(size_t)c * d + i].x, hu[(size_t)c * d + i].y)));
        su = fmax(su, std::abs(row[i]));
      }
    // y_{L-1} = g / L, y_a = (g - A~ y_{a+1}) / (a + 1); grad_k = dt Im sum_a <y_a, H~_k x_a>, H~_k = conj(H_k)
    for (int i = 0; i < d; ++i) y[i] = cd(hg[(size_t)c * d + i].x, hg[(size_t)c * d + i].y) / (double)L;
    for (int a = L - 1; a >= 0; --a) {
      for (int j = 0; j < R; ++j) {
        cd s = 0;
        for (int i = 0; i < d; ++i) {
          const cd xv = xs[a][i ^ tab.xoff[j]];
          if (xv == cd(0, 0)) continue;
          const double sg = (sgn[i] >> j) & 1 ? -1.0 : 1.0;
          s += std::conj(y[i]) * std::conj(tab.cmag[j] * sg * (tab.phase[j] ? cd(0, 1) : cd(1, 0))) * xv;
        }
        rgrad[tab.kid[j]] += dt * s.imag();
      }
      if (a > 0) {
        applyT(u, y, t2);
        for (int i = 0; i < d; ++i) y[i] = (cd(hg[(size_t)c * d + i].x, hg[(size_t)c * d + i].y) - t2[i]) / (double)a;
      }
    }
  }
  double eg = 0, sg = 0;
  for (int k = 0; k < K; ++k) { eg = fmax(eg, fabs(rgrad[k] - hgrad[k])); sg = fmax(sg, fabs(rgrad[k])); }
  printf("U rows: max|gpu - host| = %.3e (max |U| %.3e);  gradients: max|gpu - host series| = %.3e (max |g| %.3e)\n", eu, su, eg, sg);
  bool ok = eu <= 1e-12 * su && eg <= 1e-10 * sg;
  if (n <= 9) {  // the series formula itself against central differences of L(u) = Re tr(G^H U(u))
    auto loss = [&](const std::vector<double>& uu) {
      double l = 0;
      std::vector<cd> r2;
      for (int c = 0; c < d; ++c) {
        urow(uu, c, r2, nullptr);
        for (int i = 0; i < d; ++i) l += hg[(size_t)c * d + i].x * r2[i].real() + hg[(size_t)c * d + i].y * r2[i].imag();
      }
      return l;
    };
    double ef = 0;
    for (int k = 0; k < K; k += 5) {
      std::vector<double> up = u, um = u;
      const double h = 1e-4 * (1.0 + fabs(u[k]));
      up[k] += h; um[k] -= h;
      const double fd = (loss(up) - loss(um)) / (2 * h);
      ef = fmax(ef, fabs(fd - hgrad[k]));
      printf("   control %2d: gpu %.10e  central difference %.10e\n", k, hgrad[k], fd);
    }
    printf("gradients vs central differences: max diff %.3e\n", ef);
    ok = ok && ef <= 1e-6 * sg;
  }
  printf(ok ? "PROBE OK\n" : "PROBE MISMATCH\n");
  return 0;
}
