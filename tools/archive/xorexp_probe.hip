// Probe of the column-form step exponential for XOR-regular sparse generators (diffquantum_amd/csrc/xorexp.hip) at d = 2^n:
// an n-qubit Ising diagonal + X_k flips.  Checks U rows and the control gradients against a host evaluation of the same
// recursions (sparse matrix-vector products in double) and prints per-launch times.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include "../diffquantum_amd/csrc/xorexp.hip"

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 12;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int q = argc > 3 ? atoi(argv[3]) : 3;
  const int with_sign = argc > 4 ? atoi(argv[4]) : 0;
  const int d = 1 << n, K = n, R = n, np = 2 * q + 1;
  const double dt = 1e-3;
  dq::XrTab tab;
  memset(&tab, 0, sizeof(tab));
  tab.R = R;
  tab.nd = with_sign ? 2 : 1;  // the drift's diagonal; in the signed variant control 0 has a diagonal part too
  std::vector<double> diag(2 * (size_t)d), u(K);
  std::vector<unsigned> sgn(d, 0u);
  for (int i = 0; i < d; ++i) { diag[i] = sin(0.37 * i) * 3000.0; diag[d + i] = (__builtin_popcount(i & 0x333) & 1) ? -0.75 : 0.75; }
  for (int k = 0; k < K; ++k) u[k] = 300.0 + 100.0 * k;
  tab.dkid[0] = -1;
  tab.dkid[1] = 0;
  for (int b = 0; b < n; ++b) { tab.xoff[b] = 1 << (n - 1 - b); tab.kid[b] = b; tab.cmag[b] = 1.0 + 0.25 * (b & 1); }
  if (with_sign) {  // a sign pattern that keeps the diagonals symmetric: parity of some bits of i outside x_j; one two-bit offset
    tab.xoff[n - 1] = 3;
    for (int i = 0; i < d; ++i)
      for (int j = 0; j < R; ++j)
        if (__builtin_popcount(i & ~tab.xoff[j] & 0x5a5) & 1) sgn[i] |= 1u << j;
  }
  dq::XrPoly pol;
  memset(&pol, 0, sizeof(pol));
  pol.q = q;
  {
    double f = 1.0;
    for (int m = 0; m <= 2 * q + 1; ++m) {
      if (m) f *= m;
      if (m & 1) pol.sc[m / 2] = ((m / 2) & 1 ? -1.0 : 1.0) / f;
      else pol.cc[m / 2] = ((m / 2) & 1 ? -1.0 : 1.0) / f;
    }
  }
  const size_t nn = (size_t)d * d;
  double2 *U, *G;
  double *du, *dgen, *part, *grad;
  unsigned* dsgn;
  (void)hipMalloc(&U, nn * 16); (void)hipMalloc(&G, nn * 16);
  (void)hipMalloc(&du, K * 8); (void)hipMalloc(&dgen, 2 * d * 8); (void)hipMalloc(&dsgn, d * 4);
  (void)hipMalloc(&part, (size_t)dq::xr_blocks(d, 1) * dq::XR_SLOTS * 8); (void)hipMalloc(&grad, K * 8);
  (void)hipMemcpy(du, u.data(), K * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dgen, diag.data(), 2 * d * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dsgn, sgn.data(), d * 4, hipMemcpyHostToDevice);
  tab.dval = dgen;
  tab.sgn = with_sign ? dsgn : nullptr;
  std::vector<double2> hg(nn);
  for (size_t i = 0; i < nn; ++i) { hg[i].x = (double)((i * 2654435761u) % 1000) / 1000.0 - 0.5; hg[i].y = (double)((i * 40503u) % 997) / 997.0 - 0.5; }
  (void)hipMemcpy(G, hg.data(), nn * 16, hipMemcpyHostToDevice);
  if (!dq::xr_supported(d, q, R, tab.nd, 8)) { printf("unsupported shape\n"); return 1; }

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
  double* xpow; (void)hipMalloc(&xpow, dq::xr_pow_elems(d, q, 1) * 8);
  timeit("xr_forward (U = cos X + i sin X)", [&] { (void)dq::xr_forward_launch(U, (long)nn, d, tab, du, K, dt, pol, 1, 0); });
  timeit("xr_reverse (control gradients from G)", [&] { (void)dq::xr_reverse_launch(G, (long)nn, d, tab, du, K, dt, pol, 1, part, dsgn, xpow, grad, 0, 0); });

  // host evaluation
  auto val = [&](int i, int j) { return (sgn[i] >> j) & 1 ? -tab.cmag[j] : tab.cmag[j]; };
  auto apply = [&](const std::vector<double>& v, std::vector<double>& o) {
    for (int i = 0; i < d; ++i) {
      double s = 0;
      for (int jd = 0; jd < tab.nd; ++jd) s += -dt * (tab.dkid[jd] < 0 ? 1.0 : u[tab.dkid[jd]]) * diag[(size_t)jd * d + i] * v[i];
      for (int j = 0; j < R; ++j) s += -dt * u[tab.kid[j]] * val(i, j) * v[i ^ tab.xoff[j]];
      o[i] = s;
    }
  };
  auto coef = [&](int m) { return (m & 1) ? pol.sc[(m - 1) / 2] : pol.cc[m / 2]; };
  std::vector<double2> hu(nn);
  (void)hipMemcpy(hu.data(), U, nn * 16, hipMemcpyDeviceToHost);
  std::vector<double> hgrad(K), rgrad(K, 0.0);
  (void)hipMemcpy(hgrad.data(), grad, K * 8, hipMemcpyDeviceToHost);
  double eu = 0, su = 0;
  std::vector<std::vector<double>> x(np + 1, std::vector<double>(d)), xb(np + 2, std::vector<double>(d));
  std::vector<double> acc(R, 0.0), dacc(2, 0.0);
  for (int c = 0; c < d; ++c) {
    std::fill(x[0].begin(), x[0].end(), 0.0); x[0][c] = 1.0;
    for (int m = 1; m <= np; ++m) apply(x[m - 1], x[m]);
    if (c % 97 == 0 || c == d - 1)
      for (int i = 0; i < d; ++i) {
        double re = 0, im = 0;
        for (int m = 0; m <= np; ++m) (m & 1 ? im : re) += coef(m) * x[m][i];
        eu = fmax(eu, fmax(fabs(re - hu[(size_t)c * d + i].x), fabs(im - hu[(size_t)c * d + i].y)));
        su = fmax(su, fmax(fabs(re), fabs(im)));
      }
    for (int i = 0; i < d; ++i) xb[np][i] = coef(np) * hg[(size_t)c * d + i].y;
    for (int m = np - 1; m >= 1; --m) {
      apply(xb[m + 1], xb[m]);
      for (int i = 0; i < d; ++i) xb[m][i] += coef(m) * (m & 1 ? hg[(size_t)c * d + i].y : hg[(size_t)c * d + i].x);
    }
    for (int m = 0; m < np; ++m)
      for (int i = 0; i < d; ++i) {
        const double xv = x[m][i];
        if (xv == 0.0) continue;
        for (int j = 0; j < R; ++j) acc[j] += val(i, j) * xb[m + 1][i ^ tab.xoff[j]] * xv;
        for (int jd = 0; jd < tab.nd; ++jd) dacc[jd] += diag[(size_t)jd * d + i] * xb[m + 1][i] * xv;
      }
  }
  for (int j = 0; j < R; ++j) rgrad[tab.kid[j]] += -dt * acc[j];
  for (int jd = 0; jd < tab.nd; ++jd) if (tab.dkid[jd] >= 0) rgrad[tab.dkid[jd]] += -dt * dacc[jd];
  double eg = 0, sg = 0;
  for (int k = 0; k < K; ++k) { eg = fmax(eg, fabs(rgrad[k] - hgrad[k])); sg = fmax(sg, fabs(rgrad[k])); }
  printf("U rows: max|gpu - host| = %.3e (max |U| %.3e);  gradients: max|gpu - host| = %.3e (max |g| %.3e)\n", eu, su, eg, sg);
  printf((eu <= 1e-12 * su && eg <= 1e-11 * sg) ? "PROBE OK\n" : "PROBE MISMATCH\n");
  return 0;
}
