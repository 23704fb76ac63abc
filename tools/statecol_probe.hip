// Development probe for the state-column kernels (diffquantum_amd/csrc/statecol.hip): BASELINE's generator family (diagonal Ising
// drift + X_k controls) at n qubits, B columns, S steps; times the forward and the backward launch with HIP events and, built
// with -DXS_STAMPS, prints the backward kernel's shader cycles per phase.  Not part of the library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DXS_STAMPS -DXS_PROBE_4096_ONLY -Idiffquantum_amd/csrc tools/statecol_probe.hip -o tools/_build/statecol_probe
//   tools/_build/statecol_probe [n=12] [B=4096] [S=20] [L=7] [variant=2] [forms=3] [economized=1]
// economized = 1: the Chebyshev-economized step polynomial for theta = 0.03 at 2^-53 (degree 6; L is ignored), 0: Taylor of degree L
#include "../diffquantum_amd/csrc/statecol.hip"
#include "../diffquantum_amd/csrc/statecol_bwd.hip"
#include "../diffquantum_amd/csrc/statecol_bwdq.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 12, B = argc > 2 ? atoi(argv[2]) : 4096, S = argc > 3 ? atoi(argv[3]) : 20;
  int L = argc > 4 ? atoi(argv[4]) : 7;
  const int variant = argc > 5 ? atoi(argv[5]) : 2;
  const int forms = argc > 6 ? atoi(argv[6]) : 3;   // bit 0: the exact adjoint form, bit 1: the quadrature form
  const int econ = argc > 7 ? atoi(argv[7]) : 1;
  dq::XsPoly pol;
  {
    const int Le = dq::xs_poly_make(0.03, 1.1102230246251565e-16, econ, &pol);
    if (econ) L = Le;
  }
  const int d = 1 << n, K = n;
  dq::XrTab tab;
  memset(&tab, 0, sizeof(tab));
  tab.R = n;
  tab.nd = 1;
  for (int j = 0; j < n; ++j) { tab.xoff[j] = 1 << (n - 1 - j); tab.kid[j] = j; tab.cmag[j] = 1.0; tab.phase[j] = 0; }
  tab.dkid[0] = -1;
  std::vector<double> dval(d), u((size_t)S * K);
  srand(1);
  for (auto& v : dval) v = 20.0 * (rand() / (double)RAND_MAX - 0.5);
  for (auto& v : u) v = 2.0 * (rand() / (double)RAND_MAX - 0.5);
  std::vector<double2> psi((size_t)B * d), lam((size_t)B * d);
  for (auto& v : psi) { v.x = rand() / (double)RAND_MAX - 0.5; v.y = rand() / (double)RAND_MAX - 0.5; }
  for (auto& v : lam) { v.x = rand() / (double)RAND_MAX - 0.5; v.y = rand() / (double)RAND_MAX - 0.5; }
  double *d_dval, *d_u, *d_wt, *d_dgs, *d_part, *d_grad;
  double2 *d_in, *d_psi, *d_lam, *d_park;
  const int nblk = dq::xs_blocks(d, 8, B, 256, variant);
  CK(hipMalloc(&d_dval, sizeof(double) * d));
  CK(hipMalloc(&d_u, sizeof(double) * u.size()));
  CK(hipMalloc(&d_wt, sizeof(double) * S * n));
  CK(hipMemcpy(d_dval, dval.data(), sizeof(double) * d, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_u, u.data(), sizeof(double) * u.size(), hipMemcpyHostToDevice));
  tab.dval = d_dval;
  tab.sgn = nullptr;
  dq::xr_parity_signs(tab, nullptr, d);
  CK(hipMalloc(&d_dgs, sizeof(double) * dq::xs_dgs_doubles(d, S, tab)));
  CK(hipMalloc(&d_part, sizeof(double) * dq::xs_part_doubles(d, S, nblk, tab, variant)));
  CK(hipMalloc(&d_grad, sizeof(double) * S * K));
  CK(hipMalloc(&d_in, sizeof(double2) * psi.size()));
  CK(hipMalloc(&d_psi, sizeof(double2) * psi.size()));
  CK(hipMalloc(&d_lam, sizeof(double2) * psi.size()));
  CK(hipMalloc(&d_park, sizeof(double2) * dq::xs_park_elems(d, L, nblk)));
  CK(hipMemcpy(d_in, psi.data(), sizeof(double2) * psi.size(), hipMemcpyHostToDevice));
  long step = 0;
  CK(dq::xs_weights_launch(tab, d_u, K, S, d, 1e-3, d_wt, d_dgs, &step, nullptr));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  dq::XsQuad quad;
  if (!dq::xs_quad_make(0.03, 9.1e-13, 2, L, econ, pol, &quad)) { printf("no quadrature rule\n"); return 1; }
  std::vector<double> g_exact((size_t)S * K), g_quad((size_t)S * K);
  for (int form = 0; form < 2; ++form) {
  if (!((forms >> form) & 1)) continue;
  for (int rep = 0; rep < 3; ++rep) {
    float msf = 0, msb = 0;
    CK(hipEventRecord(e0));
    CK(dq::xs_forward_launch(d_in, d_psi, d, tab, d_wt, d_dgs, step, S, B, L, nblk, variant, &pol, nullptr));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&msf, e0, e1));
    CK(hipMemcpy(d_lam, lam.data(), sizeof(double2) * lam.size(), hipMemcpyHostToDevice));
#ifdef XS_STAMPS
    unsigned long long zero[8] = {0};
    CK(hipMemcpyToSymbol(HIP_SYMBOL(dq::xs_stamps), zero, sizeof(zero)));
#endif
    CK(hipEventRecord(e0));
    CK(dq::xs_backward_launch(d_psi, d_lam, d, tab, d_wt, d_dgs, step, S, B, L, nblk, d_park, d_part, K, 1e-3, d_grad, 0, variant,
                              form ? &quad : nullptr, &pol, nullptr));
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&msb, e0, e1));
    printf("%s n=%d B=%d S=%d L=%d variant=%d nblk=%d: forward %.3f ms (%.4f ms/step)  backward %.3f ms (%.4f ms/step)\n", form ? "quadrature" : "exact     ", n, B, S, L, variant,
           nblk, msf, msf / S, msb, msb / S);
#ifdef XS_STAMPS
    unsigned long long st[8];
    CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(dq::xs_stamps), sizeof(st)));
    const double per = 1.0 / ((double)nblk) / ((double)B / nblk) / S;   // cycles per column-step, averaged over workgroups
    printf("   cycles per column-step: load %.0f | reversal %.0f | forward terms %.0f | horner %.0f | reduce %.0f | store %.0f\n",
           st[0] * per * S, st[1] * per, st[2] * per, st[3] * per, st[4] * per, st[5] * per * S);
#endif
  }
  CK(hipMemcpy((form ? g_quad : g_exact).data(), d_grad, sizeof(double) * S * K, hipMemcpyDeviceToHost));
  }
  if (forms == 3) {
    double gd = 0, gm = 0;
    for (size_t i = 0; i < g_exact.size(); ++i) { gd = fmax(gd, fabs(g_exact[i] - g_quad[i])); gm = fmax(gm, fabs(g_exact[i])); }
    printf("max |grad (quadrature) - grad (exact form)| = %.3e  (max |grad| %.3e)\n", gd, gm);
  }
  // reversal check: the backward kernel leaves psi_0 in d_psi
  std::vector<double2> back(psi.size());
  CK(hipMemcpy(back.data(), d_psi, sizeof(double2) * back.size(), hipMemcpyDeviceToHost));
  double err = 0;
  for (size_t i = 0; i < back.size(); ++i) err = fmax(err, fmax(fabs(back[i].x - psi[i].x), fabs(back[i].y - psi[i].y)));
  printf("max |psi_0 (re-derived by the backward sweep) - psi_0| = %.3e\n", err);
  return 0;
}
