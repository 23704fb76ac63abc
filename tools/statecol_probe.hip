// Development probe for the state-column kernels (diffquantum_amd/csrc/statecol.hip): BASELINE's generator family (diagonal Ising
// drift + X_k controls) at n qubits, B columns, S steps; times the forward and the backward launch with HIP events and, built
// with -DXS_STAMPS, prints the backward kernel's shader cycles per phase.  Not part of the library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DXS_STAMPS -DXS_PROBE_4096_ONLY -Idiffquantum_amd/csrc tools/statecol_probe.hip -o tools/_build/statecol_probe
//   tools/_build/statecol_probe [n=12] [B=4096] [S=20] [L=7] [variant=2] [forms=3] [economized=1]
// economized = 1: the Chebyshev-economized step polynomial for theta = 0.03 at 2^-53 (degree 6; L is ignored), 0: Taylor of degree L
#include "../diffquantum_amd/csrc/statecol.hip"
#include "../diffquantum_amd/csrc/statecol_bwd.hip"
#include "../diffquantum_amd/csrc/statecol_bwdq.hip"


// ---- experiment (round 4): the forward chain with TWO independent workgroups per CU --------------------------------------------------
// 256 threads x 16 rows per column, ONE image updated in place (64 KB: two workgroups share a CU's LDS), for the headline's table only
// (n = 12: X_k at offsets 2^k, real, unsigned; offsets 1 .. 128 are LDS partners, 256 .. 2048 the thread's own rows).  One wave per
// SIMD and workgroup: whatever overlap of LDS reads and vector work there is comes from the OTHER workgroup of the CU being in another
// phase.  Timed against xs_forward_kernel (512 x 8, two images, one workgroup per CU) below.
namespace dq {
template <int XH>
__device__ __forceinline__ void xp_local(double w, const double2* cur, double2* o) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    o[r].x = __builtin_fma(w, cur[r ^ XH].y, o[r].x);
    o[r].y = __builtin_fma(-w, cur[r ^ XH].x, o[r].y);
  }
}
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) xp_forward_2wg_kernel(const double2* __restrict__ psi_in, double2* __restrict__ psi_out,
                                                           const double* __restrict__ wt, const double* __restrict__ dgs, int S, int B,
                                                           int L, XsPoly pol) {
  constexpr int NT = 256, RPT = 16, D = NT * RPT;
  extern __shared__ __align__(16) unsigned char xs_lds_raw[];
  double2* img = (double2*)xs_lds_raw;
  const int t = threadIdx.x, lane = t & 63;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    double2 cur[RPT];
    const double2* in = psi_in + (long)b * D;
#pragma unroll
    for (int r = 0; r < RPT; ++r) cur[r] = in[r * NT + t];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RPT; ++r) img[r * NT + t] = cur[r];
    __syncthreads();
    for (int s = 0; s < S; ++s) {
      const double wl = lane < 12 ? wt[(long)s * 12 + lane] : 0.0;   // diagonal j = qubit j: offset 2^(11 - j)
      double2 acc[RPT];
      const double g0 = pol.g[0];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        acc[r].x = g0 * cur[r].x;
        acc[r].y = g0 * cur[r].y;
      }
      for (int m = 1; m <= L; ++m) {
        double2 o[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          o[r].x = 0;
          o[r].y = 0;
        }
#pragma clang loop unroll(disable)
        for (int k = 0; k < 8; ++k) {   // LDS partners: offsets 1 << k, table index 11 - k
          const double w = xs_readlane(wl, 11 - k);
#pragma unroll
          for (int h = 0; h < 2; ++h) {   // two bursts of 8 rows: the partner buffer stays at 32 registers
            double2 v[8];
            xs_fetch<double, NT, 8, false>(1 << k, img + h * 8 * NT, t, v);
#pragma unroll
            for (int r = 0; r < 8; ++r) {
              o[h * 8 + r].x = __builtin_fma(w, v[r].y, o[h * 8 + r].x);
              o[h * 8 + r].y = __builtin_fma(-w, v[r].x, o[h * 8 + r].y);
            }
          }
        }
        xp_local<1>(xs_readlane(wl, 3), cur, o);   // offsets 256, 512, 1024, 2048: the thread's own rows
        xp_local<2>(xs_readlane(wl, 2), cur, o);
        xp_local<4>(xs_readlane(wl, 1), cur, o);
        xp_local<8>(xs_readlane(wl, 0), cur, o);
        const double inv = 1.0 / m, gm = pol.g[m];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const double dgv = dgs[t * RPT + r];
          const double ox = __builtin_fma(dgv, cur[r].y, o[r].x) * inv;
          const double oy = __builtin_fma(-dgv, cur[r].x, o[r].y) * inv;
          cur[r].x = ox;
          cur[r].y = oy;
          acc[r].x = __builtin_fma(gm, ox, acc[r].x);
          acc[r].y = __builtin_fma(gm, oy, acc[r].y);
        }
        if (m < L) {
          __syncthreads();
#pragma unroll
          for (int r = 0; r < RPT; ++r) img[r * NT + t] = cur[r];
          __syncthreads();
        }
      }
#pragma unroll
      for (int r = 0; r < RPT; ++r) cur[r] = acc[r];
      if (s + 1 < S) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPT; ++r) img[r * NT + t] = cur[r];
        __syncthreads();
      }
    }
    double2* out = psi_out + (long)b * D;
#pragma unroll
    for (int r = 0; r < RPT; ++r) out[r * NT + t] = cur[r];
  }
}
template <int XH>
__device__ __forceinline__ void xp_local8(double w, const double2* cur, double2* o) {
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    o[r].x = __builtin_fma(w, cur[r ^ XH].y, o[r].x);
    o[r].y = __builtin_fma(-w, cur[r ^ XH].x, o[r].y);
  }
}
// the same idea with the library's shape: 512 threads x 8 rows, one image in place, capped at 128 registers so that two workgroups
// (4 waves per SIMD) share a CU; partner bursts of 4 rows
__global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) xp_forward_2wg512_kernel(
    const double2* __restrict__ psi_in, double2* __restrict__ psi_out, const double* __restrict__ wt, const double* __restrict__ dgs,
    int S, int B, int L, XsPoly pol) {
  constexpr int NT = 512, RPT = 8, D = NT * RPT;
  extern __shared__ __align__(16) unsigned char xs_lds_raw[];
  double2* img = (double2*)xs_lds_raw;
  const int t = threadIdx.x, lane = t & 63;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    double2 cur[RPT];
    const double2* in = psi_in + (long)b * D;
#pragma unroll
    for (int r = 0; r < RPT; ++r) cur[r] = in[r * NT + t];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RPT; ++r) img[r * NT + t] = cur[r];
    __syncthreads();
    for (int s = 0; s < S; ++s) {
      const double wl = lane < 12 ? wt[(long)s * 12 + lane] : 0.0;
      double2 acc[RPT];
      const double g0 = pol.g[0];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        acc[r].x = g0 * cur[r].x;
        acc[r].y = g0 * cur[r].y;
      }
      for (int m = 1; m <= L; ++m) {
        double2 o[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          o[r].x = 0;
          o[r].y = 0;
        }
#pragma clang loop unroll(disable)
        for (int k = 0; k < 9; ++k) {   // LDS partners: offsets 1 << k, table index 11 - k
          const double w = xs_readlane(wl, 11 - k);
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            double2 v[4];
            xs_fetch<double, NT, 4, false>(1 << k, img + h * 4 * NT, t, v);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              o[h * 4 + r].x = __builtin_fma(w, v[r].y, o[h * 4 + r].x);
              o[h * 4 + r].y = __builtin_fma(-w, v[r].x, o[h * 4 + r].y);
            }
          }
        }
        xp_local8<1>(xs_readlane(wl, 2), cur, o);   // offsets 512, 1024, 2048
        xp_local8<2>(xs_readlane(wl, 1), cur, o);
        xp_local8<4>(xs_readlane(wl, 0), cur, o);
        const double inv = 1.0 / m, gm = pol.g[m];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const double dgv = dgs[t * RPT + r];
          const double ox = __builtin_fma(dgv, cur[r].y, o[r].x) * inv;
          const double oy = __builtin_fma(-dgv, cur[r].x, o[r].y) * inv;
          cur[r].x = ox;
          cur[r].y = oy;
          acc[r].x = __builtin_fma(gm, ox, acc[r].x);
          acc[r].y = __builtin_fma(gm, oy, acc[r].y);
        }
        if (m < L) {
          __syncthreads();
#pragma unroll
          for (int r = 0; r < RPT; ++r) img[r * NT + t] = cur[r];
          __syncthreads();
        }
      }
#pragma unroll
      for (int r = 0; r < RPT; ++r) cur[r] = acc[r];
      if (s + 1 < S) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < RPT; ++r) img[r * NT + t] = cur[r];
        __syncthreads();
      }
    }
    double2* out = psi_out + (long)b * D;
#pragma unroll
    for (int r = 0; r < RPT; ++r) out[r * NT + t] = cur[r];
  }
}
}  // namespace dq

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
  // table: 0 = Ising drift + X_k (BASELINE's generator: real, unsigned), 1 = X_k on half of the qubits and Y_k on the others (I/Q drives:
  // complex, parity-signed), 2 = ring QAOA (drift + ZZ-edge controls on the diagonal, X-vertex controls; demo_maxcut.py:71-79),
  // 3 = X_k on half of the qubits, Z_k on the others
  const int kind = argc > 8 ? atoi(argv[8]) : 0;
  dq::XsPoly pol;
  {
    const int Le = dq::xs_poly_make(0.03, 1.1102230246251565e-16, econ, &pol);
    if (econ) L = Le;
  }
  const int d = 1 << n, K = kind == 2 ? 2 * n : n;
  dq::XrTab tab;
  memset(&tab, 0, sizeof(tab));
  tab.nd = 1;
  tab.dkid[0] = -1;
  std::vector<unsigned> sgn;
  srand(1);
  std::vector<double> dval(d);
  for (auto& v : dval) v = 20.0 * (rand() / (double)RAND_MAX - 0.5);
  if (kind == 0) {
    tab.R = n;
    for (int j = 0; j < n; ++j) { tab.xoff[j] = 1 << (n - 1 - j); tab.kid[j] = j; tab.cmag[j] = 1.0; tab.phase[j] = 0; }
  } else if (kind == 1) {
    tab.R = n;
    sgn.assign(d, 0u);
    for (int j = 0; j < n; ++j) {
      tab.xoff[j] = 1 << (n - 1 - j); tab.kid[j] = j; tab.cmag[j] = 1.0;
      tab.phase[j] = j < n / 2 ? 0 : 1;   // Y[0][1] = -i, Y[1][0] = +i: row i carries -i where its bit is clear
      if (tab.phase[j]) for (int i = 0; i < d; ++i) if (!(i & tab.xoff[j])) sgn[i] |= 1u << j;
    }
  } else if (kind == 2) {
    tab.R = n;
    for (int j = 0; j < n; ++j) { tab.xoff[j] = 1 << (n - 1 - j); tab.kid[j] = n + j; tab.cmag[j] = 1.0; tab.phase[j] = 0; }
    tab.nd = 1 + n;
    dval.assign((size_t)(1 + n) * d, 0.0);
    for (int e = 0; e < n; ++e) {
      tab.dkid[1 + e] = e;
      const int a = 1 << (n - 1 - e), b = 1 << (n - 1 - (e + 1) % n);
      for (int i = 0; i < d; ++i) {
        const double zz = ((i & a) ? -1.0 : 1.0) * ((i & b) ? -1.0 : 1.0);
        dval[(size_t)(1 + e) * d + i] = zz;
        dval[i] += 1.0 - zz;
      }
    }
  } else {
    tab.R = n / 2;
    for (int j = 0; j < n / 2; ++j) { tab.xoff[j] = 1 << (n - 1 - j); tab.kid[j] = j; tab.cmag[j] = 1.0; tab.phase[j] = 0; }
    tab.nd = 1 + (n - n / 2);
    dval.resize((size_t)tab.nd * d);
    for (int j = n / 2; j < n; ++j) {
      tab.dkid[1 + j - n / 2] = j;
      for (int i = 0; i < d; ++i) dval[(size_t)(1 + j - n / 2) * d + i] = (i & (1 << (n - 1 - j))) ? -1.0 : 1.0;
    }
  }
  std::vector<double> u((size_t)S * K);
  for (auto& v : u) v = 2.0 * (rand() / (double)RAND_MAX - 0.5);
  std::vector<double2> psi((size_t)B * d), lam((size_t)B * d);
  for (auto& v : psi) { v.x = rand() / (double)RAND_MAX - 0.5; v.y = rand() / (double)RAND_MAX - 0.5; }
  for (auto& v : lam) { v.x = rand() / (double)RAND_MAX - 0.5; v.y = rand() / (double)RAND_MAX - 0.5; }
  double *d_dval, *d_u, *d_wt, *d_dgs, *d_part, *d_grad;
  double2 *d_in, *d_psi, *d_lam, *d_park;
  const int nblk = dq::xs_blocks(d, 8, B, 256, variant);
  CK(hipMalloc(&d_dval, sizeof(double) * dval.size()));
  unsigned* d_sgn = nullptr;
  if (!sgn.empty()) {
    CK(hipMalloc(&d_sgn, sizeof(unsigned) * d));
    CK(hipMemcpy(d_sgn, sgn.data(), sizeof(unsigned) * d, hipMemcpyHostToDevice));
  }
  CK(hipMalloc(&d_u, sizeof(double) * u.size()));
  CK(hipMalloc(&d_wt, sizeof(double) * S * tab.R));
  CK(hipMemcpy(d_dval, dval.data(), sizeof(double) * dval.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_u, u.data(), sizeof(double) * u.size(), hipMemcpyHostToDevice));
  tab.dval = d_dval;
  tab.sgn = d_sgn;
  dq::xr_parity_signs(tab, sgn.empty() ? nullptr : sgn.data(), d);
  dq::xr_diag_strings(tab, getenv("XS_NO_DSTRING") ? nullptr : dval.data(), d);
  CK(hipMalloc(&d_dgs, sizeof(double) * dq::xs_dgs_doubles(d, S, tab)));
  CK(hipMalloc(&d_part, sizeof(double) * dq::xs_part_doubles(d, S, nblk, tab, variant)));
  CK(hipMalloc(&d_grad, sizeof(double) * S * K));
  CK(hipMalloc(&d_in, sizeof(double2) * psi.size()));
  CK(hipMalloc(&d_psi, sizeof(double2) * psi.size()));
  CK(hipMalloc(&d_lam, sizeof(double2) * psi.size()));
  CK(hipMalloc(&d_park, sizeof(double2) * dq::xs_park_elems(d, L, nblk)));
  CK(hipMemcpy(d_in, psi.data(), sizeof(double2) * psi.size(), hipMemcpyHostToDevice));
  long step = 0;
  CK(dq::xs_weights_launch(tab, d_u, K, S, d, 1e-3, dq::xs_threads(d, variant), d_wt, d_dgs, &step, nullptr));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  dq::XsQuad quad;
  if (!dq::xs_quad_make(0.03, 9.1e-13, 2, L, econ, pol, &quad, dq::xs_quad_basis(tab, d, variant, sizeof(double)) ? 1 : 0)) { printf("no quadrature rule\n"); return 1; }
  printf("table kind %d: R = %d, diagonal sources %d, signed %d, complex %d, quadrature basis %d\n", kind, tab.R, tab.nd, (int)(tab.sgn != nullptr), (int)dq::xs_table_is_complex(tab), quad.basis);
  std::vector<double> g_exact((size_t)S * K), g_quad((size_t)S * K);
  for (int form = 0; form < 2; ++form) {
  if (!((forms >> form) & 1)) continue;
  for (int rep = 0; rep < 3; ++rep) {
    float msf = 0, msb = 0;
    CK(hipEventRecord(e0));
    CK(dq::xs_forward_launch(d_in, d_psi, d, tab, d_wt, d_dgs, step, S, B, L, nblk, variant, &pol, 1.0, 0.0, nullptr));
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
                              form ? &quad : nullptr, &pol, 1.0, 0.0, nullptr));
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
  if (n == 12 && econ && kind == 0 && argc > 9) {   // the two-workgroups-per-CU forward experiment against the library's forward kernel
    double2 *d_alt, *d_alt2;
    CK(hipMalloc(&d_alt, sizeof(double2) * psi.size()));
    CK(hipMalloc(&d_alt2, sizeof(double2) * psi.size()));
    CK(hipFuncSetAttribute((const void*)dq::xp_forward_2wg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    for (int rep = 0; rep < 3; ++rep) {
      float ms = 0, ms0 = 0;
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(dq::xp_forward_2wg_kernel, dim3(512), dim3(256), 65536, nullptr, d_in, d_alt, d_wt, d_dgs, S, B, L, pol);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipEventRecord(e0));
      CK(dq::xs_forward_launch(d_in, d_lam, d, tab, d_wt, d_dgs, step, S, B, L, nblk, variant, &pol, 1.0, 0.0, nullptr));
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms0, e0, e1));
      float ms2 = 0;
      CK(hipFuncSetAttribute((const void*)dq::xp_forward_2wg512_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(dq::xp_forward_2wg512_kernel, dim3(512), dim3(512), 65536, nullptr, d_in, d_alt2, d_wt, d_dgs, S, B, L, pol);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms2, e0, e1));
      printf("forward, two workgroups per CU in place: 256 x 16 %.3f ms, 512 x 8 at 128 registers %.3f ms   library kernel (512 x 8, two images): %.3f ms\n", ms, ms2, ms0);
    }
    std::vector<double2> a(psi.size()), c(psi.size());
    CK(hipMemcpy(a.data(), d_alt, sizeof(double2) * a.size(), hipMemcpyDeviceToHost));
    CK(hipMemcpy(c.data(), d_lam, sizeof(double2) * c.size(), hipMemcpyDeviceToHost));
    double df = 0, df2 = 0;
    for (size_t i = 0; i < a.size(); ++i) df = fmax(df, fmax(fabs(a[i].x - c[i].x), fabs(a[i].y - c[i].y)));
    CK(hipMemcpy(a.data(), d_alt2, sizeof(double2) * a.size(), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < a.size(); ++i) df2 = fmax(df2, fmax(fabs(a[i].x - c[i].x), fabs(a[i].y - c[i].y)));
    printf("max |final states (experiment) - final states (library)| = %.3e (256 x 16), %.3e (512 x 8)\n", df, df2);
    CK(hipFree(d_alt));
    CK(hipFree(d_alt2));
  }
  // reversal check: the backward kernel leaves psi_0 in d_psi
  std::vector<double2> back(psi.size());
  CK(hipMemcpy(back.data(), d_psi, sizeof(double2) * back.size(), hipMemcpyDeviceToHost));
  double err = 0;
  for (size_t i = 0; i < back.size(); ++i) err = fmax(err, fmax(fabs(back[i].x - psi[i].x), fabs(back[i].y - psi[i].y)));
  printf("max |psi_0 (re-derived by the backward sweep) - psi_0| = %.3e\n", err);
  return 0;
}
