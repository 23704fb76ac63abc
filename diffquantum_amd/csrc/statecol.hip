// State-column form of the chain for XOR-regular (Pauli-string) generators: whole Trotter chains on a state vector that never
// leaves LDS.
//
// The reference's inner loop is `psi <- expm(-1j * dt * H(t)) psi` per step (sim_plain.py:135-150, diffqc.cc:190-200).  When
// H(t) = H0 + sum_k u_k(t) H_k is a sum of Pauli strings -- the table of xorexp.hip: besides real diagonals, row i holds the
// entries (i, i ^ x_j), j < R, of magnitude c_j, sign sigma_ij and phase 1 or i -- the step needs no matrix at all:
//     psi' = sum_{m <= L} A^m psi / m!,   A = -i dt H(t),   (A v)[i] = -i dg_i v[i] + sum_j a_ij v[i ^ x_j]
// is L sweeps of R partner reads per row on a vector of d complex numbers.  A workgroup owns one state column (d = 4096,
// complex128: 64 KB; two images fit the 160 KB LDS), a thread the rows r NT + t, partners at offsets with no lane bits are the
// thread's own registers, the others conflict-free LDS reads at i ^ x_j, the weights dt u_k(t) c_j wave-uniform scalars -- and
// the workgroup runs ALL the steps of the chain on its column before it touches HBM again (psi_0 in, psi_S out).
//
// Backward (adjoint) sweep, per column and step t, from psi_{t+1} and lam = lambda_{t+1} (both in place of the forward data):
//     psi_t   = sum_m (-A)^m psi_{t+1} / m!                       (U^H = exp(-A): exact reversal up to the Taylor remainder)
//     x_l     = A^l psi_t / l!,  l < L                             (parked per workgroup in an L2 / Infinity-Cache resident scratch,
//                                                                  each element written and read back by the same thread)
//     y_{L-1} = lam / L,   y_{l-1} = (lam - A y_l) / l,   lambda_t = lam - A y_0                  (Horner; = U^H lam)
//     dL/du_k[t] = Re sum_l <y_l, dA_k x_l>,  dA_k = -i dt H_k
// -- the exact derivative of the same polynomial; the partner value y_l[i ^ x_j] fetched for the sweep is the factor of the
// level-l product with the thread's own x_l[i] (one real accumulator per XOR-diagonal, one per row for diagonal controls).
// 3L - 1 sweeps per step; HBM traffic of a whole gradient evaluation: psi_S and lambda_S in, a few sums per step out.
// The numpy twin of these recurrences is tests/np_state_column.py (checked against the oracle on the CPU).
#include "statecol_dev.h"

namespace dq {

// wt[s][j] = dt u[s][kid_j] cmag_j;  dgs: the rows' diagonal weights dt (H0 + sum_k u_k H_k)[i][i] -- one vector [d] when only
// the drift has diagonal entries (per_step = 0), else one per step [S][d].  grid: (ceil(d / 256), per_step ? S : 1) + S blocks for wt
// Layout of a vector (round 5): THREAD-major for the kernels' shape -- row i = r nt + t of the column is entry t (d / nt) + r, so that a
// thread's rows are contiguous (64 B at 8 rows: 16-byte loads off ONE address; the row-major table cost every kernel 8 address pairs
// that its loop optimizer kept as induction variables and, in the adjoint kernels, spilled once per step).
__global__ void __launch_bounds__(256) xs_weights_kernel(XrTab tab, const double* __restrict__ u, int K, int S, int d, double dt,
                                                         int per_step, int nt, double* __restrict__ wt, double* __restrict__ dgs) {
  const int nbd = (d + 255) / 256, nds = per_step ? S : 1;
  const int b = blockIdx.x, t = threadIdx.x;
  if (b < nbd * nds) {
    const int s = b / nbd, i = (b - s * nbd) * 256 + t;
    if (i < d) {
      const double* ur = u + (long)s * K;
      double v = 0.0;
      for (int jd = 0; jd < tab.nd; ++jd) {
        const int k = tab.dkid[jd];
        v += (k < 0 ? 1.0 : ur[k]) * tab.dval[(long)jd * d + i];
      }
      dgs[(long)s * d + (long)(i % nt) * (d / nt) + i / nt] = dt * v;
    }
  } else {
    const int s = b - nbd * nds;
    const double* ur = u + (long)s * K;
    if (t < tab.R) wt[(long)s * tab.R + t] = dt * (tab.kid[t] < 0 ? 1.0 : ur[tab.kid[t]]) * tab.cmag[t];
  }
}

// row b of the diagonal-weight table: c = (max + min) / 2 of its d entries, every entry -= c; cen[b] = c, half[b] = (max - min) / 2
__global__ void __launch_bounds__(256) xs_center_kernel(double* __restrict__ dgs, int d, double* __restrict__ cen, double* __restrict__ half) {
  double* row = dgs + (long)blockIdx.x * d;
  double mn = 1e300, mx = -1e300;
  for (int i = threadIdx.x; i < d; i += 256) {
    const double v = row[i];
    mn = fmin(mn, v);
    mx = fmax(mx, v);
  }
  for (int off = 32; off > 0; off >>= 1) {
    mn = fmin(mn, __shfl_down(mn, off, 64));
    mx = fmax(mx, __shfl_down(mx, off, 64));
  }
  __shared__ double smn[4], smx[4];
  if ((threadIdx.x & 63) == 0) {
    smn[threadIdx.x >> 6] = mn;
    smx[threadIdx.x >> 6] = mx;
  }
  __syncthreads();
  mn = fmin(fmin(smn[0], smn[1]), fmin(smn[2], smn[3]));
  mx = fmax(fmax(smx[0], smx[1]), fmax(smx[2], smx[3]));
  const double c = 0.5 * (mx + mn);
  for (int i = threadIdx.x; i < d; i += 256) row[i] -= c;
  if (threadIdx.x == 0) {
    cen[blockIdx.x] = c;
    half[blockIdx.x] = 0.5 * (mx - mn);
  }
}

// psi_out[b][:] = U_{S-1} ... U_0 psi_in[b][:] for the columns b = blockIdx.x, + gridDim.x, ...  (states stored column by
// column: [B][D]).  dgs + s * dgs_step: the rows' diagonal weights of step s.
template <class T, int NT, int RPT, int SIGN, bool PREF, bool CPLX>
__global__ void __launch_bounds__(NT) xs_forward_kernel(const typename Cx<T>::type* __restrict__ psi_in,
                                                        typename Cx<T>::type* __restrict__ psi_out, XrTab tab, XsOrder ord,
                                                        const double* __restrict__ wt, const double* __restrict__ dgs, long dgs_step,
                                                        int S, int B, int L, XsPoly pol, XsSeg seg) {
  typedef typename Cx<T>::type T2;
  constexpr int D = NT * RPT;
  extern __shared__ __align__(16) unsigned char xs_lds_raw[];
  T2* img = (T2*)xs_lds_raw;  // two images of D
  xs_lds_aligned<NT>(img);
  xs_stagger_priority(threadIdx.x, NT);
  const int t = threadIdx.x;
  const unsigned sg = SIGN ? XsCtl::thread_signs(tab, t) : 0u;   // bit j: this thread's factor of diagonal j's sign (parity-signed table)
  XsCtl ctl;
  ctl.load_table(tab, ord, t & 63);
  if (SIGN) ctl.load_row_signs(tab, t & 63, __builtin_ctz(NT));
  int p = 0;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    T2 cur[RPT];
    // segments (seg.cols > 0): column b belongs to time segment b / seg.cols and runs the steps [s0, s0 + ns) of the table only;
    // without an input the columns are those of the identity (column b % D): the launch then leaves the segments' propagators
    const int s0 = seg.cols > 0 ? (b / seg.cols) * S : 0;
    const int ns = seg.cols > 0 ? (seg.total - s0 < S ? seg.total - s0 : S) : S;
    if (psi_in) {
      const T2* in = psi_in + (long)b * D;
#pragma unroll
      for (int r = 0; r < RPT; ++r) cur[r] = in[r * NT + t];
    } else {
      const int one = b % D;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        cur[r].x = (r * NT + t == one) ? (T)1 : (T)0;
        cur[r].y = (T)0;
      }
    }
    // the step's weights (and, where the shape has the registers: 512 threads and fewer, its diagonal) are requested a step ahead:
    // at the head of the step nothing hides their latency
    constexpr bool PFD = XS_W_PREFETCH && NT <= 512;
    double wnext = ns > 0 ? ctl.fetch_step(wt + (long)s0 * tab.R, t & 63) : 0.0;
    T dgn[PFD ? RPT : 1];
    if (PFD && ns > 0) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) dgn[r] = (T)dgs[(long)s0 * dgs_step + t * RPT + r];
    }
#if XS_FWD_SPLIT
    // Split sweeps: o += (register classes) runs BEHIND the writes that publish the sweep's input and in front of the barrier -- the
    // image drains into LDS (8 ds_write_b128 per thread at a third of the read rate, nothing else of the workgroup running) while the
    // vector pipes work on the thread's own rows; the LDS classes follow the barrier.  o lives across it (the kernel has the registers).
    T2 o[RPT];
    auto publish_and_start = [&]() {   // cur is the next sweep's input, ctl.w that sweep's weights
      p ^= 1;
      xs_put<T, NT, RPT>(img + p * D, t, cur);
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        o[r].x = 0;
        o[r].y = 0;
      }
      xs_gather_local<T, RPT, SIGN, CPLX>(ctl, sg, cur, (T)1, o);
      __syncthreads();
    };
    ctl.w = wnext;
    if (ns > 0) publish_and_start();
    for (int s = 0; s < ns; ++s) {
      if (s + 1 < ns) wnext = ctl.fetch_step(wt + (long)(s0 + s + 1) * tab.R, t & 63);
      const double* dgr = dgs + (long)(s0 + s) * dgs_step;
      T2 acc[RPT];
      T dg[RPT];  // the rows' diagonal weights of this step (registers: a load per sweep would sit in the vmcnt queue of every wait)
      const T g0 = (T)pol.g[0];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        acc[r].x = g0 * cur[r].x;
        acc[r].y = g0 * cur[r].y;
        dg[r] = PFD ? dgn[r] : (T)dgr[t * RPT + r];
      }
      if (PFD && s + 1 < ns) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) dgn[r] = (T)dgr[dgs_step + t * RPT + r];
      }
      for (int m = 1; m <= L; ++m) {
        xs_gather_lds<T, NT, RPT, SIGN, 1, PREF, CPLX>(ctl, sg, img + p * D, t, (T)1, o);
        // (no division: the recurrence runs on A^m psi and the factorials sit in the table, pol.g[m] = g_m / m!: xs_forward_go)
        const T gm = (T)pol.g[m];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {  // + the diagonal: -i dg v
          const T ox = xs_fma(dg[r], cur[r].y, o[r].x);
          const T oy = xs_fma(-dg[r], cur[r].x, o[r].y);
          cur[r].x = ox;
          cur[r].y = oy;
          acc[r].x = xs_fma(gm, ox, acc[r].x);
          acc[r].y = xs_fma(gm, oy, acc[r].y);
        }
        if (m < L) publish_and_start();
      }
#pragma unroll
      for (int r = 0; r < RPT; ++r) cur[r] = acc[r];
      if (s + 1 < ns) {
        ctl.w = wnext;   // (the next step's weights: its first sweep's register classes run in publish_and_start)
        publish_and_start();
      }
    }
#else
    p ^= 1;
    xs_put<T, NT, RPT>(img + p * D, t, cur);
    __syncthreads();
    for (int s = 0; s < ns; ++s) {
#if XS_W_PREFETCH
      ctl.w = wnext;
      if (s + 1 < ns) wnext = ctl.fetch_step(wt + (long)(s0 + s + 1) * tab.R, t & 63);
#else
      ctl.load_step(wt + (long)(s0 + s) * tab.R, t & 63);
#endif
      const double* dgr = dgs + (long)(s0 + s) * dgs_step;
      T2 acc[RPT];
      T dg[RPT];  // the rows' diagonal weights of this step (registers: a load per sweep would sit in the vmcnt queue of every wait)
      const T g0 = (T)pol.g[0];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        acc[r].x = g0 * cur[r].x;
        acc[r].y = g0 * cur[r].y;
        dg[r] = PFD ? dgn[r] : (T)dgr[t * RPT + r];
      }
      if (PFD && s + 1 < ns) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) dgn[r] = (T)dgr[dgs_step + t * RPT + r];
      }
      for (int m = 1; m <= L; ++m) {
        T2 o[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          o[r].x = 0;
          o[r].y = 0;
        }
        xs_gather<T, NT, RPT, SIGN, false, 1, PREF, true, CPLX>(ctl, sg, img + p * D, cur, t, (T)1, o, nullptr, nullptr);
        // (no division: the recurrence runs on A^m psi and the factorials sit in the table, pol.g[m] = g_m / m!: xs_forward_go)
        const T gm = (T)pol.g[m];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {  // + the diagonal: -i dg v
          const T ox = xs_fma(dg[r], cur[r].y, o[r].x);
          const T oy = xs_fma(-dg[r], cur[r].x, o[r].y);
          cur[r].x = ox;
          cur[r].y = oy;
          acc[r].x = xs_fma(gm, ox, acc[r].x);
          acc[r].y = xs_fma(gm, oy, acc[r].y);
        }
        if (m < L) {
          p ^= 1;
          xs_put<T, NT, RPT>(img + p * D, t, cur);
          __syncthreads();
        }
      }
#pragma unroll
      for (int r = 0; r < RPT; ++r) cur[r] = acc[r];
      if (s + 1 < ns) {
        p ^= 1;
        xs_put<T, NT, RPT>(img + p * D, t, cur);
        __syncthreads();
      }
    }
#endif
    T2* out = psi_out + (long)b * D;
    {   // the scalar of the centred frame (xs_center_launch): e^{-i sum of the steps' midpoints}, per time segment or per launch
      const int g = seg.cols > 0 ? b / seg.cols : 0;
      const T pr = (T)(seg.cols > 0 ? (seg.seg_phase ? seg.seg_phase[2 * g] : seg.ph[2 * (g < XS_SEGMAX ? g : 0)]) : seg.pr);
      const T pi = (T)(seg.cols > 0 ? (seg.seg_phase ? seg.seg_phase[2 * g + 1] : seg.ph[2 * (g < XS_SEGMAX ? g : 0) + 1]) : seg.pi);
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        T2 v;
        v.x = cur[r].x * pr - cur[r].y * pi;
        v.y = cur[r].x * pi + cur[r].y * pr;
        out[r * NT + t] = v;
      }
    }
    __syncthreads();  // (every read of this column's images is done before the next column overwrites one)
  }
}

// grad[s][k] (+)= dt * (sum_j [kid_j == k] cmag_j sum part[.., j] + sum_jd [dkid_jd == k] (dmag_jd) sum part[.., RM + jd]); one workgroup
// per step: lane group q sums slot q over the nblk * nw partial rows in a fixed order
__global__ void __launch_bounds__(1024) xs_finish_kernel(const double* __restrict__ part, int nblk, int nw, int ns, int rm, int S,
                                                         XrTab tab, XsOrder ord, int K, double dt, double* __restrict__ grad,
                                                         int accumulate) {
  const long s = blockIdx.x;
  const int l = threadIdx.x & 31, slot0 = threadIdx.x >> 5;
  __shared__ double s_j[XR_SLOTS];
  for (int slot = slot0; slot < ns; slot += 32) {
    double v = 0.0;
    for (int q = l; q < nblk * nw; q += 32) {
      const int bk = q / nw, w = q - bk * nw;
      v += part[(((long)bk * S + s) * nw + w) * ns + slot];
    }
    for (int off = 16; off > 0; off >>= 1) v += __shfl_down(v, off, 32);
    if (l == 0) s_j[slot] = v;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 1024) {
    double g = accumulate ? grad[s * K + k] : 0.0;
    for (int q = 0; q < tab.R; ++q) {  // slot q = the diagonal at sorted position q
      const int j = ord.perm[q];
      if (tab.kid[j] == k) g += dt * tab.cmag[j] * s_j[q];
    }
    if (ns > rm)
      for (int jd = 0; jd < tab.nd; ++jd)
        if (tab.dkid[jd] == k) g += dt * (tab.dstring ? tab.dmag[jd] : 1.0) * s_j[rm + jd];   // (Z-string sources: the kernels leave the magnitude out)
    grad[s * K + k] = g;
  }
}

// ---- diagonal observables: Lambda_S = 2 M Psi_S and the energies without a product ---------------------------------------------
// out[0] = max |M[i][j]|, i != j (0: M is diagonal); mdiag[i] = Re M[i][i].  grid: d blocks of 256
template <class T2>
__global__ void __launch_bounds__(256) xs_offdiag_kernel(const T2* __restrict__ M, int d, double* __restrict__ out, double* __restrict__ mdiag) {
  const int i = blockIdx.x;
  const T2* row = M + (long)i * d;
  double mx = 0.0;
  for (int j = threadIdx.x; j < d; j += 256) {
    if (j == i) continue;
    const double a = fabs((double)row[j].x), b = fabs((double)row[j].y);
    mx = fmax(mx, fmax(a, b));
  }
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_down(mx, off, 64));
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    mx = fmax(fmax(sm[0], sm[1]), fmax(sm[2], sm[3]));
    if (mx > 0.0) atomicMax((unsigned long long*)out, (unsigned long long)__double_as_longlong(mx));  // (non-negative doubles order as integers)
    mdiag[i] = (double)row[i].x;
  }
}
// lam[b][i] = 2 m_i psi[b][i];  e[b] = sum_i m_i |psi[b][i]|^2   (states column by column: one workgroup per column)
template <class T>
__global__ void __launch_bounds__(256) xs_diag_costate_kernel(const typename Cx<T>::type* __restrict__ psi, const double* __restrict__ mdiag,
                                                              typename Cx<T>::type* __restrict__ lam, double* __restrict__ e, int d) {
  typedef typename Cx<T>::type T2;
  const long b = blockIdx.x;
  const T2* pc = psi + b * d;
  T2* lc = lam + b * d;
  double acc = 0.0;
  for (int i = threadIdx.x; i < d; i += 256) {
    const T2 v = pc[i];
    const double m = mdiag[i];
    T2 o;
    o.x = (T)(2.0 * m * (double)v.x);
    o.y = (T)(2.0 * m * (double)v.y);
    lc[i] = o;
    acc = __builtin_fma(m, (double)v.x * (double)v.x + (double)v.y * (double)v.y, acc);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  __shared__ double sm[4];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) e[b] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}

hipError_t xs_offdiag_launch(const double2* M, int d, double* out, double* mdiag, hipStream_t st) {
  hipLaunchKernelGGL(xs_offdiag_kernel<double2>, dim3((unsigned)d), dim3(256), 0, st, M, d, out, mdiag);
  return hipGetLastError();
}
hipError_t xs_offdiag_launch(const float2* M, int d, double* out, double* mdiag, hipStream_t st) {
  hipLaunchKernelGGL(xs_offdiag_kernel<float2>, dim3((unsigned)d), dim3(256), 0, st, M, d, out, mdiag);
  return hipGetLastError();
}
hipError_t xs_diag_costate_launch(const double2* psi, const double* mdiag, double2* lam, double* e, int d, int B, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(xs_diag_costate_kernel<double>, dim3((unsigned)B), dim3(256), 0, st, psi, mdiag, lam, e, d);
  return hipGetLastError();
}
hipError_t xs_diag_costate_launch(const float2* psi, const double* mdiag, float2* lam, double* e, int d, int B, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(xs_diag_costate_kernel<float>, dim3((unsigned)B), dim3(256), 0, st, psi, mdiag, lam, e, d);
  return hipGetLastError();
}

// ---- host side ---------------------------------------------------------------------------------------------------------

void xr_parity_signs(XrTab& tab, const unsigned* sgn_h, int d) {
  memset(tab.zmask, 0, sizeof(tab.zmask));
  tab.s0mask = 0;
  tab.parity = 1;
  if (!sgn_h) return;
  int n = 0;
  while ((1 << n) < d) ++n;
  for (int j = 0; j < tab.R; ++j) {
    const unsigned s0 = (sgn_h[0] >> j) & 1u;
    unsigned z = 0;
    for (int b = 0; b < n; ++b)
      if (((sgn_h[(size_t)1 << b] >> j) & 1u) != s0) z |= 1u << b;
    for (int i = 0; i < d; ++i)
      if (((sgn_h[i] >> j) & 1u) != (s0 ^ ((unsigned)__builtin_popcount((unsigned)i & z) & 1u))) {
        tab.parity = 0;   // (a sum of several strings of equal weight on one offset: the column kernels' sign words still take it)
        return;
      }
    tab.zmask[j] = z;
    tab.s0mask |= s0 << j;
  }
}

void xr_diag_strings(XrTab& tab, const double* dval_h, int d) {
  memset(tab.dzmask, 0, sizeof(tab.dzmask));
  memset(tab.dmag, 0, sizeof(tab.dmag));
  tab.dstring = 0;
  if (!dval_h) return;
  int n = 0;
  while ((1 << n) < d) ++n;
  bool any = false;
  for (int jd = 0; jd < tab.nd; ++jd) {
    if (tab.dkid[jd] < 0) continue;
    const double* v = dval_h + (size_t)jd * d;
    const double c = v[0];
    if (c == 0.0) return;
    unsigned z = 0;
    for (int b = 0; b < n; ++b)
      if (v[(size_t)1 << b] != c) z |= 1u << b;
    for (int i = 0; i < d; ++i)
      if (v[i] != ((__builtin_popcount((unsigned)i & z) & 1) ? -c : c)) return;   // (not a single Z-string: the table values stay in use)
    tab.dzmask[jd] = z;
    tab.dmag[jd] = c;
    any = true;
  }
  tab.dstring = any ? 1 : 0;
}

bool xs_supported(int d, int L, int R, int nd, size_t elem) {
  if (elem != sizeof(double) && elem != sizeof(float)) return false;
  if (R < 1 || R > XR_RMAX || nd < 0 || nd > XR_DMAX || L < 1 || L > 24) return false;
  // (d = 8192: a column is 64 KB in complex64 -- two images fit the LDS as at d = 4096 in complex128; in complex128 they do not)
  return d == 256 || d == 512 || d == 1024 || d == 2048 || d == 4096 || (d == 8192 && elem == sizeof(float));
}

size_t xs_lds_bytes(int d, size_t elem) { return (size_t)2 * 2 * elem * (size_t)d; }

// threads of a workgroup (rows per thread = d / threads).  variant bit 0 = "wide": more threads per column, fewer rows per thread
// (d = 4096: 1024 x 4 instead of 512 x 8; d = 1024: 256 x 4 instead of 128 x 8; d = 512: 256 x 2 instead of 64 x 8; d = 256:
// 256 x 1 instead of 64 x 4) -- for narrow batches, where one wave per column leaves the CUs at 2-4 waves and every sweep is a
// chain of LDS latencies
int xs_threads(int d, int variant) {
  if (d >= 8192) return 1024;
  if (d >= 4096) return (variant & 1) ? 1024 : 512;
  if (d == 2048) return 256;
  if (variant & 1) return 256;
  return d / 8 < 64 ? 64 : d / 8;
}

// workgroups of a launch over B columns: as many as fit the chip at once (LDS-limited), never more than columns
int xs_blocks(int d, size_t elem, int B, int n_cu, int variant) {
  const size_t lds = xs_lds_bytes(d, elem);
  int per_cu = (int)((size_t)160 * 1024 / lds);
  const int threads = xs_threads(d, variant);
  const int by_waves = 2048 / threads > 0 ? 2048 / threads : 1;  // 32 waves per CU
  if (per_cu > by_waves) per_cu = by_waves;
  if (per_cu < 1) per_cu = 1;
  long nb = (long)per_cu * n_cu;
  if (nb > B) nb = B;
  // (an even share of columns per workgroup: the smallest count that needs no more rounds)
  const long rounds = (B + nb - 1) / nb;
  nb = (B + rounds - 1) / rounds;
  return (int)(nb < 1 ? 1 : nb);
}
int xs_waves(int d, int variant) { return xs_threads(d, variant) / 64; }
// accumulators per thread of the adjoint kernel: one vector of 16 for up to 16 XOR-diagonals, two for up to 32
int xs_rm(const XrTab& tab) { return tab.R > XR_RSMALL ? 2 * XR_RSMALL : XR_RSMALL; }
bool xs_has_diag_controls(const XrTab& tab) {
  for (int j = 0; j < tab.nd; ++j)
    if (tab.dkid[j] >= 0) return true;
  return false;
}
int xs_slots(const XrTab& tab) { return xs_rm(tab) + (xs_has_diag_controls(tab) ? XR_DMAX : 0); }
int xs_lds_diagonals(int d, const XrTab& tab, int variant) {
  const int nt = xs_threads(d, variant);
  int n = 0;
  for (int j = 0; j < tab.R; ++j)
    if (tab.xoff[j] & (nt - 1)) ++n;
  return n;
}
// the order in which the sweeps walk the table (see XsCtl)
// local = false: offsets without lane bits join the second LDS class (the pairing sweeps' order)
XsOrder xs_make_order(int d, const XrTab& tab, int variant, bool local) {
  XsOrder o;
  memset(&o, 0, sizeof(o));
  const int nt = xs_threads(d, variant);
  int lnt = 0;
  while ((1 << lnt) < nt) ++lnt;
  // (row factor of the sign: the diagonal's parity mask reaches the row bits of the index)
  auto rowsigned = [&](int j) { return tab.sgn != nullptr && tab.parity && (tab.zmask[j] >> lnt) != 0; };
  int at = 0;
  // within every class: phase 0 first; within a phase: the diagonals without a row factor first
  for (int ph = 0; ph < 2; ++ph)
    for (int rs = 0; rs < 2; ++rs)
      for (int j = 0; j < tab.R; ++j)
        if ((tab.xoff[j] & (nt - 1)) != 0 && tab.xoff[j] < nt && (tab.phase[j] != 0) == (ph != 0) && rowsigned(j) == (rs != 0)) {
          o.perm[at++] = (unsigned char)j;
          ++o.n0;
          o.n0p1 += ph;
          o.rsn[0 + ph] += rs;
          o.gmask |= 1ull << (0 + ph * 2 + rs);
        }
  for (int ph = 0; ph < 2; ++ph)
    for (int rs = 0; rs < 2; ++rs)
      for (int j = 0; j < tab.R; ++j)
        if (((tab.xoff[j] & (nt - 1)) != 0 ? tab.xoff[j] >= nt : !local) && (tab.phase[j] != 0) == (ph != 0) && rowsigned(j) == (rs != 0)) {
          o.perm[at++] = (unsigned char)j;
          ++o.n1;
          o.n1p1 += ph;
          o.rsn[2 + ph] += rs;
          o.gmask |= 1ull << (4 + ph * 2 + rs);
        }
  for (int xh = 1; xh < 8 && local; ++xh)
    for (int ph = 0; ph < 2; ++ph)
      for (int rs = 0; rs < 2; ++rs)
        for (int j = 0; j < tab.R; ++j)
          if ((tab.xoff[j] & (nt - 1)) == 0 && tab.xoff[j] / nt == xh && (tab.phase[j] != 0) == (ph != 0) && rowsigned(j) == (rs != 0)) {
            o.perm[at++] = (unsigned char)j;
            ++o.nloc[xh];
            o.nlocp1[xh] += ph;
            o.rsn[4 + 2 * (xh - 1) + ph] += rs;
            o.gmask |= 1ull << (8 + (xh - 1) * 4 + ph * 2 + rs);
          }
  return o;
}
// any diagonal of the order with a row factor of its sign (the odd bits of gmask)
bool xs_order_row_signs(const XsOrder& o) { return (o.gmask & 0xAAAAAAAAAull) != 0; }
size_t xs_park_elems(int d, int L, int nblk) { return (size_t)nblk * (size_t)L * (size_t)d; }
size_t xs_part_doubles(int d, int S, int nblk, const XrTab& tab, int variant) {
  return (size_t)nblk * (size_t)S * (size_t)xs_waves(d, variant) * (size_t)xs_slots(tab);
}

// dgs: d doubles, or S * d when a control has diagonal entries (xs_dgs_doubles); returns the step stride through *dgs_step
size_t xs_dgs_doubles(int d, int S, const XrTab& tab) { return (size_t)d * (size_t)(xs_has_diag_controls(tab) ? (S > 0 ? S : 1) : 1); }

__global__ void __launch_bounds__(256) xs_shift_kernel(double* __restrict__ dgs, int d, double c) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < d) dgs[i] -= c;
}
hipError_t xs_shift_launch(double* dgs, int d, double c, hipStream_t st) {
  hipLaunchKernelGGL(xs_shift_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, dgs, d, c);
  return hipGetLastError();
}
hipError_t xs_center_launch(double* dgs, int d, int rows, double* cen, double* half, hipStream_t st) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(xs_center_kernel, dim3((unsigned)rows), dim3(256), 0, st, dgs, d, cen, half);
  return hipGetLastError();
}

hipError_t xs_weights_launch(const XrTab& tab, const double* u, int K, int S, int d, double dt, int nt, double* wt, double* dgs,
                             long* dgs_step, hipStream_t st) {
  const int per_step = xs_has_diag_controls(tab) ? 1 : 0;
  *dgs_step = per_step ? d : 0;
  if (S <= 0) return hipSuccess;
  const int nbd = (d + 255) / 256;
  hipLaunchKernelGGL(xs_weights_kernel, dim3((unsigned)(nbd * (per_step ? S : 1) + S)), dim3(256), 0, st, tab, u, K, S, d, dt, per_step,
                     nt, wt, dgs);
  return hipGetLastError();
}

namespace {
XsPoly xs_poly_taylor() {
  XsPoly p;
  for (int m = 0; m <= XS_LMAX; ++m) p.g[m] = 1.0;
  return p;
}
// J_k(x), 0 <= x <= 2: the power series (alternating, terms falling from the first on)
long double xs_bessel(int k, long double x) {
  long double term = 1.0L;
  for (int j = 1; j <= k; ++j) term *= 0.5L * x / (long double)j;
  long double sum = term;
  for (int j = 1; j < 60; ++j) {
    term *= -0.25L * x * x / ((long double)j * (long double)(k + j));
    sum += term;
    if (fabsl(term) < 1e-24L * fabsl(sum)) break;
  }
  return sum;
}
// level factors g[m] = m! x (coefficient of z^m, z = i theta x) of the Chebyshev sum of exp(i theta x) cut at degree L, in extended
// precision: g[m] = 1 - O(theta^2) and the sum below is led by its first term, 2 J_m(theta) 2^(m-1) / theta^m = 1 / m! (1 + O(theta^2))
void xs_poly_factors(double theta, int L, double* g) {
  if (!(theta > 0.0)) {
    for (int m = 0; m <= L; ++m) g[m] = 1.0;
    return;
  }
  struct ChebTable {   // T_k's coefficients (integers below 2^23: exact); built once (thread-safe static initialisation)
    long double c[XS_LMAX + 1][XS_LMAX + 1];
    ChebTable() {
      memset(c, 0, sizeof(c));
      c[0][0] = 1.0L;
      c[1][1] = 1.0L;
      for (int k = 2; k <= XS_LMAX; ++k)
        for (int m = 0; m <= k; ++m) c[k][m] = (m > 0 ? 2.0L * c[k - 1][m - 1] : 0.0L) - c[k - 2][m];
    }
  };
  static const ChebTable cheb;
  const auto& tc = cheb.c;
  long double jk[XS_LMAX + 1];
  for (int k = 0; k <= L; ++k) jk[k] = xs_bessel(k, (long double)theta);
  long double pw = 1.0L, fact = 1.0L;  // theta^m, m!
  for (int m = 0; m <= L; ++m) {
    if (m > 0) {
      pw *= (long double)theta;
      fact *= (long double)m;
    }
    long double acc = 0.0L;
    for (int k = m; k <= L; k += 2)  // i^k T_k(x) = sum_m tc[k][m] i^(k-m) (i x)^m, i^(k-m) = (-1)^((k-m)/2)
      acc += (k == 0 ? 1.0L : 2.0L) * jk[k] * tc[k][m] * (((k - m) / 2) & 1 ? -1.0L : 1.0L);
    g[m] = (double)(acc / pw * fact);
  }
}
}  // namespace

// the polynomial of a GIVEN degree (level factors of the Chebyshev sum cut there, or Taylor's ones)
void xs_poly_of_degree(double theta, int L, int economize, XsPoly* p) {
  *p = xs_poly_taylor();
  if (economize && theta > 0.0 && theta <= 2.0 && L >= 1 && L <= XS_LMAX) xs_poly_factors(theta, L, p->g);
}

int xs_poly_make(double theta, double tol, int economize, XsPoly* p) {
  *p = xs_poly_taylor();
  if (!(theta >= 0.0)) return 0;
  if (economize && theta <= 2.0) {
    long double tail[XS_LMAX + 42];
    for (int k = 0; k < XS_LMAX + 42; ++k) tail[k] = fabsl(xs_bessel(k, (long double)theta));
    for (int k = XS_LMAX + 40; k >= 0; --k) tail[k] += tail[k + 1];  // tail[k] = sum_{j >= k} |J_j|
    for (int L = 1; L <= XS_LMAX; ++L)
      if ((double)(2.0L * tail[L + 1]) <= tol) {
        xs_poly_factors(theta, L, p->g);
        return L;
      }
    return 0;
  }
  for (int L = 1; L <= XS_LMAX; ++L) {
    double v = 1.0;
    for (int j = 1; j <= L + 1; ++j) v *= theta / j;
    if (v <= tol) return L;
  }
  return 0;
}

// Gauss-Lobatto rules on [0, 1] with 2 and 3 interior nodes (4 / 5 points: exact to degree 5 / 7).  The rule's error for an
// integrand with |f^(p)| <= F, p = 2 ni + 2, is e F with e = |int s^p - Q[s^p]| / p! (its Peano kernel keeps one sign).
bool xs_quad_make(double theta, double tol, int ni_max, int L, int economize, const XsPoly& poly, XsQuad* qd, int basis) {
  memset(qd, 0, sizeof(*qd));
  if (!(theta >= 0.0) || L < 1 || L > XS_LMAX) return false;
  for (int ni = 2; ni <= ni_max && ni <= XS_NIMAX; ++ni) {
    double s[XS_NIMAX], w[XS_NIMAX], we;
    if (ni == 2) {
      const double a = std::sqrt(0.2);
      s[0] = 0.5 * (1.0 - a);
      s[1] = 0.5 * (1.0 + a);
      w[0] = w[1] = 5.0 / 12.0;
      we = 1.0 / 12.0;
    } else {
      const double a = std::sqrt(3.0 / 7.0);
      s[0] = 0.5 * (1.0 - a);
      s[1] = 0.5;
      s[2] = 0.5 * (1.0 + a);
      w[0] = w[2] = 49.0 / 180.0;
      w[1] = 16.0 / 45.0;
      we = 1.0 / 20.0;
    }
    const int p = 2 * ni + 2;
    double qp = we, fact = 1.0;  // Q[s^p]: the node 0 gives nothing, the node 1 its weight
    for (int i = 0; i < ni; ++i) qp += w[i] * std::pow(s[i], p);
    for (int j = 2; j <= p; ++j) fact *= j;
    const double e = std::fabs(1.0 / (p + 1) - qp) / fact;
    if (e * std::pow(2.0 * theta, p) > tol) continue;
    qd->ni = ni;
    qd->w0 = -we;
    for (int m = 0; m <= XS_LMAX; ++m) {
      qd->g[m] = poly.g[m];
      qd->w1[m] = -we * poly.g[m];
    }
    for (int i = 0; i < ni; ++i) {
      double gi[XS_LMAX + 1];
      for (int m = 0; m <= XS_LMAX; ++m) gi[m] = 1.0;
      if (economize && theta <= 2.0) xs_poly_factors(theta * s[i], L, gi);  // the same degree on the shorter interval s_i theta
      for (int m = 0; m <= XS_LMAX; ++m) {
        qd->c1[m][i] = gi[m] * std::pow(s[i], m);
        qd->c2[m][i] = -w[i] * qd->c1[m][i];
      }
    }
    if (basis != 0) {
      if (ni != 2) return false;
      // the difference basis (dq_internal.h): node 1 (s = 0.72) is the near one to the end point 1, node 0 (s = 0.28) the far one
      const double a0 = qd->c1[0][0] / qd->g[0], a1 = qd->c1[0][1] / qd->g[0];
      double e0[XS_LMAX + 1], e1[XS_LMAX + 1];
      e0[0] = e1[0] = 0.0;
      for (int m = 1; m <= XS_LMAX; ++m) {
        e0[m] = qd->c1[m][0] - a0 * qd->g[m];
        e1[m] = qd->c1[m][1] - a1 * qd->g[m];
      }
      const double beta = e0[1] / e1[1];
      for (int m = 0; m <= XS_LMAX; ++m) {
        const double c20 = qd->c2[m][0], c21 = qd->c2[m][1];   // the weights of q_m on the far / near node value
        qd->w1[m] += c20 * a0 + c21 * a1;
        qd->c2[m][0] = c20 * beta + c21;
        qd->c2[m][1] = c20;
        qd->c1[m][0] = e1[m];
        qd->c1[m][1] = m <= 1 ? 0.0 : e0[m] - beta * e1[m];
      }
      qd->basis = 1;
    }
    return true;
  }
  return false;
}
// The quadrature kernel (statecol_bwdq.hip): tables of up to 16 XOR-diagonals, the 4-point rule, shapes of up to 512 threads per
// column.  Unsigned real tables without diagonal controls (BASELINE's Ising + X_k) keep the node basis, both node vectors in
// registers; tables with parity signs / complex phases (I/Q drives) or diagonal controls (ZZ-edge + X-vertex QAOA) take the difference
// basis with u_2 in LDS (round 5): in complex128 their sign state and row accumulators did not fit beside two node vectors (129 - 175
// spilled registers, slower than the exact form in round 4).  The shapes of 1024 threads per column (d = 4096 "wide", d = 8192 in
// complex64: 128 registers per wave) keep the exact form.
#ifndef XS_FORCE_BASIS
#define XS_FORCE_BASIS 0   // probe builds: 1 = every table takes the difference-basis kernel
#endif
bool xs_quad_basis(const XrTab& tab, int d, int variant, size_t elem) {
  if (elem != sizeof(double) || d / xs_threads(d, variant) != 8) return false;   // (built for complex128 at 8 rows per thread)
  return XS_FORCE_BASIS || tab.sgn != nullptr || xs_has_diag_controls(tab);
}
bool xs_quad_supported(const XrTab& tab, int ni, int d, int variant, size_t elem) {
  if (ni != 2 || tab.R > XR_RSMALL || xs_threads(d, variant) >= 1024) return false;
  if (xs_quad_basis(tab, d, variant, elem) && xs_lds_bytes(d, elem) + sizeof(float2) * (size_t)d > (size_t)160 * 1024) return false;
  return true;
}

namespace {

template <class T, int NT, int RPT>
hipError_t xs_forward_go(const typename Cx<T>::type* in, typename Cx<T>::type* out, const XrTab& tab, const double* wt, const double* dgs,
                         long dgs_step, int S, int B, int L, int nblk, bool pref, const XsPoly& pol, const XsSeg& seg, hipStream_t st) {
  typedef typename Cx<T>::type T2;
  const size_t lds = xs_lds_bytes(NT * RPT, sizeof(T));
  const bool sgn = tab.sgn != nullptr;
  bool cplx = false;
  for (int j = 0; j < tab.R; ++j) cplx = cplx || tab.phase[j] != 0;
  if (cplx && !sgn) return hipErrorInvalidValue;  // (complex tables always carry sign words)
  const XsOrder ord = xs_make_order(NT * RPT, tab, xs_threads(NT * RPT, 0) == NT ? 0 : 1, true);
  const bool rs = xs_order_row_signs(ord);   // (SIGN = 2: the loops of row-signed diagonals exist; 1: parity signs within the thread bits)
  // the kernel's recurrence is v_m = A v_{m-1} without the division by m (16 multiplications and an f64 division per sweep and
  // thread): the factorials go into the level factors, g_m / m!
  XsPoly pk = pol;
  {
    double f = 1.0;
    for (int m = 1; m <= XS_LMAX; ++m) {
      f /= (double)m;
      pk.g[m] = pol.g[m] * f;
    }
  }
  void (*kf)(const T2*, T2*, XrTab, XsOrder, const double*, const double*, long, int, int, int, XsPoly, XsSeg) =
      cplx ? (rs ? (pref ? xs_forward_kernel<T, NT, RPT, 2, true, true> : xs_forward_kernel<T, NT, RPT, 2, false, true>)
                 : (pref ? xs_forward_kernel<T, NT, RPT, 1, true, true> : xs_forward_kernel<T, NT, RPT, 1, false, true>))
           : (sgn ? (rs ? (pref ? xs_forward_kernel<T, NT, RPT, 2, true, false> : xs_forward_kernel<T, NT, RPT, 2, false, false>)
                        : (pref ? xs_forward_kernel<T, NT, RPT, 1, true, false> : xs_forward_kernel<T, NT, RPT, 1, false, false>))
                  : (pref ? xs_forward_kernel<T, NT, RPT, 0, true, false> : xs_forward_kernel<T, NT, RPT, 0, false, false>));
  hipError_t e = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kf, dim3((unsigned)nblk), dim3(NT), lds, st, in, out, tab, ord, wt, dgs,
                     dgs_step, S, B, L, pk, seg);
  return hipGetLastError();
}

// variant: bit 0 = 1024 threads x 4 rows at d = 4096; bit 1 = prefetch of the next diagonal's partners in the sweeps without pairings
template <class T>
hipError_t xs_forward_t(const typename Cx<T>::type* in, typename Cx<T>::type* out, int d, const XrTab& tab, const double* wt,
                        const double* dgs, long dgs_step, int S, int B, int L, int nblk, int variant, const XsPoly* poly, const XsSeg& seg,
                        hipStream_t st) {
  if (B <= 0) return hipSuccess;
  if (!in && seg.cols <= 0) return hipErrorInvalidValue;
  const XsPoly pol = poly ? *poly : xs_poly_taylor();
  if (!xs_supported(d, L, tab.R, tab.nd, sizeof(T))) return hipErrorInvalidValue;
  const bool pf = (variant & 2) != 0;
  if constexpr (sizeof(T) == sizeof(float)) {
    if (d == 8192) return xs_forward_go<T, 1024, 8>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, pol, seg, st);
  }
  switch (d) {
    case 4096:
      return (variant & 1) ? xs_forward_go<T, 1024, 4>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, pol, seg, st)
                           : xs_forward_go<T, 512, 8>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, pol, seg, st);
#ifndef XS_PROBE_4096_ONLY
    case 2048: return xs_forward_go<T, 256, 8>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, pol, seg, st);
    case 1024:
      return (variant & 1) ? xs_forward_go<T, 256, 4>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, pol, seg, st)
                           : xs_forward_go<T, 128, 8>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, pol, seg, st);
    case 512:
      return (variant & 1) ? xs_forward_go<T, 256, 2>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, pol, seg, st)
                           : xs_forward_go<T, 64, 8>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, pol, seg, st);
    default:
      return (variant & 1) ? xs_forward_go<T, 256, 1>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, pol, seg, st)
                           : xs_forward_go<T, 64, 4>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, pol, seg, st);
#else
    default: return hipErrorInvalidValue;
#endif
  }
}

template <class T>
hipError_t xs_backward_t(typename Cx<T>::type* psi, typename Cx<T>::type* lam, int d, const XrTab& tab, const double* wt,
                         const double* dgs, long dgs_step, int S, int B, int L, int nblk, typename Cx<T>::type* park, double* part,
                         int K, double dt, double* grad, int accumulate, int variant, const XsQuad* quad, const XsPoly* poly, double pr, double pi,
                         hipStream_t st) {
  if (B <= 0 || S <= 0) return hipSuccess;
  const XsPoly pol = poly ? *poly : xs_poly_taylor();
  if (!xs_supported(d, L, tab.R, tab.nd, sizeof(T)) || (!park && !quad) || !part) return hipErrorInvalidValue;
  const hipError_t e = quad ? xs_backward_quad_launch<T>(psi, lam, d, tab, wt, dgs, dgs_step, S, B, L, nblk, *quad, part, variant, pr, pi, st)
                            : xs_backward_exact_launch<T>(psi, lam, d, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, variant, pr, pi, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(xs_finish_kernel, dim3((unsigned)S), dim3(1024), 0, st, part, nblk, xs_waves(d, variant), xs_slots(tab),
                     xs_rm(tab), S, tab, xs_make_order(d, tab, variant, quad && XS_PAIR_LOCAL), K, dt, grad, accumulate);
  return hipGetLastError();
}

}  // namespace

namespace {
// the segments' factors: by value up to XS_SEGMAX of them (host table), identity when there is none
XsSeg xs_seg_make(int d, int S_total, int n_seg, const double* seg_phase_host) {
  XsSeg sg{d, S_total, nullptr, 1.0, 0.0, {}};
  for (int g = 0; g < XS_SEGMAX; ++g) {
    sg.ph[2 * g] = seg_phase_host && g < n_seg ? seg_phase_host[2 * g] : 1.0;
    sg.ph[2 * g + 1] = seg_phase_host && g < n_seg ? seg_phase_host[2 * g + 1] : 0.0;
  }
  return sg;
}
}  // namespace

hipError_t xs_forward_launch(const double2* in, double2* out, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                             int S, int B, int L, int nblk, int variant, const XsPoly* poly, double pr, double pi, hipStream_t st) {
  return xs_forward_t<double>(in, out, d, tab, wt, dgs, dgs_step, S, B, L, nblk, variant, poly, XsSeg{0, 0, nullptr, pr, pi, {}}, st);
}
hipError_t xs_forward_launch(const float2* in, float2* out, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                             int S, int B, int L, int nblk, int variant, const XsPoly* poly, double pr, double pi, hipStream_t st) {
  return xs_forward_t<float>(in, out, d, tab, wt, dgs, dgs_step, S, B, L, nblk, variant, poly, XsSeg{0, 0, nullptr, pr, pi, {}}, st);
}
// The propagators of n_seg consecutive time segments of seg_steps steps each (the last one what is left of S_total): column i of
// segment g -- exp(A_{s0+ns-1}) ... exp(A_{s0}) e_i, s0 = g seg_steps -- goes to out[(g d + i) d ..]: the d x d block of segment g
// holds the TRANSPOSE of its propagator.  wt / dgs: the weight tables of all S_total steps.
hipError_t xs_forward_segments_launch(double2* out, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                                      int S_total, int seg_steps, int n_seg, int L, int nblk, int variant, const XsPoly* poly,
                                      const double* seg_phase, hipStream_t st) {
  return xs_forward_t<double>(nullptr, out, d, tab, wt, dgs, dgs_step, seg_steps, n_seg * d, L, nblk, variant, poly,
                              xs_seg_make(d, S_total, n_seg, seg_phase), st);
}
hipError_t xs_forward_segments_launch(float2* out, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                                      int S_total, int seg_steps, int n_seg, int L, int nblk, int variant, const XsPoly* poly,
                                      const double* seg_phase, hipStream_t st) {
  return xs_forward_t<float>(nullptr, out, d, tab, wt, dgs, dgs_step, seg_steps, n_seg * d, L, nblk, variant, poly,
                             xs_seg_make(d, S_total, n_seg, seg_phase), st);
}
hipError_t xs_backward_launch(double2* psi, double2* lam, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                              int S, int B, int L, int nblk, double2* park, double* part, int K, double dt, double* grad,
                              int accumulate, int variant, const XsQuad* quad, const XsPoly* poly, double pr, double pi, hipStream_t st) {
  return xs_backward_t<double>(psi, lam, d, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, K, dt, grad, accumulate, variant, quad, poly, pr, pi, st);
}
hipError_t xs_backward_launch(float2* psi, float2* lam, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                              int S, int B, int L, int nblk, float2* park, double* part, int K, double dt, double* grad, int accumulate,
                              int variant, const XsQuad* quad, const XsPoly* poly, double pr, double pi, hipStream_t st) {
  return xs_backward_t<float>(psi, lam, d, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, K, dt, grad, accumulate, variant, quad, poly, pr, pi, st);
}

}  // namespace dq
