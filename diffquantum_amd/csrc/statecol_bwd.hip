// State-column form, adjoint sweep in its EXACT form: the derivative of the step polynomial itself (3 L - 1 sweeps per step; the
// forward terms of psi_t parked in a scratch).  Runs when the quadrature form (statecol_bwdq.hip) is switched off, for steps too
// large for its rules, for tables of more than 16 XOR-diagonals and for tables with row signs.  See statecol.hip for the scheme.
#include "statecol_dev.h"

namespace dq {

// Backward sweep of the chain for the columns b = blockIdx.x, + gridDim.x, ...:  psi [b][:] holds psi_S on entry and psi_0 on exit,
// lam [b][:] holds lambda_S on entry and lambda_0 on exit.
// LDS: ONE work image W (the vector the sweeps gather from, updated in place between two barriers) and the lambda image Lm --
// lambda stays in LDS for the whole chain, each row touched by its own thread only (no barrier concerns it).  park: per
// workgroup the L forward terms x_0 .. x_{L-1} of the current step (device memory, written and read back by the same thread; 134 MB
// for the whole grid at d = 4096 -- it streams through the Infinity Cache / HBM and is the kernel's memory traffic: 2 L d 16 B per
// column and step).
// part[((blockIdx.x * S + s) * NW + wave) * NS + slot]: this workgroup's share, over its columns, of
//   slot < RM: sum_i of the level pairings of the XOR-diagonal at SORTED position `slot` (before the factors cmag and dt);
//   slot = RM + jd (DIAGCTL): sum_i dval_jd[i] rowacc[i] of the diagonal control jd.
template <class T, int NT, int RPT, int SIGN, bool DIAGCTL, int RM, bool PREF, bool CPLX>
__global__ void __launch_bounds__(NT) xs_backward_kernel(typename Cx<T>::type* __restrict__ psi, typename Cx<T>::type* __restrict__ lam,
                                                         XrTab tab, XsOrder ord, XsOrder ordh, const double* __restrict__ wt,
                                                         const double* __restrict__ dgs, long dgs_step, int S, int B, int L,
                                                         typename Cx<T>::type* __restrict__ park,
                                                         double* __restrict__ part, XsPoly pol, T pr, T pi) {
  typedef typename Cx<T>::type T2;
  constexpr int D = NT * RPT;
  constexpr int NW = NT / 64;
  constexpr int NS = RM + (DIAGCTL ? XR_DMAX : 0);
  extern __shared__ __align__(16) unsigned char xs_lds_raw[];
  T2* W = (T2*)xs_lds_raw;
  T2* Lm = W + D;
  xs_lds_aligned<NT>(W);
  xs_stagger_priority(threadIdx.x, NT);
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned sg = SIGN ? XsCtl::thread_signs(tab, t) : 0u;   // bit j: this thread's factor of diagonal j's sign (parity-signed table)
  const unsigned dsg = (DIAGCTL && tab.dstring) ? xs_diag_thread_signs(tab, t) : 0u;
  XsDiagCtl dct;
  if (DIAGCTL) dct.load(tab, __builtin_ctz(NT));
  XsCtl ctl, ctlh;  // (ctlh: the order of the pairing sweeps -- every partner from LDS, phase 0 first)
  ctl.load_table(tab, ord, lane);
  if (SIGN) ctl.load_row_signs(tab, lane, __builtin_ctz(NT));
  ctlh.load_table(tab, ordh, lane);
  if (SIGN) ctlh.load_row_signs(tab, lane, __builtin_ctz(NT));
  T2* pk = park + (long)blockIdx.x * L * D;  // [l][r * NT + t] (uniform base, lane index t)
  // in-place update of the work image: every read of the sweep that just ran is done before any row is overwritten
  // (split sweeps as in statecol_bwdq.hip: publish_begin = barrier + the thread's writes; the barrier that makes the image visible
  // follows the sweep's register-only work -- the register classes of the plain sweeps, the bracket of the Horner levels)
  // (512 threads and fewer: the 1024-thread shapes have 128 registers and spill what lives across one more barrier)
  constexpr bool SPLIT = NT <= 512;
  auto publish_begin = [&](const T2* v) {
    __syncthreads();
    xs_put<T, NT, RPT>(W, t, v);
    if constexpr (!SPLIT) __syncthreads();
  };
  bool first = true;
#ifdef XS_STAMPS
  long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_t = clock64();
#endif
  for (int b = blockIdx.x; b < B; b += gridDim.x, first = false) {
    T2 cur[RPT];
    {
      const T2* in = psi + (long)b * D;
      const T2* li = lam + (long)b * D;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const T2 a = in[r * NT + t], l = li[r * NT + t];   // (pr, pi): into the frame of the centred diagonal (xs_center_launch)
        cur[r].x = a.x * pr - a.y * pi;
        cur[r].y = a.x * pi + a.y * pr;
        T2 lv;
        lv.x = l.x * pr - l.y * pi;
        lv.y = l.x * pi + l.y * pr;
        Lm[r * NT + t] = lv;
      }
    }
    XS_STAMP(0);
    double wnext = ctl.fetch_step(wt + (long)(S - 1) * tab.R, lane), wnexth = ctlh.fetch_step(wt + (long)(S - 1) * tab.R, lane);
    for (int s = S - 1; s >= 0; --s) {
#if XS_W_PREFETCH   // (requested a step ahead: statecol_bwdq.hip)
      ctl.w = wnext;
      ctlh.w = wnexth;
      if (s > 0) {
        wnext = ctl.fetch_step(wt + (long)(s - 1) * tab.R, lane);
        wnexth = ctlh.fetch_step(wt + (long)(s - 1) * tab.R, lane);
      }
#else
      ctl.load_step(wt + (long)s * tab.R, lane);
      ctlh.load_step(wt + (long)s * tab.R, lane);
#endif
      const double* dgr = dgs + (long)s * dgs_step;
      {  // (scope of dg: the reversal and the forward terms keep the rows' diagonal weights in registers)
      T dg[RPT];
#pragma unroll
      for (int r = 0; r < RPT; ++r) dg[r] = (T)dgr[t * RPT + r];
      // ---- reversal: psi_s = sum_m (-A)^m psi_{s+1} / m! -------------------------------------------------------------
      publish_begin(cur);
      {
        T2 acc[RPT];
        const T g0 = (T)pol.g[0];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          acc[r].x = g0 * cur[r].x;
          acc[r].y = g0 * cur[r].y;
        }
        for (int m = 1; m <= L; ++m) {
          T2 o[RPT];
#pragma unroll
          for (int r = 0; r < RPT; ++r) {
            o[r].x = 0;
            o[r].y = 0;
          }
          if constexpr (SPLIT) {
            xs_gather_local<T, RPT, SIGN, CPLX>(ctl, sg, cur, (T)-1, o);
            __syncthreads();
            xs_gather_lds<T, NT, RPT, SIGN, 1, PREF, CPLX>(ctl, sg, W, t, (T)-1, o);
          } else {
            xs_gather<T, NT, RPT, SIGN, false, 1, PREF, true, CPLX>(ctl, sg, W, cur, t, (T)-1, o, nullptr, nullptr);
          }
          const T inv = (T)1 / (T)m, gm = (T)pol.g[m];
#pragma unroll
          for (int r = 0; r < RPT; ++r) {  // + i dg v
            const T ox = xs_fma(-dg[r], cur[r].y, o[r].x) * inv;
            const T oy = xs_fma(dg[r], cur[r].x, o[r].y) * inv;
            cur[r].x = ox;
            cur[r].y = oy;
            acc[r].x = xs_fma(gm, ox, acc[r].x);
            acc[r].y = xs_fma(gm, oy, acc[r].y);
          }
          if (m < L) publish_begin(cur);
        }
#pragma unroll
        for (int r = 0; r < RPT; ++r) cur[r] = acc[r];
      }
      XS_STAMP(1);
      // ---- the forward terms x_l = A^l psi_s / l! --------------------------------------------------------------------
#ifndef XS_NOPARK
#pragma unroll
      for (int r = 0; r < RPT; ++r) pk[r * NT + t] = cur[r];
#endif
      if (L > 1) publish_begin(cur);
      for (int l = 1; l < L; ++l) {
        T2 o[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          o[r].x = 0;
          o[r].y = 0;
        }
        if constexpr (SPLIT) {
          xs_gather_local<T, RPT, SIGN, CPLX>(ctl, sg, cur, (T)1, o);
          __syncthreads();
          xs_gather_lds<T, NT, RPT, SIGN, 1, PREF, CPLX>(ctl, sg, W, t, (T)1, o);
        } else {
          xs_gather<T, NT, RPT, SIGN, false, 1, PREF, true, CPLX>(ctl, sg, W, cur, t, (T)1, o, nullptr, nullptr);
        }
        const T inv = (T)1 / (T)l;
        T2* pl = pk + (long)l * D;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const T ox = xs_fma(dg[r], cur[r].y, o[r].x) * inv;
          const T oy = xs_fma(-dg[r], cur[r].x, o[r].y) * inv;
          cur[r].x = ox;
          cur[r].y = oy;
#ifndef XS_NOPARK
          pl[r * NT + t] = cur[r];
#endif
        }
        if (l < L - 1) publish_begin(cur);
      }
      }  // (dg)
      XS_STAMP(2);
      // ---- Horner on lambda with the level pairings ----------------------------------------------------------------
      static_assert(RM == 16 || RM == 32, "one or two accumulator vectors of 16 lanes");
      xs_acc16 dacc[RM / 16];
#pragma unroll
      for (int g = 0; g < RM / 16; ++g) dacc[g] = 0.0;
      double rowacc[DIAGCTL ? RPT : 1];
#pragma unroll
      for (int r = 0; r < (DIAGCTL ? RPT : 1); ++r) rowacc[r] = 0.0;
      T2 xm[RPT];
      {
        // x_{L-1} is still in the registers (cur): it is this sweep's pairing partner
        const T inv = (T)pol.g[L] / (T)L;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          xm[r] = cur[r];
          const T2 g = Lm[r * NT + t];
          cur[r].x = g.x * inv;
          cur[r].y = g.y * inv;
        }
      }
      publish_begin(cur);
      for (int l = L - 1; l >= 0; --l) {  // forms y_{l-1} from y_l (l = 0: lambda_s), pairing y_l with x_l
        T2 o[RPT];
        if (l < L - 1) {  // this level's forward term (x_{L-1} is still in the registers).  No prefetch of the next one: its buffer
          const T2* pl = pk + (long)l * D;  // costs more in spills than the load's latency (22.9 against 24.8 ms per 20 steps)
#ifndef XS_NOPARK
#pragma unroll
          for (int r = 0; r < RPT; ++r) xm[r] = pl[r * NT + t];
#else
          (void)pl;
#endif
        }
        const T gl = (T)pol.g[l];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const T dgv = (T)dgr[t * RPT + r];  // (L2-resident table; not kept in registers across the sweep)
          const T2 g = Lm[r * NT + t];
          o[r].x = xs_fma(-dgv, cur[r].y, gl * g.x);  // g_l lam - (-i dg y)
          o[r].y = xs_fma(dgv, cur[r].x, gl * g.y);
          if (DIAGCTL) rowacc[r] += (double)cur[r].x * (double)xm[r].y - (double)cur[r].y * (double)xm[r].x;
        }
        if constexpr (SPLIT) __syncthreads();   // (the image of y_l: written in front of this level's forward term and bracket)
        xs_gather_pairings<T, NT, RPT, SIGN, RM, CPLX>(ctlh, sg, W, t, (T)-1, o, xm, dacc);
        if (l > 0) {
          const T inv = (T)1 / (T)l;
#pragma unroll
          for (int r = 0; r < RPT; ++r) {
            cur[r].x = o[r].x * inv;
            cur[r].y = o[r].y * inv;
          }
          publish_begin(cur);
        } else {
#pragma unroll
          for (int r = 0; r < RPT; ++r) Lm[r * NT + t] = o[r];
        }
      }
      XS_STAMP(3);
      // ---- this step's sums: per wave, accumulated over the workgroup's columns --------------------------------------
      double* prow = part + (((long)blockIdx.x * S + s) * NW + wave) * NS;
#pragma unroll
      for (int g = 0; g < RM / 16; ++g) {
        double da[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) da[j] = dacc[g][j];
        const double v = xs_wave_sums16(da, lane);
        if (lane < 16) xs_part_add(prow + g * 16 + lane, v, first);
      }
      if (DIAGCTL) {
        double dv[XR_DMAX];
        if (tab.dstring) {   // Z-string controls: signs from the masks (statecol_dev.h), no table reads
          xs_diag_string_sums<RPT>(dct, rowacc, dsg, dv);
        } else {
#pragma unroll
          for (int jd = 0; jd < XR_DMAX; ++jd) dv[jd] = 0.0;
#pragma clang loop unroll(disable)
          for (int jd = 0; jd < tab.nd; ++jd) {  // (runtime loop + tree select: unrolled, the 16 x RPT loads are hoisted and spill)
            if (tab.dkid[jd] < 0) continue;
            const double* dvr = tab.dval + (long)jd * D;
            double v = 0.0;
#pragma unroll
            for (int r = 0; r < RPT; ++r) v = __builtin_fma(dvr[r * NT + t], rowacc[r], v);
            xs_acc<0, XR_DMAX>(dv, jd, v);
          }
        }
        const double v = xs_wave_sums16(dv, lane);
        if (lane < 16) xs_part_add(prow + RM + lane, v, first);
      }
      // psi_s = x_0 is in the registers (the last sweep's pairing partner): the next (earlier) step starts from it
#pragma unroll
      for (int r = 0; r < RPT; ++r) cur[r] = xm[r];
      XS_STAMP(4);
    }
    {
      T2* out = psi + (long)b * D;
      T2* lo = lam + (long)b * D;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        out[r * NT + t] = cur[r];
        lo[r * NT + t] = Lm[r * NT + t];
      }
    }
    __syncthreads();  // (the last sweep's reads of W are done before the next column overwrites it)
    XS_STAMP(5);
  }
#ifdef XS_STAMPS
  if (t == 0)
    for (int i = 0; i < 8; ++i) atomicAdd(&xs_stamps[i], (unsigned long long)stamp_acc[i]);
#endif
}

namespace {

template <class T, int NT, int RPT, bool PREF, int RM>
hipError_t xs_backward_rm(typename Cx<T>::type* psi, typename Cx<T>::type* lam, const XrTab& tab, const double* wt, const double* dgs,
                          long dgs_step, int S, int B, int L, int nblk, typename Cx<T>::type* park, double* part, const XsPoly& pol, double pr, double pi, hipStream_t st) {
  typedef typename Cx<T>::type T2;
  const size_t lds = xs_lds_bytes(NT * RPT, sizeof(T));
  const bool sgn = tab.sgn != nullptr, dc = xs_has_diag_controls(tab);
  bool cplx = false;
  for (int j = 0; j < tab.R; ++j) cplx = cplx || tab.phase[j] != 0;
  if (cplx && !sgn) return hipErrorInvalidValue;
  // (SIGN = 2 for every signed table: the exact form is the fallback of the quadrature kernels, one instantiation per class)
  void (*kf)(T2*, T2*, XrTab, XsOrder, XsOrder, const double*, const double*, long, int, int, int, T2*, double*, XsPoly, T, T) =
      cplx ? (dc ? xs_backward_kernel<T, NT, RPT, 2, true, RM, PREF, true> : xs_backward_kernel<T, NT, RPT, 2, false, RM, PREF, true>)
           : (sgn ? (dc ? xs_backward_kernel<T, NT, RPT, 2, true, RM, PREF, false>
                        : xs_backward_kernel<T, NT, RPT, 2, false, RM, PREF, false>)
                  : (dc ? xs_backward_kernel<T, NT, RPT, 0, true, RM, PREF, false>
                        : xs_backward_kernel<T, NT, RPT, 0, false, RM, PREF, false>));
  hipError_t e = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kf, dim3((unsigned)nblk), dim3(NT), lds, st, psi, lam, tab, xs_make_order(NT * RPT, tab, xs_threads(NT * RPT, 0) == NT ? 0 : 1, true),
                     xs_make_order(NT * RPT, tab, xs_threads(NT * RPT, 0) == NT ? 0 : 1, false), wt, dgs, dgs_step, S, B, L, park, part, pol, (T)pr, (T)pi);
  return hipGetLastError();
}

template <class T, int NT, int RPT, bool PREF>
hipError_t xs_backward_go(typename Cx<T>::type* psi, typename Cx<T>::type* lam, const XrTab& tab, const double* wt, const double* dgs,
                          long dgs_step, int S, int B, int L, int nblk, typename Cx<T>::type* park, double* part, const XsPoly& pol, double pr, double pi, hipStream_t st) {
  return xs_rm(tab) > XR_RSMALL
             ? xs_backward_rm<T, NT, RPT, PREF, 2 * XR_RSMALL>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, pr, pi, st)
             : xs_backward_rm<T, NT, RPT, PREF, XR_RSMALL>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, pr, pi, st);
}

template <class T, bool PREF>
hipError_t xs_backward_pick(typename Cx<T>::type* psi, typename Cx<T>::type* lam, int d, const XrTab& tab, const double* wt,
                            const double* dgs, long dgs_step, int S, int B, int L, int nblk, typename Cx<T>::type* park, double* part,
                            const XsPoly& pol, int variant, double pr, double pi, hipStream_t st) {
  if constexpr (sizeof(T) == sizeof(float)) {
    if (d == 8192) return xs_backward_go<T, 1024, 8, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, pr, pi, st);
  }
  switch (d) {
    case 4096:
      return (variant & 1) ? xs_backward_go<T, 1024, 4, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, pr, pi, st)
                           : xs_backward_go<T, 512, 8, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, pr, pi, st);
#ifndef XS_PROBE_4096_ONLY
    case 2048: return xs_backward_go<T, 256, 8, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, pr, pi, st);
    case 1024:
      return (variant & 1) ? xs_backward_go<T, 256, 4, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, pr, pi, st)
                           : xs_backward_go<T, 128, 8, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, pr, pi, st);
    case 512:
      return (variant & 1) ? xs_backward_go<T, 256, 2, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, pr, pi, st)
                           : xs_backward_go<T, 64, 8, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, pr, pi, st);
    default:
      return (variant & 1) ? xs_backward_go<T, 256, 1, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, pr, pi, st)
                           : xs_backward_go<T, 64, 4, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, pr, pi, st);
#else
    default: return hipErrorInvalidValue;
#endif
  }
}

}  // namespace

template <class T>
hipError_t xs_backward_exact_launch(typename Cx<T>::type* psi, typename Cx<T>::type* lam, int d, const XrTab& tab, const double* wt,
                                    const double* dgs, long dgs_step, int S, int B, int L, int nblk, typename Cx<T>::type* park,
                                    double* part, const XsPoly& pol, int variant, double pr, double pi, hipStream_t st) {
  // (no prefetch in the backward kernel: its sweeps carry more live registers, and the second partner buffer costs more in spills
  // than it hides in latency -- measured 24.5 against 26.7 ms per 20 steps at d = B = 4096)
  return xs_backward_pick<T, false>(psi, lam, d, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, pol, variant, pr, pi, st);
}
template hipError_t xs_backward_exact_launch<double>(double2*, double2*, int, const XrTab&, const double*, const double*, long, int, int, int,
                                                     int, double2*, double*, const XsPoly&, int, double, double, hipStream_t);
template hipError_t xs_backward_exact_launch<float>(float2*, float2*, int, const XrTab&, const double*, const double*, long, int, int, int,
                                                    int, float2*, double*, const XsPoly&, int, double, double, hipStream_t);

}  // namespace dq
