// State-column form, adjoint sweep in its QUADRATURE form (round 4): 2 L sweeps per step, no scratch.
#include "statecol_dev.h"

namespace dq {

// ---- backward sweep, quadrature form (round 4) ---------------------------------------------------------------------------------
// The same quantities as xs_backward_kernel with 2 L sweeps per step instead of 3 L - 1, no forward-term family and no scratch.
// With phi(s) = exp(-s A) phi_{t+1} for phi = psi, lambda (both run backward from t + 1 under the same operator, A anti-Hermitian)
//     dL/du_k[t] = Re <lambda_{t+1}, dexp(A)[dA_k] psi_t> = Re int_0^1 <lambda(s), dA_k psi(s)> ds
// (dexp(A)[E] = int exp((1-s')A) E exp(s'A) ds', s = 1 - s'), and the integrand is smooth on the scale of the step: its p-th
// derivative is <lambda(s), ad_A^p(dA_k) psi(s)>, bounded by (2 theta)^p.  A Gauss-Lobatto rule with NI interior nodes (exact to
// degree 2 NI + 1; the host picks NI from theta so that the bound of the rule's error stays below 2^-40, xs_quad_nodes) needs the
// node values lambda(s_i) = sum_m s_i^m z_m, z_m = (-A)^m lambda_{t+1} / m! -- sums of the Taylor terms the propagation of lambda
// produces anyway -- and pairs them with psi(s_i) = sum_m s_i^m r_m, r_m = (-A)^m psi_{t+1} / m!:
//     sum_i w_i <lambda(s_i), dA_k psi(s_i)> = sum_m <q_m, dA_k r_m>,     q_m = sum_i w_i s_i^m lambda(s_i)
// so the sweep that forms r_{m+1} from r_m pairs the partner values r_m[i ^ x_j] it fetches with the thread's own rows of q_m,
// exactly as the Horner sweeps of xs_backward_kernel pair y_l with x_l.  Per column and step:
//     phase 1 (lambda, LDS-resident between steps): L plain sweeps; accumulates lambda_t and the NI interior node values; then
//             q_0 = w_0 lambda_{t+1} + sum_i w_i lambda(s_i) + w_1 lambda_t (the end points 0 and 1 are nodes of the rule)
//     phase 2 (psi, register-resident between steps): L pairing sweeps; accumulates psi_t; q_m = sum_i w_i s_i^m lambda(s_i) + w_1 lambda_t.
// The numpy twin is tests/np_state_column.py::backward_step_quad.  qd.c1[m][i] = s_i^m, qd.c2[m][i] = -w_i s_i^m (the sign:
// <q, dA r> = -<r, dA q>, and the pairing code takes the fetched partners as its first factor), qd.w0 = -w_0, qd.w1[m] = -w_1.
// With an economized step polynomial (XsPoly, dq_internal.h) exp(-s A) ~ sum_m g^(s)[m] (-s A)^m / m! with level factors of the
// interval s theta: every s_i^m above carries its g_i[m] (folded into c1 / c2 by the host), lambda_t and psi_t accumulate
// qd.g[m] z_m / qd.g[m] r_m, and the end point 1 pairs r_m with g[m] lambda_t (qd.w1[m] = -w_1 g[m]).
template <class T, int NT, int RPT, bool SIGN, bool DIAGCTL, bool CPLX, int NI>
__global__ void __launch_bounds__(NT) xs_backward_q_kernel(typename Cx<T>::type* __restrict__ psi, typename Cx<T>::type* __restrict__ lam,
                                                           XrTab tab, XsOrder ord, XsOrder ordh, const double* __restrict__ wt,
                                                           const double* __restrict__ dgs, long dgs_step, int S, int B, int L, XsQuad qd,
                                                           double* __restrict__ part) {
  typedef typename Cx<T>::type T2;
  constexpr int D = NT * RPT;
  constexpr int NW = NT / 64;
  constexpr int RM = XR_RSMALL;
  constexpr int NS = RM + (DIAGCTL ? XR_DMAX : 0);
  extern __shared__ __align__(16) unsigned char xs_lds_raw[];
  T2* W = (T2*)xs_lds_raw;
  T2* Lm = W + D;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned sg = SIGN ? XsCtl::thread_signs(tab, t) : 0u;   // bit j: this thread's factor of diagonal j's sign (parity-signed table)
  XsCtl ctl, ctlh;
  ctl.load_table(tab, ord, lane);
  if (SIGN) ctl.load_row_signs(tab, lane, __builtin_ctz(NT));
  ctlh.load_table(tab, ordh, lane);
  if (SIGN) ctlh.load_row_signs(tab, lane, __builtin_ctz(NT));
  auto publish = [&](const T2* v) {
    __syncthreads();
    xs_put<T, NT, RPT>(W, t, v);
    __syncthreads();
  };
  bool first = true;
#ifdef XS_STAMPS
  long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_t = clock64();
#endif
  for (int b = blockIdx.x; b < B; b += gridDim.x, first = false) {
    T2 cur[RPT];  // psi_{s+1}: the thread's rows, in registers between steps
    {
      const T2* in = psi + (long)b * D;
      const T2* li = lam + (long)b * D;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        cur[r] = in[r * NT + t];
        Lm[r * NT + t] = li[r * NT + t];
      }
    }
    XS_STAMP(0);
    for (int s = S - 1; s >= 0; --s) {
      ctl.load_step(wt + (long)s * tab.R, lane);
      ctlh.load_step(wt + (long)s * tab.R, lane);
      const double* dgr = dgs + (long)s * dgs_step;
      T2 nd[NI][RPT];  // the interior node values lambda(s_i)
      T2 q[RPT];
      // ---- phase 1: lambda_s = sum_m z_m, node values ---------------------------------------------------------------------
      {
        T2 zc[RPT], acc[RPT];
        const T g0 = (T)qd.g[0];
        T cf0[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) cf0[i] = (T)qd.c1[0][i];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          zc[r] = Lm[r * NT + t];
          acc[r].x = g0 * zc[r].x;
          acc[r].y = g0 * zc[r].y;
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            nd[i][r].x = cf0[i] * zc[r].x;
            nd[i][r].y = cf0[i] * zc[r].y;
          }
        }
        publish(zc);
        for (int m = 1; m <= L; ++m) {
          T2 o[RPT];
#pragma unroll
          for (int r = 0; r < RPT; ++r) {
            o[r].x = 0;
            o[r].y = 0;
          }
          xs_gather<T, NT, RPT, SIGN, false, 1, false, true, CPLX, true>(ctl, sg, W, zc, t, (T)-1, o, nullptr, nullptr);
          const T inv = (T)1 / (T)m, gm = (T)qd.g[m];
          T cf[NI];
#pragma unroll
          for (int i = 0; i < NI; ++i) cf[i] = (T)qd.c1[m][i];
#pragma unroll
          for (int r = 0; r < RPT; ++r) {  // + i dg v
            const T dgv = (T)dgr[r * NT + t];
            const T ox = xs_fma(-dgv, zc[r].y, o[r].x) * inv;
            const T oy = xs_fma(dgv, zc[r].x, o[r].y) * inv;
            zc[r].x = ox;
            zc[r].y = oy;
            acc[r].x = xs_fma(gm, ox, acc[r].x);
            acc[r].y = xs_fma(gm, oy, acc[r].y);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
              nd[i][r].x = xs_fma(cf[i], ox, nd[i][r].x);
              nd[i][r].y = xs_fma(cf[i], oy, nd[i][r].y);
            }
          }
          if (m < L) publish(zc);
        }
        T cw[NI];
#pragma unroll
        for (int i = 0; i < NI; ++i) cw[i] = (T)qd.c2[0][i];
        const T w0 = (T)qd.w0, w1 = (T)qd.w1[0];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const T2 g = Lm[r * NT + t];
          T qx = xs_fma(w0, g.x, w1 * acc[r].x), qy = xs_fma(w0, g.y, w1 * acc[r].y);
#pragma unroll
          for (int i = 0; i < NI; ++i) {
            qx = xs_fma(cw[i], nd[i][r].x, qx);
            qy = xs_fma(cw[i], nd[i][r].y, qy);
          }
          q[r].x = qx;
          q[r].y = qy;
          Lm[r * NT + t] = acc[r];  // lambda_s (own rows only: no barrier concerns it)
        }
      }
      XS_STAMP(1);
      // ---- phase 2: psi_s = sum_m r_m with the level pairings -----------------------------------------------------------------
      xs_acc16 dacc[1];
      dacc[0] = 0.0;
      double rowacc[DIAGCTL ? RPT : 1];
#pragma unroll
      for (int r = 0; r < (DIAGCTL ? RPT : 1); ++r) rowacc[r] = 0.0;
      T2 acc[RPT];
      {
        const T g0 = (T)qd.g[0];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          acc[r].x = g0 * cur[r].x;
          acc[r].y = g0 * cur[r].y;
        }
      }
      publish(cur);
      for (int m = 0; m < L; ++m) {
        if (m > 0) {
          T cw[NI];
#pragma unroll
          for (int i = 0; i < NI; ++i) cw[i] = (T)qd.c2[m][i];
          const T w1 = (T)qd.w1[m];
#pragma unroll
          for (int r = 0; r < RPT; ++r) {
            const T2 g = Lm[r * NT + t];
            T qx = w1 * g.x, qy = w1 * g.y;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
              qx = xs_fma(cw[i], nd[i][r].x, qx);
              qy = xs_fma(cw[i], nd[i][r].y, qy);
            }
            q[r].x = qx;
            q[r].y = qy;
          }
        }
        T2 o[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const T dgv = (T)dgr[r * NT + t];
          o[r].x = -dgv * cur[r].y;  // i dg r_m
          o[r].y = dgv * cur[r].x;
          if (DIAGCTL) rowacc[r] += (double)cur[r].x * (double)q[r].y - (double)cur[r].y * (double)q[r].x;
        }
#if XS_PAIR_LOCAL
        // the partners at offsets without lane bits are the thread's own rows of r_m: those diagonals FIRST, while cur is still
        // live and the partner ring of the LDS classes is not yet (the two share registers in time), then the LDS classes
        xs_local_classes<T, RPT, 1, SIGN, true, CPLX>(ctl, ctl.n0 + ctl.n1, sg, cur, (T)-1, o, q, (double*)&dacc[0]);
        xs_lds_group<T, NT, RPT, SIGN, true, RM, false, false, 0>(ctl, 0, ctl.n0 - ctl.n0p1, sg, W, t, (T)-1, o, q, (double*)&dacc[0]);
        if constexpr (CPLX)
          xs_lds_group<T, NT, RPT, SIGN, true, RM, false, false, 1>(ctl, ctl.n0 - ctl.n0p1, ctl.n0p1, sg, W, t, (T)-1, o, q, (double*)&dacc[0]);
        xs_lds_group<T, NT, RPT, SIGN, true, RM, true, false, 0>(ctl, ctl.n0, ctl.n1 - ctl.n1p1, sg, W, t, (T)-1, o, q, (double*)&dacc[0]);
        if constexpr (CPLX)
          xs_lds_group<T, NT, RPT, SIGN, true, RM, true, false, 1>(ctl, ctl.n0 + ctl.n1 - ctl.n1p1, ctl.n1p1, sg, W, t, (T)-1, o, q, (double*)&dacc[0]);
#else
        xs_gather_pairings<T, NT, RPT, SIGN, RM, CPLX>(ctlh, sg, W, t, (T)-1, o, q, dacc);
#endif
        const T inv = (T)1 / (T)(m + 1), gm = (T)qd.g[m + 1];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          cur[r].x = o[r].x * inv;
          cur[r].y = o[r].y * inv;
          acc[r].x = xs_fma(gm, cur[r].x, acc[r].x);
          acc[r].y = xs_fma(gm, cur[r].y, acc[r].y);
        }
        if (m + 1 < L) publish(cur);
      }
#pragma unroll
      for (int r = 0; r < RPT; ++r) cur[r] = acc[r];
      XS_STAMP(3);
      // ---- this step's sums: per wave, accumulated over the workgroup's columns --------------------------------------
      double* prow = part + (((long)blockIdx.x * S + s) * NW + wave) * NS;
      {
        double da[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) da[j] = dacc[0][j];
        const double v = xs_wave_sums16(da, lane);
        if (lane < 16) prow[lane] = first ? v : prow[lane] + v;
      }
      if (DIAGCTL) {
        double dv[XR_DMAX];
#pragma unroll
        for (int jd = 0; jd < XR_DMAX; ++jd) dv[jd] = 0.0;
#pragma clang loop unroll(disable)
        for (int jd = 0; jd < tab.nd; ++jd) {
          if (tab.dkid[jd] < 0) continue;
          const double* dvr = tab.dval + (long)jd * D;
          double v = 0.0;
#pragma unroll
          for (int r = 0; r < RPT; ++r) v = __builtin_fma(dvr[r * NT + t], rowacc[r], v);
          xs_acc<0, XR_DMAX>(dv, jd, v);
        }
        const double v = xs_wave_sums16(dv, lane);
        if (lane < 16) prow[RM + lane] = first ? v : prow[RM + lane] + v;
      }
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

// the quadrature form (tables of up to 16 XOR-diagonals, 4-point rule)
template <class T, int NT, int RPT>
hipError_t xs_backward_q_go(typename Cx<T>::type* psi, typename Cx<T>::type* lam, const XrTab& tab, const double* wt, const double* dgs,
                            long dgs_step, int S, int B, int L, int nblk, const XsQuad& qd, double* part, hipStream_t st) {
  typedef typename Cx<T>::type T2;
  if (!xs_quad_supported(tab, qd.ni, NT * RPT, xs_threads(NT * RPT, 0) == NT ? 0 : 1)) return hipErrorInvalidValue;
  const size_t lds = xs_lds_bytes(NT * RPT, sizeof(T));
  // (xs_quad_supported: unsigned tables without diagonal controls only -- the one instantiation of the kernel's flag space that is built)
  void (*kf)(T2*, T2*, XrTab, XsOrder, XsOrder, const double*, const double*, long, int, int, int, XsQuad, double*) =
      xs_backward_q_kernel<T, NT, RPT, false, false, false, 2>;
  hipError_t e = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const int var = xs_threads(NT * RPT, 0) == NT ? 0 : 1;
  hipLaunchKernelGGL(kf, dim3((unsigned)nblk), dim3(NT), lds, st, psi, lam, tab, xs_make_order(NT * RPT, tab, var, true),
                     xs_make_order(NT * RPT, tab, var, false), wt, dgs, dgs_step, S, B, L, qd, part);
  return hipGetLastError();
}

template <class T>
hipError_t xs_backward_q_pick(typename Cx<T>::type* psi, typename Cx<T>::type* lam, int d, const XrTab& tab, const double* wt,
                              const double* dgs, long dgs_step, int S, int B, int L, int nblk, const XsQuad& qd, double* part,
                              int variant, hipStream_t st) {
  // (no 1024-thread shapes -- d = 8192 in complex64, the wide form of d = 4096: xs_quad_supported keeps them on the exact kernel)
  switch (d) {
    case 4096:
      return (variant & 1) ? hipErrorInvalidValue
                           : xs_backward_q_go<T, 512, 8>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, st);
#ifndef XS_PROBE_4096_ONLY
    case 2048: return xs_backward_q_go<T, 256, 8>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, st);
    case 1024:
      return (variant & 1) ? xs_backward_q_go<T, 256, 4>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, st)
                           : xs_backward_q_go<T, 128, 8>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, st);
    case 512:
      return (variant & 1) ? xs_backward_q_go<T, 256, 2>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, st)
                           : xs_backward_q_go<T, 64, 8>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, st);
    default:
      return (variant & 1) ? xs_backward_q_go<T, 256, 1>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, st)
                           : xs_backward_q_go<T, 64, 4>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, st);
#else
    default: return hipErrorInvalidValue;
#endif
  }
}

}  // namespace

template <class T>
hipError_t xs_backward_quad_launch(typename Cx<T>::type* psi, typename Cx<T>::type* lam, int d, const XrTab& tab, const double* wt,
                                   const double* dgs, long dgs_step, int S, int B, int L, int nblk, const XsQuad& qd, double* part,
                                   int variant, hipStream_t st) {
  return xs_backward_q_pick<T>(psi, lam, d, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, variant, st);
}
template hipError_t xs_backward_quad_launch<double>(double2*, double2*, int, const XrTab&, const double*, const double*, long, int, int, int,
                                                    int, const XsQuad&, double*, int, hipStream_t);
template hipError_t xs_backward_quad_launch<float>(float2*, float2*, int, const XrTab&, const double*, const double*, long, int, int, int, int,
                                                   const XsQuad&, double*, int, hipStream_t);

}  // namespace dq
