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
template <class T, int NT, int RPT, int SIGN, bool DIAGCTL, bool CPLX, int NI, bool U2L>
__global__ void __launch_bounds__(NT) xs_backward_q_kernel(typename Cx<T>::type* __restrict__ psi, typename Cx<T>::type* __restrict__ lam,
                                                           XrTab tab, XsOrder ord, XsOrder ordh, const double* __restrict__ wt,
                                                           const double* __restrict__ dgs, long dgs_step, int S, int B, int L, XsQuad qd,
                                                           double* __restrict__ part, T pr, T pi) {
  typedef typename Cx<T>::type T2;
  constexpr int D = NT * RPT;
  constexpr int NW = NT / 64;
  constexpr int RM = XR_RSMALL;
  // the one-instruction accumulator update (xs_acc_add) in the kernels without diagonal controls: complex128, 8 rows per thread
  constexpr bool ACCA = !DIAGCTL && sizeof(T) == sizeof(double) && RPT == 8 && NT == 512;   // (d = 4096: the shape the pinned tuple was checked on)
  constexpr int RMP = ACCA ? 17 : RM;   // (what the pairing sweeps' LDS classes are told: statecol_dev.h, xs_lds_class)
  constexpr int NS = RM + (DIAGCTL ? XR_DMAX : 0);
  extern __shared__ __align__(16) unsigned char xs_lds_raw[];
  T2* W = (T2*)xs_lds_raw;
  T2* Lm = W + D;
  xs_lds_aligned<NT>(W);
  xs_stagger_priority(threadIdx.x, NT);
  // U2L (difference basis, XsQuad::basis = 1; complex128, 8 rows per thread): the rule's second vector u_2 = O(theta^2) |lambda|
  // spends phase 2 in the LDS a column's two images leave free, rounded to fp32 (8 B per row: 32 of the 160 KB at d = 4096), two rows
  // to a 16-byte slot [k NT + t].  Relative to |lambda| that rounding is 6e-8 x 0.16 theta^2 (2e-11 at theta = 0.05), and it enters q_m
  // with the weight w s^m <= 0.12 from m = 1 on, where the pairing partner r_m is O(theta^m) itself: 1e-13 of the gradient's scale
  // (q_0, whose partner is psi itself, is formed while u_2 is still in its fp64 registers: accumulating u_2 in fp32 as well measured
  // 1e-12 instead of 1e-14 against the exact form).  During phase 1 the same slots hold rows 0 .. 3 of psi_{s+1}, which has nothing to
  // do there: phase 1 then holds 4 rows of psi, zc, acc, u_1, u_2, o and the partner buffer = 208 registers, phase 2 o, q, acc, u_1,
  // 16 accumulators, the partner ring (+ 8 row accumulators with diagonal controls) = 176 - 208.  Every slot is written and read by
  // its own thread only: no barrier concerns them.
  static_assert(!U2L || (sizeof(T) == sizeof(double) && RPT == 8 && NI == 2), "difference-basis kernel: complex128, 8 rows per thread");
  typedef float xs_f4 __attribute__((ext_vector_type(4)));
  T2* Pk = Lm + D;              // [k NT + t], k < RPT / 2: psi rows during phase 1 ...
  xs_f4* U2v = (xs_f4*)(Lm + D);   // ... and (u_2[2k], u_2[2k + 1]) during phase 2
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const unsigned sg = SIGN ? XsCtl::thread_signs(tab, t) : 0u;   // bit j: this thread's factor of diagonal j's sign (parity-signed table)
  const unsigned dsg = (DIAGCTL && tab.dstring) ? xs_diag_thread_signs(tab, t) : 0u;
  XsDiagCtl dct;
  if (DIAGCTL) dct.load(tab, __builtin_ctz(NT));
  XsCtl ctl, ctlh;
  ctl.load_table(tab, ord, lane);
  if (SIGN) ctl.load_row_signs(tab, lane, __builtin_ctz(NT));
  ctlh.load_table(tab, ordh, lane);
  if (SIGN) ctlh.load_row_signs(tab, lane, __builtin_ctz(NT));
  // Split sweeps (round 5, XS_BWD_SPLIT): a sweep's input is published in two halves -- publish_begin (barrier: every wave is done with
  // the old image; then the thread's 8 ds_write_b128) and, BEHIND the sweep's register-only work (the classes whose partners are the
  // thread's own rows; in phase 2 also q_m and the diagonal term), the barrier that makes the image visible to the LDS classes.  The
  // writes drain at a third of the read rate with nothing else of the workgroup running: the vector work now runs under them.
  auto publish_begin = [&](const T2* v) {
    __syncthreads();
    xs_put<T, NT, RPT>(W, t, v);
  };
#if !XS_BWD_SPLIT
  auto publish = [&](const T2* v) {
    publish_begin(v);
    __syncthreads();
  };
#endif
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
      for (int r = 0; r < RPT; ++r) {   // (pr, pi): into the frame of the centred diagonal (dq_internal.h, xs_center_launch)
        const T2 a = in[r * NT + t], l = li[r * NT + t];
        cur[r].x = a.x * pr - a.y * pi;
        cur[r].y = a.x * pi + a.y * pr;
        T2 lv;
        lv.x = l.x * pr - l.y * pi;
        lv.y = l.x * pi + l.y * pr;
        Lm[r * NT + t] = lv;
      }
    }
    XS_STAMP(0);
    double wnext = ctl.fetch_step(wt + (long)(S - 1) * tab.R, lane);
    for (int s = S - 1; s >= 0; --s) {
      // (the step's weights were requested a step ahead: the load sat at the head of every step with nothing to hide it)
#if XS_W_PREFETCH
      ctl.w = wnext;
      if (s > 0) wnext = ctl.fetch_step(wt + (long)(s - 1) * tab.R, lane);
#else
      ctl.load_step(wt + (long)s * tab.R, lane);
#endif
      ctlh.load_step(wt + (long)s * tab.R, lane);
      const double* dgr = dgs + (long)s * dgs_step;
      T2 nd[NI][RPT];  // the interior node values lambda(s_i)
      T2 q[RPT];
      T2 u2f[U2L ? RPT : 1];   // u_2 in the difference-basis kernels (nd[1] is not used there): fp64 while phase 1 accumulates it and forms q_0
      if constexpr (U2L) {
#pragma unroll
        for (int k = 0; k < RPT / 2; ++k) Pk[k * NT + t] = cur[k];
      }
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
          for (int i = 0; i < (U2L ? NI - 1 : NI); ++i) {
            nd[i][r].x = cf0[i] * zc[r].x;
            nd[i][r].y = cf0[i] * zc[r].y;
          }
          if constexpr (U2L) {   // (c1[0][1] = c1[1][1] = 0: u_2 starts at the second-order term)
            u2f[r].x = 0;
            u2f[r].y = 0;
          }
        }
#if XS_BWD_SPLIT
        publish_begin(zc);
#else
        publish(zc);
#endif
        for (int m = 1; m <= L; ++m) {
          T2 o[RPT];
#pragma unroll
          for (int r = 0; r < RPT; ++r) {
            o[r].x = 0;
            o[r].y = 0;
          }
#if XS_BWD_SPLIT
          xs_gather_local<T, RPT, SIGN, CPLX>(ctl, sg, zc, (T)-1, o);
          __syncthreads();
          xs_gather_lds<T, NT, RPT, SIGN, (SIGN != 0 ? 2 : 1), false, CPLX>(ctl, sg, W, t, (T)-1, o);   // (signed tables: the partner ring here too)
#else
          xs_gather<T, NT, RPT, SIGN, false, 1, false, true, CPLX, true>(ctl, sg, W, zc, t, (T)-1, o, nullptr, nullptr);
#endif
          // (no division by m: the sweeps run on (-A)^m lambda and (-A)^m psi, the factorials sit in the rule's tables -- the launcher
          // hands over g[m], c1[m], c2[m], w1[m] divided by m!: xs_backward_q_start)
          const T gm = (T)qd.g[m];
          T cf[NI];
#pragma unroll
          for (int i = 0; i < NI; ++i) cf[i] = (T)qd.c1[m][i];
#pragma unroll
          for (int r = 0; r < RPT; ++r) {  // + i dg v
            const T dgv = (T)dgr[t * RPT + r];
            const T ox = xs_fma(-dgv, zc[r].y, o[r].x);
            const T oy = xs_fma(dgv, zc[r].x, o[r].y);
            zc[r].x = ox;
            zc[r].y = oy;
            acc[r].x = xs_fma(gm, ox, acc[r].x);
            acc[r].y = xs_fma(gm, oy, acc[r].y);
#pragma unroll
            for (int i = 0; i < (U2L ? NI - 1 : NI); ++i) {
              nd[i][r].x = xs_fma(cf[i], ox, nd[i][r].x);
              nd[i][r].y = xs_fma(cf[i], oy, nd[i][r].y);
            }
            if constexpr (U2L) {
              u2f[r].x = xs_fma(cf[NI - 1], ox, u2f[r].x);
              u2f[r].y = xs_fma(cf[NI - 1], oy, u2f[r].y);
            }
          }
#if XS_BWD_SPLIT
          if (m < L) publish_begin(zc);
#else
          if (m < L) publish(zc);
#endif
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
          for (int i = 0; i < (U2L ? NI - 1 : NI); ++i) {
            qx = xs_fma(cw[i], nd[i][r].x, qx);
            qy = xs_fma(cw[i], nd[i][r].y, qy);
          }
          if constexpr (U2L) {
            qx = xs_fma(cw[NI - 1], u2f[r].x, qx);
            qy = xs_fma(cw[NI - 1], u2f[r].y, qy);
          }
          q[r].x = qx;
          q[r].y = qy;
          Lm[r * NT + t] = acc[r];  // lambda_s (own rows only: no barrier concerns it)
        }
        if constexpr (U2L) {   // rows 0 .. 3 of psi_{s+1} come back, their slots take u_2
#pragma unroll
          for (int k = 0; k < RPT / 2; ++k) {
            cur[k] = Pk[k * NT + t];
            xs_f4 f;
            f.x = (float)u2f[2 * k].x;
            f.y = (float)u2f[2 * k].y;
            f.z = (float)u2f[2 * k + 1].x;
            f.w = (float)u2f[2 * k + 1].y;
            U2v[k * NT + t] = f;
          }
        }
      }
      XS_STAMP(1);
      // ---- phase 2: psi_s = sum_m r_m with the level pairings -----------------------------------------------------------------
      T2 acc[RPT];
      {
        const T g0 = (T)qd.g[0];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          acc[r].x = g0 * cur[r].x;
          acc[r].y = g0 * cur[r].y;
        }
      }
#if XS_BWD_SPLIT
      publish_begin(cur);
#else
      publish(cur);
#endif
      // (the accumulators are born here, behind the publish: declared at the top of the phase the 32 registers of zeros were spilled
      // across it -- 128 B per thread and step of scratch traffic in the signed kernels)
      xs_acc16 dacc[1];
      dacc[0] = 0.0;
      double rowacc[DIAGCTL ? RPT : 1];
#pragma unroll
      for (int r = 0; r < (DIAGCTL ? RPT : 1); ++r) rowacc[r] = 0.0;
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
            for (int i = 0; i < (U2L ? NI - 1 : NI); ++i) {
              qx = xs_fma(cw[i], nd[i][r].x, qx);
              qy = xs_fma(cw[i], nd[i][r].y, qy);
            }
            q[r].x = qx;
            q[r].y = qy;
          }
          if constexpr (U2L) {
#pragma unroll
            for (int k = 0; k < RPT / 2; ++k) {
              const xs_f4 f = U2v[k * NT + t];
              q[2 * k].x = xs_fma(cw[NI - 1], (T)f.x, q[2 * k].x);
              q[2 * k].y = xs_fma(cw[NI - 1], (T)f.y, q[2 * k].y);
              q[2 * k + 1].x = xs_fma(cw[NI - 1], (T)f.z, q[2 * k + 1].x);
              q[2 * k + 1].y = xs_fma(cw[NI - 1], (T)f.w, q[2 * k + 1].y);
            }
          }
        }
        T2 o[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const T dgv = (T)dgr[t * RPT + r];
          o[r].x = -dgv * cur[r].y;  // i dg r_m
          o[r].y = dgv * cur[r].x;
          if (DIAGCTL) rowacc[r] += (double)cur[r].x * (double)q[r].y - (double)cur[r].y * (double)q[r].x;
        }
#if XS_PAIR_LOCAL
        // the partners at offsets without lane bits are the thread's own rows of r_m: those diagonals FIRST, while cur is still
        // live and the partner ring of the LDS classes is not yet (the two share registers in time), then the LDS classes
        xs_local_classes<T, RPT, 1, SIGN, true, CPLX, ACCA>(ctl, ctl.n0 + ctl.n1, sg, cur, (T)-1, o, q, (double*)&dacc[0]);
#if XS_BWD_SPLIT
        __syncthreads();   // (the image of r_m, written by publish_begin in front of q_m, the diagonal term and the register classes)
#endif
        xs_lds_group<T, NT, RPT, SIGN, true, RMP, false, false, 0>(ctl, 0, ctl.n0 - ctl.n0p1, sg, W, t, (T)-1, o, q, (double*)&dacc[0]);
        if constexpr (CPLX)
          xs_lds_group<T, NT, RPT, SIGN, true, RMP, false, false, 1>(ctl, ctl.n0 - ctl.n0p1, ctl.n0p1, sg, W, t, (T)-1, o, q, (double*)&dacc[0]);
        xs_lds_group<T, NT, RPT, SIGN, true, RMP, true, false, 0>(ctl, ctl.n0, ctl.n1 - ctl.n1p1, sg, W, t, (T)-1, o, q, (double*)&dacc[0]);
        if constexpr (CPLX)
          xs_lds_group<T, NT, RPT, SIGN, true, RMP, true, false, 1>(ctl, ctl.n0 + ctl.n1 - ctl.n1p1, ctl.n1p1, sg, W, t, (T)-1, o, q, (double*)&dacc[0]);
#else
#if XS_BWD_SPLIT
        __syncthreads();
#endif
        xs_gather_pairings<T, NT, RPT, SIGN, RM, CPLX>(ctlh, sg, W, t, (T)-1, o, q, dacc);
#endif
        const T gm = (T)qd.g[m + 1];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          cur[r].x = o[r].x;
          cur[r].y = o[r].y;
          acc[r].x = xs_fma(gm, cur[r].x, acc[r].x);
          acc[r].y = xs_fma(gm, cur[r].y, acc[r].y);
        }
#if XS_BWD_SPLIT
        if (m + 1 < L) publish_begin(cur);
#else
        if (m + 1 < L) publish(cur);
#endif
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
        if (lane < 16) xs_part_add(prow + lane, v, first);
      }
      if (DIAGCTL) {
        double dv[XR_DMAX];
        if (tab.dstring) {
          xs_diag_string_sums<RPT>(dct, rowacc, dsg, dv);
        } else {
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
        }
        const double v = xs_wave_sums16(dv, lane);
        if (lane < 16) xs_part_add(prow + RM + lane, v, first);
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

// the quadrature form (tables of up to 16 XOR-diagonals, 4-point rule).  The headline's table class (real, unsigned, no diagonal
// controls) keeps the node basis with both vectors in registers; every other class (parity signs, complex phases, diagonal controls:
// the reference's own control sets -- ZZ-edge + X-vertex QAOA, demo_maxcut.py:71-79, and I/Q drives) runs the difference basis with
// u_2 in LDS (qd.basis = 1, made by xs_quad_make for those tables): its 32 registers are what the sign state / the row accumulators need
template <class T, int NT, int RPT, int SIGN, bool DIAGCTL, bool CPLX, bool U2L>
hipError_t xs_backward_q_start(typename Cx<T>::type* psi, typename Cx<T>::type* lam, const XrTab& tab, const XsOrder& ord, const XsOrder& ordh,
                               const double* wt, const double* dgs, long dgs_step, int S, int B, int L, int nblk, const XsQuad& qd,
                               double* part, double pr, double pi, hipStream_t st) {
  typedef typename Cx<T>::type T2;
  const size_t lds = xs_lds_bytes(NT * RPT, sizeof(T)) + (U2L ? sizeof(float2) * (size_t)(NT * RPT) : 0);   // + the u_2 slots
  void (*kf)(T2*, T2*, XrTab, XsOrder, XsOrder, const double*, const double*, long, int, int, int, XsQuad, double*, T, T) =
      xs_backward_q_kernel<T, NT, RPT, SIGN, DIAGCTL, CPLX, 2, U2L>;
  hipError_t e = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  XsQuad qk = qd;   // the factorials of the division-free recurrence, into the tables (row m / m!)
  {
    double f = 1.0;
    for (int m = 1; m <= XS_LMAX; ++m) {
      f /= (double)m;
      qk.g[m] *= f;
      qk.w1[m] *= f;
      for (int i = 0; i < XS_NIMAX; ++i) {
        qk.c1[m][i] *= f;
        qk.c2[m][i] *= f;
      }
    }
  }
  hipLaunchKernelGGL(kf, dim3((unsigned)nblk), dim3(NT), lds, st, psi, lam, tab, ord, ordh, wt, dgs, dgs_step, S, B, L, qk, part, (T)pr, (T)pi);
  return hipGetLastError();
}
template <class T, int NT, int RPT>
hipError_t xs_backward_q_go(typename Cx<T>::type* psi, typename Cx<T>::type* lam, const XrTab& tab, const double* wt, const double* dgs,
                            long dgs_step, int S, int B, int L, int nblk, const XsQuad& qd, double* part, double pr, double pi, hipStream_t st) {
  const int var = xs_threads(NT * RPT, 0) == NT ? 0 : 1;
  if (!xs_quad_supported(tab, qd.ni, NT * RPT, var, sizeof(T))) return hipErrorInvalidValue;
  const bool sgn = tab.sgn != nullptr, dc = xs_has_diag_controls(tab), cplx = xs_table_is_complex(tab);
  if (cplx && !sgn) return hipErrorInvalidValue;  // (complex tables always carry sign words)
  constexpr bool UL = sizeof(T) == sizeof(double) && RPT == 8;   // the shapes the difference-basis kernels are built for
  if ((qd.basis != 0) != xs_quad_basis(tab, NT * RPT, var, sizeof(T))) return hipErrorInvalidValue;   // (the rule's tables are made for the kernel's basis)
  if (qd.basis != 0 && !UL) return hipErrorInvalidValue;
  const XsOrder ord = xs_make_order(NT * RPT, tab, var, true), ordh = xs_make_order(NT * RPT, tab, var, false);
  const bool rs = xs_order_row_signs(ord) || xs_order_row_signs(ordh);   // SIGN = 2: some diagonal's sign has a row factor
#define XS_QGO(SG, DC, CX, U) xs_backward_q_start<T, NT, RPT, SG, DC, CX, U>(psi, lam, tab, ord, ordh, wt, dgs, dgs_step, S, B, L, nblk, qd, part, pr, pi, st)
#define XS_QSG(DC, CX, U) (rs ? XS_QGO(2, DC, CX, U) : XS_QGO(1, DC, CX, U))
  if (!sgn && !dc && !qd.basis) return XS_QGO(0, false, false, false);
  if constexpr (UL) {   // (signed tables / diagonal controls always take the difference basis on these shapes: xs_quad_basis)
    if (!qd.basis) return hipErrorInvalidValue;
    if (!sgn) return dc ? XS_QGO(0, true, false, true) : XS_QGO(0, false, false, true);
    if (cplx) return dc ? XS_QSG(true, true, true) : XS_QSG(false, true, true);
    return dc ? XS_QSG(true, false, true) : XS_QSG(false, false, true);
  } else {
    if (!sgn) return XS_QGO(0, true, false, false);
    if (cplx) return dc ? XS_QSG(true, true, false) : XS_QSG(false, true, false);
    return dc ? XS_QSG(true, false, false) : XS_QSG(false, false, false);
  }
#undef XS_QSG
#undef XS_QGO
}

template <class T>
hipError_t xs_backward_q_pick(typename Cx<T>::type* psi, typename Cx<T>::type* lam, int d, const XrTab& tab, const double* wt,
                              const double* dgs, long dgs_step, int S, int B, int L, int nblk, const XsQuad& qd, double* part,
                              int variant, double pr, double pi, hipStream_t st) {
  // (no 1024-thread shapes -- d = 8192 in complex64, the wide form of d = 4096: xs_quad_supported keeps them on the exact kernel)
  switch (d) {
    case 4096:
      return (variant & 1) ? hipErrorInvalidValue
                           : xs_backward_q_go<T, 512, 8>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, pr, pi, st);
#ifndef XS_PROBE_4096_ONLY
    case 2048: return xs_backward_q_go<T, 256, 8>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, pr, pi, st);
    case 1024:
      return (variant & 1) ? xs_backward_q_go<T, 256, 4>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, pr, pi, st)
                           : xs_backward_q_go<T, 128, 8>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, pr, pi, st);
    case 512:
      return (variant & 1) ? xs_backward_q_go<T, 256, 2>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, pr, pi, st)
                           : xs_backward_q_go<T, 64, 8>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, pr, pi, st);
    default:
      return (variant & 1) ? xs_backward_q_go<T, 256, 1>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, pr, pi, st)
                           : xs_backward_q_go<T, 64, 4>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, pr, pi, st);
#else
    default: return hipErrorInvalidValue;
#endif
  }
}

}  // namespace

template <class T>
hipError_t xs_backward_quad_launch(typename Cx<T>::type* psi, typename Cx<T>::type* lam, int d, const XrTab& tab, const double* wt,
                                   const double* dgs, long dgs_step, int S, int B, int L, int nblk, const XsQuad& qd, double* part,
                                   int variant, double pr, double pi, hipStream_t st) {
  return xs_backward_q_pick<T>(psi, lam, d, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, variant, pr, pi, st);
}
template hipError_t xs_backward_quad_launch<double>(double2*, double2*, int, const XrTab&, const double*, const double*, long, int, int, int,
                                                    int, const XsQuad&, double*, int, double, double, hipStream_t);
template hipError_t xs_backward_quad_launch<float>(float2*, float2*, int, const XrTab&, const double*, const double*, long, int, int, int, int,
                                                   const XsQuad&, double*, int, double, double, hipStream_t);

}  // namespace dq
