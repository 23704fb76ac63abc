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
#include <cstring>
#include <type_traits>

#include "dq_internal.h"

namespace dq {

namespace {

template <class T>
struct XsV2;
template <>
struct XsV2<double> {
  typedef double type __attribute__((ext_vector_type(2)));
};
template <>
struct XsV2<float> {
  typedef float type __attribute__((ext_vector_type(2)));
};

__device__ inline double xs_flip(double v, unsigned bit) {
  return __longlong_as_double(__double_as_longlong(v) ^ ((long long)bit << 63));
}
__device__ inline float xs_flip(float v, unsigned bit) { return __int_as_float(__float_as_int(v) ^ (int)(bit << 31)); }
__device__ inline double xs_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ inline float xs_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// the per-diagonal accumulators of a thread: one LLVM vector, indexed with a wave-uniform index -- the backend addresses the
// registers indirectly (s_set_gpr_idx / v_movrel); a tree of scalar compares over a plain array costs a dozen register copies per
// diagonal where its paths merge, an unrolled loop over the diagonals drowns the register allocator
typedef double xs_acc16 __attribute__((ext_vector_type(16)));

// dacc[j] += s for a wave-uniform j (register arrays need static indices: a binary tree of scalar compares)
template <int LO, int HI>
__device__ __forceinline__ void xs_acc(double* dacc, int j, double s) {
  if constexpr (HI - LO == 1) {
    asm volatile("" : "+v"(dacc[LO]));
    dacc[LO] += s;
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (j < MID) xs_acc<LO, MID>(dacc, j, s);
    else xs_acc<MID, HI>(dacc, j, s);
  }
}

// One XOR-diagonal's contribution with the partner values in v[]:  o += a_ij v,  a_ij = -i w sigma_ij (PH = 0: entries +-c) or
// w sigma_ij (PH = 1: entries +-i c).  The phase is a template parameter (the host orders every class phase 0 first, so each loop is
// of one phase: a runtime branch cost 16 register copies per diagonal where its two updates merged).  SIGN: the row's sign
// sigma_ij goes into the weight (one xor on its high word) instead of into both components of the partner value.
// DOT: the level's gradient pairing of the partners with the thread's own rows xm, into lane `slot` of the accumulator vector.
template <class T, int RPT, bool SIGN, bool DOT, int PH>
__device__ __forceinline__ void xs_use(int j, int slot, T w, const unsigned* sg, const typename Cx<T>::type* v, typename Cx<T>::type* o,
                                       const typename Cx<T>::type* xm, double* dacc) {
  T wr[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) wr[r] = SIGN ? xs_flip(w, (sg[r] >> j) & 1u) : w;
  if (PH) {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      o[r].x = xs_fma(wr[r], v[r].x, o[r].x);
      o[r].y = xs_fma(wr[r], v[r].y, o[r].y);
    }
  } else {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      o[r].x = xs_fma(wr[r], v[r].y, o[r].x);
      o[r].y = xs_fma(-wr[r], v[r].x, o[r].y);
    }
  }
  if (DOT) {  // (dacc: an xs_acc16; slot = the diagonal's sorted position, wave-uniform)
    T sa = 0, sb = 0;  // two chains: the sums over the rows are latency-bound otherwise
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      T p = PH ? -xs_fma(v[r].x, xm[r].x, v[r].y * xm[r].y) : xs_fma(v[r].x, xm[r].y, -v[r].y * xm[r].x);
      if (SIGN) p = xs_flip(p, (sg[r] >> j) & 1u);
      if (r & 1) sb += p;
      else sa += p;
    }
    (*(xs_acc16*)dacc)[slot] += (double)sa + (double)sb;
  }
}

__device__ __forceinline__ double xs_readlane(double v, int j) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), j), hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
  return __hiloint2double(hi, lo);
}

// The table in the order the sweeps walk it, and this step's weights -- one entry per lane (lane q < R), read back with
// v_readlane: no memory access in the loop over the diagonals.  Order (XsOrder, made on the host): first the n0 diagonals whose
// partners are LDS reads that keep their place in the thread's row list (offset < NT), then the n1 LDS ones that permute the row
// list too, then those whose offset has no lane bits (partners = the thread's own registers), grouped by xh = offset / NT.
struct XsCtl {
  int R, n0, n1, n0p1, n1p1;  // (n0p1 / n1p1 / nlocp1: how many of a class's diagonals, its LAST ones, have phase 1)
  int nloc[8], nlocp1[8];
  int ent;    // xoff | phase << 16 | (index in the table: sign bit, accumulator) << 20
  double w;   // dt u c_j of the current step
  int perm;   // this lane's index in the table
  __device__ __forceinline__ void load_table(const XrTab& tab, const XsOrder& ord, int lane) {
    R = tab.R;
    n0 = ord.n0;
    n1 = ord.n1;
    n0p1 = ord.n0p1;
    n1p1 = ord.n1p1;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      nloc[i] = ord.nloc[i];
      nlocp1[i] = ord.nlocp1[i];
    }
    perm = lane < tab.R ? ord.perm[lane] : 0;
    ent = lane < tab.R ? (tab.xoff[perm] | (tab.phase[perm] << 16) | (perm << 20)) : 0;
  }
  __device__ __forceinline__ void load_step(const double* __restrict__ wrow, int lane) { w = lane < R ? wrow[perm] : 0.0; }
};

// partner values by LDS reads (volatile: neighbours are not to be merged into ds_read2 forms, which the LDS array serves at half
// the rate).  PERM: the offset permutes the thread's row list as well (xh != 0)
template <class T, int NT, int RPT, bool PERM>
__device__ __forceinline__ void xs_fetch(int xo, const typename Cx<T>::type* __restrict__ src, int t, typename Cx<T>::type* v) {
  typedef typename XsV2<T>::type V2;
  typedef const volatile __attribute__((address_space(3))) V2* lds_ptr;
  lds_ptr base = (lds_ptr)(src + (t ^ (xo & (NT - 1))));
  const int xh = xo / NT;
#pragma unroll
  for (int r = 0; r < RPT; ++r) {
    const V2 q = PERM ? base[(r ^ xh) * NT] : base[r * NT];
    v[r].x = q.x;
    v[r].y = q.y;
  }
}

// Ring form of a class's loop (complex128, 8 rows per thread): the partner buffer v[] is a ring over the rows -- as soon as row r
// of diagonal q has been used, its slot takes the read of row r of diagonal q + 1, so 8 reads stay in flight all the time, in the
// SAME 32 registers (the plain loop drains to zero at the end of every diagonal and pays the LDS latency again at the start of the
// next; a second buffer does not fit the adjoint kernels' register budget).  hipcc will not keep such a schedule by itself (it
// hoists the next diagonal's reads into a second buffer or sinks the uses behind them), so the reads and the counted waits are
// inline asm: a wait is tied to its slot (in/out operand: the uses follow it), a read overwrites its slot (in/out: the uses
// precede it), and volatile asm statements keep their order -- everything else (the arithmetic) is plain C++ between the two.
// LDS returns in order, so before row r is used all but the 7 youngest reads must be done: s_waitcnt lgkmcnt(7) every time.
// Measured at d = B = 4096 (tools/statecol_probe.hip, profiles/r4_probe_ring*.txt): the pairing sweeps (4 FMAs per partner value)
// gain 25 % from the ring, best at 4 - 6 reads in flight (8: the queue in front of the LDS holds up the other wave's issue); the
// plain sweeps (2 FMAs per partner value: LDS-bound) lose 10 - 30 % at every depth against the compiler's burst loop -> off there.
#ifndef XS_RING_PLAIN
#define XS_RING_PLAIN 0
#endif
#ifndef XS_RING_PAIR
#define XS_RING_PAIR 1
#endif
#ifndef XS_PAIR_LOCAL
#define XS_PAIR_LOCAL 1   // quadrature form: the pairing sweeps take the partners at offsets without lane bits from registers
#endif
#ifndef XS_RING_DEPTH_PLAIN
#define XS_RING_DEPTH_PLAIN 8
#endif
#ifndef XS_RING_DEPTH_PAIR
#define XS_RING_DEPTH_PAIR 4
#endif
typedef double xs_d2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned xs_lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
template <int OFF>
__device__ __forceinline__ void xs_ds_read_first(xs_d2& q, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void xs_ds_read_next(xs_d2& q, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(q) : "v"(addr), "n"(OFF));
}
template <int N>
__device__ __forceinline__ void xs_ds_wait(xs_d2& q) {
  asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(q) : "n"(N));
}
template <int R0, int R1, class F>
__device__ __forceinline__ void xs_rows(F&& f) {
  if constexpr (R0 < R1) {
    f(std::integral_constant<int, R0>{});
    xs_rows<R0 + 1, R1>(f);
  }
}
// one row's share of a diagonal: o += a_ij v and, with DOT, the pairing sums in four chains s[0..3]
template <bool SIGN, bool DOT, int PH, int R>
__device__ __forceinline__ void xs_ring_use(double w, unsigned sbit, xs_d2 v, double2& o, const double2* xm, double* s) {
  if (SIGN && DOT) {  // the row's sign goes into the partner value (both uses take it from there)
    v.x = xs_flip(v.x, sbit);
    v.y = xs_flip(v.y, sbit);
  } else if (SIGN) {
    w = xs_flip(w, sbit);
  }
  if (PH) {
    o.x = __builtin_fma(w, v.x, o.x);
    o.y = __builtin_fma(w, v.y, o.y);
  } else {
    o.x = __builtin_fma(w, v.y, o.x);
    o.y = __builtin_fma(-w, v.x, o.y);
  }
  if (DOT) {
    constexpr int c = (R & 1) * 2;
    if (PH) {
      s[c] = __builtin_fma(v.x, xm[R].x, s[c]);
      s[c + 1] = __builtin_fma(v.y, xm[R].y, s[c + 1]);
    } else {
      s[c] = __builtin_fma(v.x, xm[R].y, s[c]);
      s[c + 1] = __builtin_fma(v.y, xm[R].x, s[c + 1]);
    }
  }
}
// DEPTH = reads in flight (the read of element k + DEPTH of the class's (diagonal, row) sequence is issued right after element k
// has been used; DEPTH = RPT: a slot is refilled with the same row of the next diagonal)
template <int NT, int RPT, bool SIGN, bool DOT, bool PERM, int PH, int SB, int DEPTH>
__device__ __forceinline__ void xs_lds_ring(const XsCtl& ctl, int q0, int n, const unsigned* sg, const double2* __restrict__ src, int t,
                                            double sgn, double2* o, const double2* xm, double* dacc) {
  static_assert(DEPTH >= 1 && DEPTH <= RPT, "ring depth");
  if (n <= 0) return;
  xs_d2 v[RPT];
  const unsigned base = xs_lds_addr(src);
  int e = __builtin_amdgcn_readlane(ctl.ent, q0);
  unsigned a = base + (unsigned)((t ^ (e & (NT - 1))) << 4);
  int xh = (e & 0xffff) / NT;
  xs_rows<0, DEPTH>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if constexpr (PERM) xs_ds_read_first<0>(v[r], a + (unsigned)(((r ^ xh) * NT) << 4));
    else xs_ds_read_first<r * NT * 16>(v[r], a);
  });
  xs_rows<DEPTH, RPT>([&](auto rc) { v[decltype(rc)::value] = 0.0; });  // (defined before their first in/out use)
  const int qe = q0 + n;
  auto finish = [&](int q, const double* s) {
    if (DOT) {
      const double tot = PH ? -((s[0] + s[1]) + (s[2] + s[3])) : (s[0] - s[1]) + (s[2] - s[3]);
      (*(xs_acc16*)dacc)[q - SB] += tot;
    }
  };
#pragma clang loop unroll(disable)
  for (int q = q0; q + 1 < qe; ++q) {
    const int en = __builtin_amdgcn_readlane(ctl.ent, q + 1);
    const unsigned an = base + (unsigned)((t ^ (en & (NT - 1))) << 4);
    const int xhn = (en & 0xffff) / NT;
    const double w = sgn * xs_readlane(ctl.w, q);
    const int j = e >> 20;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    xs_rows<0, RPT>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      xs_ds_wait<DEPTH - 1>(v[r]);
      xs_ring_use<SIGN, DOT, PH, r>(w, SIGN ? (sg[r] >> j) & 1u : 0u, v[r], o[r], xm, s);
      constexpr int k = r + DEPTH;  // the element whose read goes out now
      if constexpr (k < RPT) {
        if constexpr (PERM) xs_ds_read_next<0>(v[k], a + (unsigned)(((k ^ xh) * NT) << 4));
        else xs_ds_read_next<k * NT * 16>(v[k], a);
      } else {
        if constexpr (PERM) xs_ds_read_next<0>(v[k - RPT], an + (unsigned)((((k - RPT) ^ xhn) * NT) << 4));
        else xs_ds_read_next<(k - RPT) * NT * 16>(v[k - RPT], an);
      }
    });
    finish(q, s);
    e = en;
    a = an;
    xh = xhn;
  }
  {
    const double w = sgn * xs_readlane(ctl.w, qe - 1);
    const int j = e >> 20;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    xs_rows<0, RPT>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      xs_ds_wait<(RPT - 1 - r < DEPTH - 1 ? RPT - 1 - r : DEPTH - 1)>(v[r]);
      xs_ring_use<SIGN, DOT, PH, r>(w, SIGN ? (sg[r] >> j) & 1u : 0u, v[r], o[r], xm, s);
      constexpr int k = r + DEPTH;
      if constexpr (k < RPT) {
        if constexpr (PERM) xs_ds_read_next<0>(v[k], a + (unsigned)(((k ^ xh) * NT) << 4));
        else xs_ds_read_next<k * NT * 16>(v[k], a);
      }
    });
    finish(qe - 1, s);
  }
}

// the n diagonals q0 .. q0 + n - 1 of one LDS class; PREF: the next diagonal's partners are fetched while this one's are used
template <class T, int NT, int RPT, bool SIGN, bool DOT, int RM, bool PERM, bool PREF, int PH, int SB = 0>
__device__ __forceinline__ void xs_lds_class(const XsCtl& ctl, int q0, int n, const unsigned* sg,
                                             const typename Cx<T>::type* __restrict__ src, int t, T sgn, typename Cx<T>::type* o,
                                             const typename Cx<T>::type* xm, double* dacc) {  // (SB: accumulator lane = position - SB)
  typedef typename Cx<T>::type T2;
  if (n <= 0) return;
  if constexpr (!PREF && RPT == 8 && sizeof(T) == sizeof(double) && ((DOT && XS_RING_PAIR) || (!DOT && XS_RING_PLAIN))) {
    xs_lds_ring<NT, RPT, SIGN, DOT, PERM, PH, SB, DOT ? XS_RING_DEPTH_PAIR : XS_RING_DEPTH_PLAIN>(ctl, q0, n, sg, src, t, sgn, o, xm, dacc);
    return;
  }
  if (PREF) {
    T2 va[RPT], vb[RPT];
    int e0 = __builtin_amdgcn_readlane(ctl.ent, q0), e1 = 0;
    xs_fetch<T, NT, RPT, PERM>(e0 & 0xffff, src, t, va);
    int q = q0;
    const int qe = q0 + n;
#pragma clang loop unroll(disable)
    for (; q + 1 < qe; q += 2) {
      e1 = __builtin_amdgcn_readlane(ctl.ent, q + 1);
      xs_fetch<T, NT, RPT, PERM>(e1 & 0xffff, src, t, vb);
      xs_use<T, RPT, SIGN, DOT, PH>(e0 >> 20, q - SB, sgn * (T)xs_readlane(ctl.w, q), sg, va, o, xm, dacc);
      if (q + 2 < qe) {
        e0 = __builtin_amdgcn_readlane(ctl.ent, q + 2);
        xs_fetch<T, NT, RPT, PERM>(e0 & 0xffff, src, t, va);
      }
      xs_use<T, RPT, SIGN, DOT, PH>(e1 >> 20, q + 1 - SB, sgn * (T)xs_readlane(ctl.w, q + 1), sg, vb, o, xm, dacc);
    }
    if (q < qe) xs_use<T, RPT, SIGN, DOT, PH>(e0 >> 20, q - SB, sgn * (T)xs_readlane(ctl.w, q), sg, va, o, xm, dacc);
  } else {
#pragma clang loop unroll(disable)
    for (int q = q0; q < q0 + n; ++q) {
      const int e = __builtin_amdgcn_readlane(ctl.ent, q);
      T2 v[RPT];
      xs_fetch<T, NT, RPT, PERM>(e & 0xffff, src, t, v);
      xs_use<T, RPT, SIGN, DOT, PH>(e >> 20, q - SB, sgn * (T)xs_readlane(ctl.w, q), sg, v, o, xm, dacc);
    }
  }
}

// a class of a pairing sweep with up to 32 diagonals: positions below 16 accumulate into the first accumulator vector, the others
// into the second -- two instantiations of the loop, no per-diagonal choice of the vector
template <class T, int NT, int RPT, bool SIGN, int RM, bool PERM, int PH>
__device__ __forceinline__ void xs_pair_class(const XsCtl& ctl, int q0, int n, const unsigned* sg,
                                              const typename Cx<T>::type* __restrict__ src, int t, T sgn, typename Cx<T>::type* o,
                                              const typename Cx<T>::type* xm, xs_acc16* dacc) {
  const int qe = q0 + n;
  const int lo_end = qe < 16 ? qe : 16;
  xs_lds_class<T, NT, RPT, SIGN, true, RM, PERM, false, PH, 0>(ctl, q0, lo_end - q0, sg, src, t, sgn, o, xm, (double*)&dacc[0]);
  if constexpr (RM > 16) {
    const int hi0 = q0 > 16 ? q0 : 16;
    xs_lds_class<T, NT, RPT, SIGN, true, RM, PERM, false, PH, 16>(ctl, hi0, qe - hi0, sg, src, t, sgn, o, xm, (double*)&dacc[1]);
  }
}

// the pairing sweep (Horner on lambda): o += sgn (A - diag) v with every partner from LDS, in the order of ctl (no register class)
template <class T, int NT, int RPT, bool SIGN, int RM, bool CPLX>
__device__ __forceinline__ void xs_gather_pairings(const XsCtl& ctl, const unsigned* sg, const typename Cx<T>::type* __restrict__ src,
                                                   int t, T sgn, typename Cx<T>::type* o, const typename Cx<T>::type* xm,
                                                   xs_acc16* dacc) {
  xs_pair_class<T, NT, RPT, SIGN, RM, false, 0>(ctl, 0, ctl.n0 - ctl.n0p1, sg, src, t, sgn, o, xm, dacc);
  if constexpr (CPLX) xs_pair_class<T, NT, RPT, SIGN, RM, false, 1>(ctl, ctl.n0 - ctl.n0p1, ctl.n0p1, sg, src, t, sgn, o, xm, dacc);
  xs_pair_class<T, NT, RPT, SIGN, RM, true, 0>(ctl, ctl.n0, ctl.n1 - ctl.n1p1, sg, src, t, sgn, o, xm, dacc);
  if constexpr (CPLX) xs_pair_class<T, NT, RPT, SIGN, RM, true, 1>(ctl, ctl.n0 + ctl.n1 - ctl.n1p1, ctl.n1p1, sg, src, t, sgn, o, xm, dacc);
}

// the diagonals whose partners are the thread's own rows r ^ XH (register copies), class by class, phase 0 then phase 1
template <class T, int RPT, int XH, bool SIGN, bool DOT, int PH>
__device__ __forceinline__ int xs_local_run(const XsCtl& ctl, int q, int n, const unsigned* sg, const typename Cx<T>::type* cur, T sgn,
                                            typename Cx<T>::type* o, const typename Cx<T>::type* xm, double* dacc) {
#pragma clang loop unroll(disable)
  for (int i = 0; i < n; ++i, ++q) {
    const int e = __builtin_amdgcn_readlane(ctl.ent, q);
    typename Cx<T>::type v[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) v[r] = cur[r ^ XH];
    xs_use<T, RPT, SIGN, DOT, PH>(e >> 20, q, sgn * (T)xs_readlane(ctl.w, q), sg, v, o, xm, dacc);
  }
  return q;
}
template <class T, int RPT, int XH, bool SIGN, bool DOT, bool CPLX>
__device__ __forceinline__ void xs_local_classes(const XsCtl& ctl, int q, const unsigned* sg, const typename Cx<T>::type* cur, T sgn,
                                                 typename Cx<T>::type* o, const typename Cx<T>::type* xm, double* dacc) {
  if constexpr (XH < RPT && XH < 8) {
    q = xs_local_run<T, RPT, XH, SIGN, DOT, 0>(ctl, q, ctl.nloc[XH] - ctl.nlocp1[XH], sg, cur, sgn, o, xm, dacc);
    if constexpr (CPLX) q = xs_local_run<T, RPT, XH, SIGN, DOT, 1>(ctl, q, ctl.nlocp1[XH], sg, cur, sgn, o, xm, dacc);
    xs_local_classes<T, RPT, XH + 1, SIGN, DOT, CPLX>(ctl, q, sg, cur, sgn, o, xm, dacc);
  }
}

// o += sgn * (A - diag) v,  v = cur (the thread's rows, also in the LDS image `src`), class by class and phase by phase: no
// per-diagonal branching.  LOCAL = false: the caller's order (xs_make_order with local = false) has no register class -- every
// partner is read from LDS and the caller's copy of v need not stay in registers.  DOT: the sweep carries the gradient pairings; dacc is then an xs_acc16 indexed by the diagonal's SORTED
// position (xs_finish_kernel maps back).
// CPLX = false (real tables: every phase 0): the phase-1 loops are not even instantiated (their code alone cost the real kernels 4 %).
template <class T, int NT, int RPT, bool SIGN, bool DOT, int RM, bool PREF, bool LOCAL, bool CPLX>
__device__ __forceinline__ void xs_gather(const XsCtl& ctl, const unsigned* sg, const typename Cx<T>::type* __restrict__ src,
                                          const typename Cx<T>::type* cur, int t, T sgn, typename Cx<T>::type* o,
                                          const typename Cx<T>::type* xm, double* dacc) {
  xs_lds_class<T, NT, RPT, SIGN, DOT, RM, false, PREF, 0>(ctl, 0, ctl.n0 - ctl.n0p1, sg, src, t, sgn, o, xm, dacc);
  if constexpr (CPLX) xs_lds_class<T, NT, RPT, SIGN, DOT, RM, false, PREF, 1>(ctl, ctl.n0 - ctl.n0p1, ctl.n0p1, sg, src, t, sgn, o, xm, dacc);
  xs_lds_class<T, NT, RPT, SIGN, DOT, RM, true, PREF, 0>(ctl, ctl.n0, ctl.n1 - ctl.n1p1, sg, src, t, sgn, o, xm, dacc);
  if constexpr (CPLX) xs_lds_class<T, NT, RPT, SIGN, DOT, RM, true, PREF, 1>(ctl, ctl.n0 + ctl.n1 - ctl.n1p1, ctl.n1p1, sg, src, t, sgn, o, xm, dacc);
  if (LOCAL) xs_local_classes<T, RPT, 1, SIGN, DOT, CPLX>(ctl, ctl.n0 + ctl.n1, sg, cur, sgn, o, xm, dacc);
}

template <class T, int NT, int RPT>
__device__ __forceinline__ void xs_put(typename Cx<T>::type* img, int t, const typename Cx<T>::type* v) {
#pragma unroll
  for (int r = 0; r < RPT; ++r) img[r * NT + t] = v[r];
}

}  // namespace

// wt[s][j] = dt u[s][kid_j] cmag_j;  dgs: the rows' diagonal weights dt (H0 + sum_k u_k H_k)[i][i] -- one vector [d] when only
// the drift has diagonal entries (per_step = 0), else one per step [S][d].  grid: (ceil(d / 256), per_step ? S : 1) + S blocks for wt
__global__ void __launch_bounds__(256) xs_weights_kernel(XrTab tab, const double* __restrict__ u, int K, int S, int d, double dt,
                                                         int per_step, double* __restrict__ wt, double* __restrict__ dgs) {
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
      dgs[(long)s * d + i] = dt * v;
    }
  } else {
    const int s = b - nbd * nds;
    const double* ur = u + (long)s * K;
    if (t < tab.R) wt[(long)s * tab.R + t] = dt * (tab.kid[t] < 0 ? 1.0 : ur[tab.kid[t]]) * tab.cmag[t];
  }
}

// psi_out[b][:] = U_{S-1} ... U_0 psi_in[b][:] for the columns b = blockIdx.x, + gridDim.x, ...  (states stored column by
// column: [B][D]).  dgs + s * dgs_step: the rows' diagonal weights of step s.
template <class T, int NT, int RPT, bool SIGN, bool PREF, bool CPLX>
__global__ void __launch_bounds__(NT) xs_forward_kernel(const typename Cx<T>::type* __restrict__ psi_in,
                                                        typename Cx<T>::type* __restrict__ psi_out, XrTab tab, XsOrder ord,
                                                        const double* __restrict__ wt, const double* __restrict__ dgs, long dgs_step,
                                                        int S, int B, int L) {
  typedef typename Cx<T>::type T2;
  constexpr int D = NT * RPT;
  extern __shared__ __align__(16) unsigned char xs_lds_raw[];
  T2* img = (T2*)xs_lds_raw;  // two images of D
  const int t = threadIdx.x;
  unsigned sg[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) sg[r] = SIGN ? tab.sgn[r * NT + t] : 0u;
  XsCtl ctl;
  ctl.load_table(tab, ord, t & 63);
  int p = 0;
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    T2 cur[RPT];
    const T2* in = psi_in + (long)b * D;
#pragma unroll
    for (int r = 0; r < RPT; ++r) cur[r] = in[r * NT + t];
    p ^= 1;
    xs_put<T, NT, RPT>(img + p * D, t, cur);
    __syncthreads();
    for (int s = 0; s < S; ++s) {
      ctl.load_step(wt + (long)s * tab.R, t & 63);
      const double* dgr = dgs + (long)s * dgs_step;
      T2 acc[RPT];
      T dg[RPT];  // the rows' diagonal weights of this step (registers: a load per sweep would sit in the vmcnt queue of every wait)
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        acc[r] = cur[r];
        dg[r] = (T)dgr[r * NT + t];
      }
      for (int m = 1; m <= L; ++m) {
        T2 o[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          o[r].x = 0;
          o[r].y = 0;
        }
        xs_gather<T, NT, RPT, SIGN, false, 1, PREF, true, CPLX>(ctl, sg, img + p * D, cur, t, (T)1, o, nullptr, nullptr);
        const T inv = (T)1 / (T)m;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {  // + the diagonal: -i dg v
          const T ox = xs_fma(dg[r], cur[r].y, o[r].x) * inv;
          const T oy = xs_fma(-dg[r], cur[r].x, o[r].y) * inv;
          cur[r].x = ox;
          cur[r].y = oy;
          acc[r].x += ox;
          acc[r].y += oy;
        }
        if (m < L) {
          p ^= 1;
          xs_put<T, NT, RPT>(img + p * D, t, cur);
          __syncthreads();
        }
      }
#pragma unroll
      for (int r = 0; r < RPT; ++r) cur[r] = acc[r];
      if (s + 1 < S) {
        p ^= 1;
        xs_put<T, NT, RPT>(img + p * D, t, cur);
        __syncthreads();
      }
    }
    T2* out = psi_out + (long)b * D;
#pragma unroll
    for (int r = 0; r < RPT; ++r) out[r * NT + t] = cur[r];
    __syncthreads();  // (every read of this column's images is done before the next column overwrites one)
  }
}

// 16 per-lane values -> lane l < 16 holds the wave's total of value l (17 exchanges instead of 96).  Every stage has its own
// constant-indexed array, so that the values stay in registers.
__device__ __forceinline__ double xs_wave_sums16(const double* a, int lane) {
  double b8[8], b4[4], b2[2];
  {
    const bool up = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) b8[i] = (up ? a[i + 8] : a[i]) + __shfl_xor(up ? a[i] : a[i + 8], 8, 64);
  }
  {
    const bool up = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) b4[i] = (up ? b8[i + 4] : b8[i]) + __shfl_xor(up ? b8[i] : b8[i + 4], 4, 64);
  }
  {
    const bool up = (lane & 2) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) b2[i] = (up ? b4[i + 2] : b4[i]) + __shfl_xor(up ? b4[i] : b4[i + 2], 2, 64);
  }
  const bool up = (lane & 1) != 0;
  double v = (up ? b2[1] : b2[0]) + __shfl_xor(up ? b2[0] : b2[1], 1, 64);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;  // lane l: the total of value (l & 15)
}

#ifdef XS_STAMPS
// development builds (tools/statecol_probe.hip): shader cycles per phase of the backward kernel, summed over workgroups
__device__ unsigned long long xs_stamps[8];
#define XS_STAMP(i)                                    \
  do {                                                 \
    const long long now__ = clock64();                 \
    stamp_acc[i] += now__ - stamp_t;                   \
    stamp_t = now__;                                   \
  } while (0)
#else
#define XS_STAMP(i) do { } while (0)
#endif

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
template <class T, int NT, int RPT, bool SIGN, bool DIAGCTL, int RM, bool PREF, bool CPLX>
__global__ void __launch_bounds__(NT) xs_backward_kernel(typename Cx<T>::type* __restrict__ psi, typename Cx<T>::type* __restrict__ lam,
                                                         XrTab tab, XsOrder ord, XsOrder ordh, const double* __restrict__ wt,
                                                         const double* __restrict__ dgs, long dgs_step, int S, int B, int L,
                                                         typename Cx<T>::type* __restrict__ park,
                                                         double* __restrict__ part) {
  typedef typename Cx<T>::type T2;
  constexpr int D = NT * RPT;
  constexpr int NW = NT / 64;
  constexpr int NS = RM + (DIAGCTL ? XR_DMAX : 0);
  extern __shared__ __align__(16) unsigned char xs_lds_raw[];
  T2* W = (T2*)xs_lds_raw;
  T2* Lm = W + D;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  unsigned sg[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) sg[r] = SIGN ? tab.sgn[r * NT + t] : 0u;
  XsCtl ctl, ctlh;  // (ctlh: the order of the pairing sweeps -- every partner from LDS, phase 0 first)
  ctl.load_table(tab, ord, lane);
  ctlh.load_table(tab, ordh, lane);
  T2* pk = park + (long)blockIdx.x * L * D;  // [l][r * NT + t] (uniform base, lane index t)
  // in-place update of the work image: every read of the sweep that just ran is done before any row is overwritten
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
    T2 cur[RPT];
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
      {  // (scope of dg: the reversal and the forward terms keep the rows' diagonal weights in registers)
      T dg[RPT];
#pragma unroll
      for (int r = 0; r < RPT; ++r) dg[r] = (T)dgr[r * NT + t];
      // ---- reversal: psi_s = sum_m (-A)^m psi_{s+1} / m! -------------------------------------------------------------
      publish(cur);
      {
        T2 acc[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) acc[r] = cur[r];
        for (int m = 1; m <= L; ++m) {
          T2 o[RPT];
#pragma unroll
          for (int r = 0; r < RPT; ++r) {
            o[r].x = 0;
            o[r].y = 0;
          }
          xs_gather<T, NT, RPT, SIGN, false, 1, PREF, true, CPLX>(ctl, sg, W, cur, t, (T)-1, o, nullptr, nullptr);
          const T inv = (T)1 / (T)m;
#pragma unroll
          for (int r = 0; r < RPT; ++r) {  // + i dg v
            const T ox = xs_fma(-dg[r], cur[r].y, o[r].x) * inv;
            const T oy = xs_fma(dg[r], cur[r].x, o[r].y) * inv;
            cur[r].x = ox;
            cur[r].y = oy;
            acc[r].x += ox;
            acc[r].y += oy;
          }
          if (m < L) publish(cur);
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
      if (L > 1) publish(cur);
      for (int l = 1; l < L; ++l) {
        T2 o[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          o[r].x = 0;
          o[r].y = 0;
        }
        xs_gather<T, NT, RPT, SIGN, false, 1, PREF, true, CPLX>(ctl, sg, W, cur, t, (T)1, o, nullptr, nullptr);
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
        if (l < L - 1) publish(cur);
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
        const T inv = (T)1 / (T)L;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          xm[r] = cur[r];
          const T2 g = Lm[r * NT + t];
          cur[r].x = g.x * inv;
          cur[r].y = g.y * inv;
        }
      }
      publish(cur);
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
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          const T dgv = (T)dgr[r * NT + t];  // (L2-resident table; not kept in registers across the sweep)
          const T2 g = Lm[r * NT + t];
          o[r].x = xs_fma(-dgv, cur[r].y, g.x);  // lam - (-i dg y)
          o[r].y = xs_fma(dgv, cur[r].x, g.y);
          if (DIAGCTL) rowacc[r] += (double)cur[r].x * (double)xm[r].y - (double)cur[r].y * (double)xm[r].x;
        }
        xs_gather_pairings<T, NT, RPT, SIGN, RM, CPLX>(ctlh, sg, W, t, (T)-1, o, xm, dacc);
        if (l > 0) {
          const T inv = (T)1 / (T)l;
#pragma unroll
          for (int r = 0; r < RPT; ++r) {
            cur[r].x = o[r].x * inv;
            cur[r].y = o[r].y * inv;
          }
          publish(cur);
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
        if (lane < 16) prow[g * 16 + lane] = first ? v : prow[g * 16 + lane] + v;
      }
      if (DIAGCTL) {
        double dv[XR_DMAX];
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
        const double v = xs_wave_sums16(dv, lane);
        if (lane < 16) prow[RM + lane] = first ? v : prow[RM + lane] + v;
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
// <q, dA r> = -<r, dA q>, and the pairing code takes the fetched partners as its first factor), qd.w0 = -w_0, qd.w1 = -w_1.
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
  unsigned sg[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) sg[r] = SIGN ? tab.sgn[r * NT + t] : 0u;
  XsCtl ctl, ctlh;
  ctl.load_table(tab, ord, lane);
  ctlh.load_table(tab, ordh, lane);
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
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          zc[r] = Lm[r * NT + t];
          acc[r] = zc[r];
#pragma unroll
          for (int i = 0; i < NI; ++i) nd[i][r] = zc[r];
        }
        publish(zc);
        for (int m = 1; m <= L; ++m) {
          T2 o[RPT];
#pragma unroll
          for (int r = 0; r < RPT; ++r) {
            o[r].x = 0;
            o[r].y = 0;
          }
          xs_gather<T, NT, RPT, SIGN, false, 1, false, true, CPLX>(ctl, sg, W, zc, t, (T)-1, o, nullptr, nullptr);
          const T inv = (T)1 / (T)m;
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
            acc[r].x += ox;
            acc[r].y += oy;
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
        const T w0 = (T)qd.w0, w1 = (T)qd.w1;
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
#pragma unroll
      for (int r = 0; r < RPT; ++r) acc[r] = cur[r];
      publish(cur);
      for (int m = 0; m < L; ++m) {
        if (m > 0) {
          T cw[NI];
#pragma unroll
          for (int i = 0; i < NI; ++i) cw[i] = (T)qd.c2[m][i];
          const T w1 = (T)qd.w1;
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
        xs_lds_class<T, NT, RPT, SIGN, true, RM, false, false, 0>(ctl, 0, ctl.n0 - ctl.n0p1, sg, W, t, (T)-1, o, q, (double*)&dacc[0]);
        if constexpr (CPLX)
          xs_lds_class<T, NT, RPT, SIGN, true, RM, false, false, 1>(ctl, ctl.n0 - ctl.n0p1, ctl.n0p1, sg, W, t, (T)-1, o, q, (double*)&dacc[0]);
        xs_lds_class<T, NT, RPT, SIGN, true, RM, true, false, 0>(ctl, ctl.n0, ctl.n1 - ctl.n1p1, sg, W, t, (T)-1, o, q, (double*)&dacc[0]);
        if constexpr (CPLX)
          xs_lds_class<T, NT, RPT, SIGN, true, RM, true, false, 1>(ctl, ctl.n0 + ctl.n1 - ctl.n1p1, ctl.n1p1, sg, W, t, (T)-1, o, q, (double*)&dacc[0]);
#else
        xs_gather_pairings<T, NT, RPT, SIGN, RM, CPLX>(ctlh, sg, W, t, (T)-1, o, q, dacc);
#endif
        const T inv = (T)1 / (T)(m + 1);
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
          cur[r].x = o[r].x * inv;
          cur[r].y = o[r].y * inv;
          acc[r].x += cur[r].x;
          acc[r].y += cur[r].y;
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

// grad[s][k] (+)= dt * (sum_j [kid_j == k] cmag_j sum part[.., j] + sum_jd [dkid_jd == k] sum part[.., RM + jd]); one workgroup
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
        if (tab.dkid[jd] == k) g += dt * s_j[rm + jd];
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
static bool xs_has_diag_controls(const XrTab& tab) {
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
  int at = 0;
  for (int ph = 0; ph < 2; ++ph)
    for (int j = 0; j < tab.R; ++j)
      if ((tab.xoff[j] & (nt - 1)) != 0 && tab.xoff[j] < nt && (tab.phase[j] != 0) == (ph != 0)) {
        o.perm[at++] = (unsigned char)j;
        ++o.n0;
        o.n0p1 += ph;
      }
  for (int ph = 0; ph < 2; ++ph)
    for (int j = 0; j < tab.R; ++j)
      if (((tab.xoff[j] & (nt - 1)) != 0 ? tab.xoff[j] >= nt : !local) && (tab.phase[j] != 0) == (ph != 0)) {
        o.perm[at++] = (unsigned char)j;
        ++o.n1;
        o.n1p1 += ph;
      }
  for (int xh = 1; xh < 8 && local; ++xh)
    for (int ph = 0; ph < 2; ++ph)
      for (int j = 0; j < tab.R; ++j)
        if ((tab.xoff[j] & (nt - 1)) == 0 && tab.xoff[j] / nt == xh && (tab.phase[j] != 0) == (ph != 0)) {
          o.perm[at++] = (unsigned char)j;
          ++o.nloc[xh];
          o.nlocp1[xh] += ph;
        }
  return o;
}
size_t xs_park_elems(int d, int L, int nblk) { return (size_t)nblk * (size_t)L * (size_t)d; }
size_t xs_part_doubles(int d, int S, int nblk, const XrTab& tab, int variant) {
  return (size_t)nblk * (size_t)S * (size_t)xs_waves(d, variant) * (size_t)xs_slots(tab);
}

// dgs: d doubles, or S * d when a control has diagonal entries (xs_dgs_doubles); returns the step stride through *dgs_step
size_t xs_dgs_doubles(int d, int S, const XrTab& tab) { return (size_t)d * (size_t)(xs_has_diag_controls(tab) ? (S > 0 ? S : 1) : 1); }

hipError_t xs_weights_launch(const XrTab& tab, const double* u, int K, int S, int d, double dt, double* wt, double* dgs, long* dgs_step,
                             hipStream_t st) {
  const int per_step = xs_has_diag_controls(tab) ? 1 : 0;
  *dgs_step = per_step ? d : 0;
  if (S <= 0) return hipSuccess;
  const int nbd = (d + 255) / 256;
  hipLaunchKernelGGL(xs_weights_kernel, dim3((unsigned)(nbd * (per_step ? S : 1) + S)), dim3(256), 0, st, tab, u, K, S, d, dt, per_step,
                     wt, dgs);
  return hipGetLastError();
}

// Gauss-Lobatto rules on [0, 1] with 2 and 3 interior nodes (4 / 5 points: exact to degree 5 / 7).  The rule's error for an
// integrand with |f^(p)| <= F, p = 2 ni + 2, is e F with e = |int s^p - Q[s^p]| / p! (its Peano kernel keeps one sign).
bool xs_quad_make(double theta, double tol, int ni_max, XsQuad* qd) {
  memset(qd, 0, sizeof(*qd));
  if (!(theta >= 0.0)) return false;
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
    qd->w1 = -we;
    for (int m = 0; m <= XS_LMAX; ++m)
      for (int i = 0; i < ni; ++i) {
        qd->c1[m][i] = std::pow(s[i], m);
        qd->c2[m][i] = -w[i] * qd->c1[m][i];
      }
    return true;
  }
  return false;
}
// (instantiated for tables of up to 16 XOR-diagonals -- one accumulator vector -- and the 4-point rule)
bool xs_quad_supported(const XrTab& tab, int ni) { return ni == 2 && tab.R <= XR_RSMALL; }

namespace {

bool xs_diagctl(const XrTab& tab) { return xs_has_diag_controls(tab); }

template <class T, int NT, int RPT>
hipError_t xs_forward_go(const typename Cx<T>::type* in, typename Cx<T>::type* out, const XrTab& tab, const double* wt, const double* dgs,
                         long dgs_step, int S, int B, int L, int nblk, bool pref, hipStream_t st) {
  typedef typename Cx<T>::type T2;
  const size_t lds = xs_lds_bytes(NT * RPT, sizeof(T));
  const bool sgn = tab.sgn != nullptr;
  bool cplx = false;
  for (int j = 0; j < tab.R; ++j) cplx = cplx || tab.phase[j] != 0;
  if (cplx && !sgn) return hipErrorInvalidValue;  // (complex tables always carry sign words)
  void (*kf)(const T2*, T2*, XrTab, XsOrder, const double*, const double*, long, int, int, int) =
      cplx ? (pref ? xs_forward_kernel<T, NT, RPT, true, true, true> : xs_forward_kernel<T, NT, RPT, true, false, true>)
           : (sgn ? (pref ? xs_forward_kernel<T, NT, RPT, true, true, false> : xs_forward_kernel<T, NT, RPT, true, false, false>)
                  : (pref ? xs_forward_kernel<T, NT, RPT, false, true, false> : xs_forward_kernel<T, NT, RPT, false, false, false>));
  hipError_t e = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kf, dim3((unsigned)nblk), dim3(NT), lds, st, in, out, tab, xs_make_order(NT * RPT, tab, xs_threads(NT * RPT, 0) == NT ? 0 : 1, true), wt, dgs,
                     dgs_step, S, B, L);
  return hipGetLastError();
}

template <class T, int NT, int RPT, bool PREF, int RM>
hipError_t xs_backward_rm(typename Cx<T>::type* psi, typename Cx<T>::type* lam, const XrTab& tab, const double* wt, const double* dgs,
                          long dgs_step, int S, int B, int L, int nblk, typename Cx<T>::type* park, double* part, hipStream_t st) {
  typedef typename Cx<T>::type T2;
  const size_t lds = xs_lds_bytes(NT * RPT, sizeof(T));
  const bool sgn = tab.sgn != nullptr, dc = xs_diagctl(tab);
  bool cplx = false;
  for (int j = 0; j < tab.R; ++j) cplx = cplx || tab.phase[j] != 0;
  if (cplx && !sgn) return hipErrorInvalidValue;
  void (*kf)(T2*, T2*, XrTab, XsOrder, XsOrder, const double*, const double*, long, int, int, int, T2*, double*) =
      cplx ? (dc ? xs_backward_kernel<T, NT, RPT, true, true, RM, PREF, true> : xs_backward_kernel<T, NT, RPT, true, false, RM, PREF, true>)
           : (sgn ? (dc ? xs_backward_kernel<T, NT, RPT, true, true, RM, PREF, false>
                        : xs_backward_kernel<T, NT, RPT, true, false, RM, PREF, false>)
                  : (dc ? xs_backward_kernel<T, NT, RPT, false, true, RM, PREF, false>
                        : xs_backward_kernel<T, NT, RPT, false, false, RM, PREF, false>));
  hipError_t e = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kf, dim3((unsigned)nblk), dim3(NT), lds, st, psi, lam, tab, xs_make_order(NT * RPT, tab, xs_threads(NT * RPT, 0) == NT ? 0 : 1, true),
                     xs_make_order(NT * RPT, tab, xs_threads(NT * RPT, 0) == NT ? 0 : 1, false), wt, dgs, dgs_step, S, B, L, park, part);
  return hipGetLastError();
}

// the quadrature form (tables of up to 16 XOR-diagonals, 4-point rule)
template <class T, int NT, int RPT>
hipError_t xs_backward_q_go(typename Cx<T>::type* psi, typename Cx<T>::type* lam, const XrTab& tab, const double* wt, const double* dgs,
                            long dgs_step, int S, int B, int L, int nblk, const XsQuad& qd, double* part, hipStream_t st) {
  typedef typename Cx<T>::type T2;
  if (!xs_quad_supported(tab, qd.ni)) return hipErrorInvalidValue;
  const size_t lds = xs_lds_bytes(NT * RPT, sizeof(T));
  const bool sgn = tab.sgn != nullptr, dc = xs_diagctl(tab);
  bool cplx = false;
  for (int j = 0; j < tab.R; ++j) cplx = cplx || tab.phase[j] != 0;
  if (cplx && !sgn) return hipErrorInvalidValue;
  void (*kf)(T2*, T2*, XrTab, XsOrder, XsOrder, const double*, const double*, long, int, int, int, XsQuad, double*) =
      cplx ? (dc ? xs_backward_q_kernel<T, NT, RPT, true, true, true, 2> : xs_backward_q_kernel<T, NT, RPT, true, false, true, 2>)
           : (sgn ? (dc ? xs_backward_q_kernel<T, NT, RPT, true, true, false, 2> : xs_backward_q_kernel<T, NT, RPT, true, false, false, 2>)
                  : (dc ? xs_backward_q_kernel<T, NT, RPT, false, true, false, 2> : xs_backward_q_kernel<T, NT, RPT, false, false, false, 2>));
  hipError_t e = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const int var = xs_threads(NT * RPT, 0) == NT ? 0 : 1;
  hipLaunchKernelGGL(kf, dim3((unsigned)nblk), dim3(NT), lds, st, psi, lam, tab, xs_make_order(NT * RPT, tab, var, true),
                     xs_make_order(NT * RPT, tab, var, false), wt, dgs, dgs_step, S, B, L, qd, part);
  return hipGetLastError();
}

template <class T, int NT, int RPT, bool PREF>
hipError_t xs_backward_go(typename Cx<T>::type* psi, typename Cx<T>::type* lam, const XrTab& tab, const double* wt, const double* dgs,
                          long dgs_step, int S, int B, int L, int nblk, typename Cx<T>::type* park, double* part, hipStream_t st) {
  return xs_rm(tab) > XR_RSMALL
             ? xs_backward_rm<T, NT, RPT, PREF, 2 * XR_RSMALL>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, st)
             : xs_backward_rm<T, NT, RPT, PREF, XR_RSMALL>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, st);
}

// variant: bit 0 = 1024 threads x 4 rows at d = 4096; bit 1 = prefetch of the next diagonal's partners in the sweeps without pairings
template <class T>
hipError_t xs_forward_t(const typename Cx<T>::type* in, typename Cx<T>::type* out, int d, const XrTab& tab, const double* wt,
                        const double* dgs, long dgs_step, int S, int B, int L, int nblk, int variant, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  if (!xs_supported(d, L, tab.R, tab.nd, sizeof(T))) return hipErrorInvalidValue;
  const bool pf = (variant & 2) != 0;
  if constexpr (sizeof(T) == sizeof(float)) {
    if (d == 8192) return xs_forward_go<T, 1024, 8>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, st);
  }
  switch (d) {
    case 4096:
      return (variant & 1) ? xs_forward_go<T, 1024, 4>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, st)
                           : xs_forward_go<T, 512, 8>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, st);
#ifndef XS_PROBE_4096_ONLY
    case 2048: return xs_forward_go<T, 256, 8>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, st);
    case 1024:
      return (variant & 1) ? xs_forward_go<T, 256, 4>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, st)
                           : xs_forward_go<T, 128, 8>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, st);
    case 512:
      return (variant & 1) ? xs_forward_go<T, 256, 2>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, st)
                           : xs_forward_go<T, 64, 8>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, st);
    default:
      return (variant & 1) ? xs_forward_go<T, 256, 1>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, st)
                           : xs_forward_go<T, 64, 4>(in, out, tab, wt, dgs, dgs_step, S, B, L, nblk, pf, st);
#else
    default: return hipErrorInvalidValue;
#endif
  }
}

template <class T, bool PREF>
hipError_t xs_backward_pick(typename Cx<T>::type* psi, typename Cx<T>::type* lam, int d, const XrTab& tab, const double* wt,
                            const double* dgs, long dgs_step, int S, int B, int L, int nblk, typename Cx<T>::type* park, double* part,
                            int variant, hipStream_t st) {
  if constexpr (sizeof(T) == sizeof(float)) {
    if (d == 8192) return xs_backward_go<T, 1024, 8, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, st);
  }
  switch (d) {
    case 4096:
      return (variant & 1) ? xs_backward_go<T, 1024, 4, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, st)
                           : xs_backward_go<T, 512, 8, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, st);
#ifndef XS_PROBE_4096_ONLY
    case 2048: return xs_backward_go<T, 256, 8, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, st);
    case 1024:
      return (variant & 1) ? xs_backward_go<T, 256, 4, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, st)
                           : xs_backward_go<T, 128, 8, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, st);
    case 512:
      return (variant & 1) ? xs_backward_go<T, 256, 2, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, st)
                           : xs_backward_go<T, 64, 8, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, st);
    default:
      return (variant & 1) ? xs_backward_go<T, 256, 1, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, st)
                           : xs_backward_go<T, 64, 4, PREF>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, st);
#else
    default: return hipErrorInvalidValue;
#endif
  }
}

template <class T>
hipError_t xs_backward_q_pick(typename Cx<T>::type* psi, typename Cx<T>::type* lam, int d, const XrTab& tab, const double* wt,
                              const double* dgs, long dgs_step, int S, int B, int L, int nblk, const XsQuad& qd, double* part,
                              int variant, hipStream_t st) {
  if constexpr (sizeof(T) == sizeof(float)) {
    if (d == 8192) return xs_backward_q_go<T, 1024, 8>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, st);
  }
  switch (d) {
    case 4096:
      return (variant & 1) ? xs_backward_q_go<T, 1024, 4>(psi, lam, tab, wt, dgs, dgs_step, S, B, L, nblk, qd, part, st)
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

template <class T>
hipError_t xs_backward_t(typename Cx<T>::type* psi, typename Cx<T>::type* lam, int d, const XrTab& tab, const double* wt,
                         const double* dgs, long dgs_step, int S, int B, int L, int nblk, typename Cx<T>::type* park, double* part,
                         int K, double dt, double* grad, int accumulate, int variant, const XsQuad* quad, hipStream_t st) {
  if (B <= 0 || S <= 0) return hipSuccess;
  if (!xs_supported(d, L, tab.R, tab.nd, sizeof(T)) || (!park && !quad) || !part) return hipErrorInvalidValue;
  // (no prefetch in the backward kernel: its sweeps carry more live registers, and the second partner buffer costs more in spills
  // than it hides in latency -- measured 24.5 against 26.7 ms per 20 steps at d = B = 4096)
  const hipError_t e = quad ? xs_backward_q_pick<T>(psi, lam, d, tab, wt, dgs, dgs_step, S, B, L, nblk, *quad, part, variant, st)
                            : xs_backward_pick<T, false>(psi, lam, d, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, variant, st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(xs_finish_kernel, dim3((unsigned)S), dim3(1024), 0, st, part, nblk, xs_waves(d, variant), xs_slots(tab),
                     xs_rm(tab), S, tab, xs_make_order(d, tab, variant, quad && XS_PAIR_LOCAL), K, dt, grad, accumulate);
  return hipGetLastError();
}

}  // namespace

hipError_t xs_forward_launch(const double2* in, double2* out, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                             int S, int B, int L, int nblk, int variant, hipStream_t st) {
  return xs_forward_t<double>(in, out, d, tab, wt, dgs, dgs_step, S, B, L, nblk, variant, st);
}
hipError_t xs_forward_launch(const float2* in, float2* out, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                             int S, int B, int L, int nblk, int variant, hipStream_t st) {
  return xs_forward_t<float>(in, out, d, tab, wt, dgs, dgs_step, S, B, L, nblk, variant, st);
}
hipError_t xs_backward_launch(double2* psi, double2* lam, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                              int S, int B, int L, int nblk, double2* park, double* part, int K, double dt, double* grad,
                              int accumulate, int variant, const XsQuad* quad, hipStream_t st) {
  return xs_backward_t<double>(psi, lam, d, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, K, dt, grad, accumulate, variant, quad, st);
}
hipError_t xs_backward_launch(float2* psi, float2* lam, int d, const XrTab& tab, const double* wt, const double* dgs, long dgs_step,
                              int S, int B, int L, int nblk, float2* park, double* part, int K, double dt, double* grad, int accumulate,
                              int variant, const XsQuad* quad, hipStream_t st) {
  return xs_backward_t<float>(psi, lam, d, tab, wt, dgs, dgs_step, S, B, L, nblk, park, part, K, dt, grad, accumulate, variant, quad, st);
}

}  // namespace dq
