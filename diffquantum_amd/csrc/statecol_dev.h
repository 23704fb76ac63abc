// Device-side pieces shared by the state-column kernels (statecol.hip: forward chain + host side; statecol_bwd.hip: adjoint sweep,
// exact form; statecol_bwdq.hip: adjoint sweep, quadrature form): the table walk (XsCtl), the partner gathers class by class, the
// partner ring of the pairing sweeps, the wave sums.  Everything here has internal linkage (anonymous namespace).
#pragma once
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

// acc[slot] += tot for a wave-uniform slot as ONE vector instruction: v_add_f64 with its first source and its destination indexed
// (s_set_gpr_idx_on ..., SRC0 | DST).  hipcc's inline asm cannot name a register of a 32-register operand, so the accumulator vector
// is pinned to a fixed physical tuple and the instruction names its first pair; what the compiler makes of `acc[slot] += tot` is two
// indexed reads, the add and two indexed writes in three s_set_gpr_idx windows: 5 vector + 6 scalar instructions per diagonal.
// Taken by the quadrature kernels WITHOUT diagonal controls (ACCA; v[4:35] is where the allocator keeps the vector in those kernels
// anyway): adjoint launch -2.0 % on the headline table, -1.0 % on X/Y drives; with diagonal controls the pinned tuple collides with the
// row accumulators (ring QAOA +13 %, X/Z controls +16 %: spills) -- those keep the compiler's form.  (m0: saved and restored inside.)
#ifndef XS_ACC_ASM
#define XS_ACC_ASM 1
#endif
template <bool ACCA>
__device__ __forceinline__ void xs_acc_add(xs_acc16& acc, int slot, double tot) {
  if constexpr (ACCA && XS_ACC_ASM) {
    unsigned m0_keep;
    const unsigned idx = (unsigned)__builtin_amdgcn_readfirstlane(slot) * 2u;
    asm volatile("s_mov_b32 %0, m0\n\ts_set_gpr_idx_on %3, 0x9\n\tv_add_f64 v[4:5], v[4:5], %2\n\ts_set_gpr_idx_off\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_keep), "+{v[4:35]}"(acc)
                 : "v"(tot), "s"(idx));
  } else {
    acc[slot] += tot;
  }
}

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
// of one phase: a runtime branch cost 16 register copies per diagonal where its two updates merged).
// SIGN (0 / 1 / 2: unsigned table / parity signs whose masks stay within the thread bits / parity signs with row factors -- the loops
// of row-signed diagonals are instantiated for 2 only: their code alone, never run, cost an X / Y-drive table 17 % of its adjoint sweep):
// the signs along a diagonal of a Pauli string are a parity, sigma_ij = (-1)^(s0_j + parity(i & z_j)); with i = r NT + t that is a
// factor per THREAD (bit j of tsg: s0_j ^ parity(t & z_j), folded into the weight once per diagonal) times a factor per ROW that is
// the same for every thread (bit r of the diagonal's row mask: parity(r & (z_j / NT)), a scalar).  Most diagonals have no row
// factor at all (z_j within the thread bits): their loop is the unsigned one with a per-thread weight.  (Round 3 kept a sign word per
// row -- a shift, a mask and an xor per row and diagonal in every sweep, twice in the pairing sweeps.)
// ej = table entry >> 20: the diagonal's index in the table (5 bits), row-sign bits of rows 1 .. 7 at bits 5 .. 11.
// DOT: the level's gradient pairing of the partners with the thread's own rows xm, into lane `slot` of the accumulator vector.
template <class T, int RPT, int SIGN, bool DOT, int PH, bool RS = false, bool ACCA = false>
__device__ __forceinline__ void xs_use(int ej, int slot, T w, unsigned tsg, const typename Cx<T>::type* v, typename Cx<T>::type* o,
                                       const typename Cx<T>::type* xm, double* dacc) {
  // RS: the diagonal has a row factor (the host orders those last within every class and phase: a loop of their own, no branch here)
  const unsigned tb = SIGN ? (tsg >> (ej & 31)) & 1u : 0u;
  const unsigned rm = (SIGN && RS) ? ((unsigned)ej >> 4) & 0xfeu : 0u;   // bit r: row r carries one more minus (the same in every thread)
  const T wt = SIGN ? xs_flip(w, tb) : w;
  T wr[RPT];
#pragma unroll
  for (int r = 0; r < RPT; ++r) wr[r] = (SIGN && RS) ? xs_flip(wt, (rm >> r) & 1u) : wt;
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
    if constexpr (SIGN && RS) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        T p = PH ? -xs_fma(v[r].x, xm[r].x, v[r].y * xm[r].y) : xs_fma(v[r].x, xm[r].y, -v[r].y * xm[r].x);
        p = xs_flip(p, (rm >> r) & 1u);
        if (r & 1) sb += p;
        else sa += p;
      }
    } else {  // no row factor: the two products of a row are the two chains (2 FMAs per row; a product, an FMA and an add before)
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        sa = xs_fma(v[r].x, PH ? xm[r].x : xm[r].y, sa);
        sb = xs_fma(v[r].y, PH ? xm[r].y : xm[r].x, sb);
      }
      if (PH) sa = -sa;   // PH: -(sa + sb);  else sa - sb
      sb = -sb;
    }
    double tot = (double)sa + (double)sb;
    if (SIGN) tot = xs_flip(tot, tb);   // the thread's factor, once per diagonal
    xs_acc_add<ACCA>(*(xs_acc16*)dacc, slot, tot);
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
  int ent;    // xoff | phase << 16 | (index in the table: sign bit, accumulator) << 20 | row-sign bits << 25
  double w;   // dt u c_j of the current step
  int perm;   // this lane's index in the table
  unsigned long long gmask;   // XsOrder::gmask: which loops of the walk have diagonals
  int rsn;    // lane g: how many diagonals of group g (its LAST ones) carry a row factor of the sign; groups: LDS class 0 / 1 by phase
              // (0, 1 / 2, 3), then the register classes xh = 1 .. 7 by phase (4 + 2 (xh - 1) + phase)
  __device__ __forceinline__ void load_table(const XrTab& tab, const XsOrder& ord, int lane) {
    rsn = lane < 18 ? ord.rsn[lane] : 0;
    gmask = ord.gmask;
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
  // + the row-sign bits of a parity-signed table at bits 25 .. 31: bit 24 + r = parity(r & (z / NT)), r = 1 .. 7
  __device__ __forceinline__ void load_row_signs(const XrTab& tab, int lane, int log_nt) {
    if (lane < tab.R) {
      const unsigned zh = tab.zmask[perm] >> log_nt;
      unsigned rmask = 0;
#pragma unroll
      for (unsigned r = 1; r < 8; ++r) rmask |= (unsigned)(__popc(r & zh) & 1) << (24 + r);
      ent |= (int)rmask;
    }
  }
  // bit j of the result: the thread's factor of diagonal j, s0_j ^ parity(t & z_j) (t < NT: the thread bits of the row index)
  static __device__ __forceinline__ unsigned thread_signs(const XrTab& tab, int t) {
    unsigned w = tab.s0mask;
    for (int j = 0; j < tab.R; ++j) w ^= (unsigned)(__popc((unsigned)t & tab.zmask[j]) & 1) << j;
    return w;
  }
  __device__ __forceinline__ void load_step(const double* __restrict__ wrow, int lane) { w = lane < R ? wrow[perm] : 0.0; }
  __device__ __forceinline__ double fetch_step(const double* __restrict__ wrow, int lane) const { return lane < R ? wrow[perm] : 0.0; }
};

__device__ __forceinline__ unsigned xs_lds_addr(const void* p) {
  return (unsigned)(unsigned long long)(const __attribute__((address_space(3))) void*)p;
}
// the images of a column start at multiples of NT * sizeof(complex) in LDS (xs_fetch / xs_lds_ring fold the image's offset into the
// diagonal's scalar and XOR the thread's part onto it): dynamic LDS at offset 0 and images D * sizeof apart make it so -- trap otherwise
template <int NT, class T2>
__device__ __forceinline__ void xs_lds_aligned(const T2* first_image) {
  if ((xs_lds_addr(first_image) & (unsigned)(NT * sizeof(T2) - 1)) != 0u) __builtin_trap();
}
// partner values by LDS reads (volatile: neighbours are not to be merged into ds_read2 forms, which the LDS array serves at half
// the rate).  PERM: the offset permutes the thread's row list as well (xh != 0)
template <class T, int NT, int RPT, bool PERM>
__device__ __forceinline__ void xs_fetch(int xo, const typename Cx<T>::type* __restrict__ src, int t, typename Cx<T>::type* v) {
  typedef typename XsV2<T>::type V2;
  typedef const volatile __attribute__((address_space(3))) V2* lds_ptr;
  // address = image + ((t ^ xlo) << log2 sizeof): the images start at multiples of NT * sizeof (checked by xs_lds_aligned at kernel
  // start), so the image's offset and the diagonal's go into ONE scalar and the thread's part is one v_xor (was: v_bitop3 + v_lshl_add)
  const unsigned sa = xs_lds_addr(src) | ((unsigned)(xo & (NT - 1)) * (unsigned)sizeof(V2));
  lds_ptr base = (lds_ptr)(__attribute__((address_space(3))) void*)(unsigned long long)(((unsigned)t * (unsigned)sizeof(V2)) ^ sa);
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
#ifndef XS_NO_RS
#define XS_NO_RS 0        // probe builds: 1 = the loops of diagonals with a row factor of the sign are not instantiated
#endif
#ifndef XS_RING_SLOTS_DEPTH
#define XS_RING_SLOTS_DEPTH 1   // 1: the ring has DEPTH slots instead of one per row (see xs_lds_ring); 0: a slot per row (round 4)
#endif
#ifndef XS_RING_DEPTH_PLAIN
#define XS_RING_DEPTH_PLAIN 8
#endif
#ifndef XS_RING_DEPTH_PAIR
#define XS_RING_DEPTH_PAIR 4
#endif
typedef double xs_d2 __attribute__((ext_vector_type(2)));
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
// one counted wait for the reads of TWO slots (XS_RING_WAIT2: a wait per pair of rows instead of one per row)
template <int N>
__device__ __forceinline__ void xs_ds_wait2(xs_d2& q0, xs_d2& q1) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(q0), "+v"(q1) : "n"(N));
}
#ifndef XS_RING_CHAINS
#define XS_RING_CHAINS 2   // chains of the pairing sums per diagonal in the ring (4 until round 5: two more additions per diagonal)
#endif
#ifndef XS_RING_WAIT2
#define XS_RING_WAIT2 0
#endif
template <int R0, int R1, class F>
__device__ __forceinline__ void xs_rows(F&& f) {
  if constexpr (R0 < R1) {
    f(std::integral_constant<int, R0>{});
    xs_rows<R0 + 1, R1>(f);
  }
}
// one row's share of a diagonal: o += a_ij v and, with DOT, the pairing sums in four chains s[0..3]
// (sbit: the ROW's factor of the sign, a scalar; the thread's factor is in w and, for the pairing, applied to the diagonal's total)
template <int SIGN, bool DOT, int PH, int R>
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
    constexpr int c = XS_RING_CHAINS == 4 ? (R & 1) * 2 : 0;   // (two chains: a chain's FMAs are 4 instructions apart as it is -- the row's two updates of o in between)
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
// NSLOT: registers of the ring in complex numbers -- RPT (a slot per row) or DEPTH (RPT a multiple of DEPTH: element k of the class's
// (diagonal, row) sequence lives in slot k % DEPTH and its slot takes the read of element k + DEPTH as soon as it has been used: the
// same schedule in half the registers at DEPTH = 4, for the kernels that have none to spare)
template <int NT, int RPT, int SIGN, bool DOT, bool PERM, int PH, int SB, int DEPTH, bool RS, int NSLOT = RPT, bool ACCA = false>
__device__ __forceinline__ void xs_lds_ring(const XsCtl& ctl, int q0, int n, unsigned sg, const double2* __restrict__ src, int t,
                                            double sgn, double2* o, const double2* xm, double* dacc) {
  static_assert(DEPTH >= 1 && DEPTH <= RPT, "ring depth");
  static_assert(NSLOT == RPT || (NSLOT == DEPTH && RPT % DEPTH == 0), "ring slots");
  if (n <= 0) return;
  xs_d2 v[NSLOT];
  const unsigned base = xs_lds_addr(src), t16 = (unsigned)t << 4;   // (one v_xor per address: xs_fetch)
  int e = __builtin_amdgcn_readlane(ctl.ent, q0);
  unsigned a = t16 ^ (base | ((unsigned)(e & (NT - 1)) << 4));
  int xh = (e & 0xffff) / NT;
  xs_rows<0, DEPTH>([&](auto rc) {
    constexpr int r = decltype(rc)::value;
    if constexpr (PERM) xs_ds_read_first<0>(v[r], a + (unsigned)(((r ^ xh) * NT) << 4));
    else xs_ds_read_first<r * NT * 16>(v[r], a);
  });
  xs_rows<DEPTH, NSLOT>([&](auto rc) { v[decltype(rc)::value] = 0.0; });  // (defined before their first in/out use)
  const int qe = q0 + n;
  auto finish = [&](int q, const double* s, unsigned tb) {
    if (DOT) {
      const double tot = XS_RING_CHAINS == 4 ? (PH ? -((s[0] + s[1]) + (s[2] + s[3])) : (s[0] - s[1]) + (s[2] - s[3]))
                                             : (PH ? -(s[0] + s[1]) : s[0] - s[1]);
      xs_acc_add<ACCA>(*(xs_acc16*)dacc, q - SB, xs_flip(tot, tb));   // (the pairing takes the thread's factor once per diagonal)
    }
  };
#pragma clang loop unroll(disable)
  for (int q = q0; q + 1 < qe; ++q) {
    const int en = __builtin_amdgcn_readlane(ctl.ent, q + 1);
    const unsigned an = t16 ^ (base | ((unsigned)(en & (NT - 1)) << 4));
    const int xhn = (en & 0xffff) / NT;
    const unsigned tb = SIGN ? (sg >> ((e >> 20) & 31)) & 1u : 0u, rm = (SIGN && RS) ? ((unsigned)e >> 24) & 0xfeu : 0u;
    const double w = xs_flip(sgn * xs_readlane(ctl.w, q), tb);   // (the thread's factor of the sign goes into the weight)
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    xs_rows<0, RPT>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      if constexpr (XS_RING_WAIT2 && DEPTH >= 2 && RPT % 2 == 0) {
        if constexpr (r % 2 == 0) xs_ds_wait2<DEPTH - 2>(v[r % NSLOT], v[(r + 1) % NSLOT]);
      } else {
        xs_ds_wait<DEPTH - 1>(v[r % NSLOT]);
      }
      xs_ring_use<SIGN && RS, DOT, PH, r>(w, (SIGN && RS) ? (rm >> r) & 1u : 0u, v[r % NSLOT], o[r], xm, s);
      constexpr int k = r + DEPTH;  // the element whose read goes out now
      if constexpr (k < RPT) {
        if constexpr (PERM) xs_ds_read_next<0>(v[k % NSLOT], a + (unsigned)(((k ^ xh) * NT) << 4));
        else xs_ds_read_next<k * NT * 16>(v[k % NSLOT], a);
      } else {
        if constexpr (PERM) xs_ds_read_next<0>(v[(k - RPT) % NSLOT], an + (unsigned)((((k - RPT) ^ xhn) * NT) << 4));
        else xs_ds_read_next<(k - RPT) * NT * 16>(v[(k - RPT) % NSLOT], an);
      }
    });
    finish(q, s, tb);
    e = en;
    a = an;
    xh = xhn;
  }
  {
    const unsigned tb = SIGN ? (sg >> ((e >> 20) & 31)) & 1u : 0u, rm = (SIGN && RS) ? ((unsigned)e >> 24) & 0xfeu : 0u;
    const double w = xs_flip(sgn * xs_readlane(ctl.w, qe - 1), tb);
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    xs_rows<0, RPT>([&](auto rc) {
      constexpr int r = decltype(rc)::value;
      if constexpr (XS_RING_WAIT2 && DEPTH >= 2 && RPT % 2 == 0) {
        constexpr int c = RPT - r < DEPTH ? RPT - r : DEPTH;   // reads in flight at this pair
        if constexpr (r % 2 == 0) xs_ds_wait2<(c >= 2 ? c - 2 : 0)>(v[r % NSLOT], v[(r + 1) % NSLOT]);
      } else {
        xs_ds_wait<(RPT - 1 - r < DEPTH - 1 ? RPT - 1 - r : DEPTH - 1)>(v[r % NSLOT]);
      }
      xs_ring_use<SIGN && RS, DOT, PH, r>(w, (SIGN && RS) ? (rm >> r) & 1u : 0u, v[r % NSLOT], o[r], xm, s);
      constexpr int k = r + DEPTH;
      if constexpr (k < RPT) {
        if constexpr (PERM) xs_ds_read_next<0>(v[k % NSLOT], a + (unsigned)(((k ^ xh) * NT) << 4));
        else xs_ds_read_next<k * NT * 16>(v[k % NSLOT], a);
      }
    });
    finish(qe - 1, s, tb);
  }
}

// the n diagonals q0 .. q0 + n - 1 of one LDS class; PREF: the next diagonal's partners are fetched while this one's are used
template <class T, int NT, int RPT, int SIGN, bool DOT, int RM, bool PERM, bool PREF, int PH, int SB = 0, bool RS = false>
__device__ __forceinline__ void xs_lds_class(const XsCtl& ctl, int q0, int n, unsigned sg,
                                             const typename Cx<T>::type* __restrict__ src, int t, T sgn, typename Cx<T>::type* o,
                                             const typename Cx<T>::type* xm, double* dacc) {  // (SB: accumulator lane = position - SB)
  typedef typename Cx<T>::type T2;
  if (n <= 0) return;
  // (RM == 2 on a plain sweep: the caller asks for the ring there too -- phase 1 of the quadrature adjoint on parity-signed tables, where
  // it is 3 - 4 % faster per launch; on unsigned tables and in the forward kernel the burst loop is 4 - 7 % faster: round 5's A/B)
  if constexpr (!PREF && RPT == 8 && sizeof(T) == sizeof(double) && ((DOT && XS_RING_PAIR) || (!DOT && (XS_RING_PLAIN || RM == 2)))) {
    constexpr int DEPTH = DOT ? XS_RING_DEPTH_PAIR : XS_RING_DEPTH_PLAIN;
    xs_lds_ring<NT, RPT, SIGN, DOT, PERM, PH, SB, DEPTH, RS, (XS_RING_SLOTS_DEPTH && RPT % DEPTH == 0) ? DEPTH : RPT, (DOT && RM == 17)>(ctl, q0, n, sg, src, t, sgn, o, xm, dacc);   // (RM == 17: 16 accumulators, updated by xs_acc_add's one-instruction form)
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
      xs_use<T, RPT, SIGN, DOT, PH, RS, (DOT && RM == 17)>(e0 >> 20, q - SB, sgn * (T)xs_readlane(ctl.w, q), sg, va, o, xm, dacc);
      if (q + 2 < qe) {
        e0 = __builtin_amdgcn_readlane(ctl.ent, q + 2);
        xs_fetch<T, NT, RPT, PERM>(e0 & 0xffff, src, t, va);
      }
      xs_use<T, RPT, SIGN, DOT, PH, RS, (DOT && RM == 17)>(e1 >> 20, q + 1 - SB, sgn * (T)xs_readlane(ctl.w, q + 1), sg, vb, o, xm, dacc);
    }
    if (q < qe) xs_use<T, RPT, SIGN, DOT, PH, RS, (DOT && RM == 17)>(e0 >> 20, q - SB, sgn * (T)xs_readlane(ctl.w, q), sg, va, o, xm, dacc);
  } else {
#pragma clang loop unroll(disable)
    for (int q = q0; q < q0 + n; ++q) {
      const int e = __builtin_amdgcn_readlane(ctl.ent, q);
      T2 v[RPT];
      xs_fetch<T, NT, RPT, PERM>(e & 0xffff, src, t, v);
      xs_use<T, RPT, SIGN, DOT, PH, RS, (DOT && RM == 17)>(e >> 20, q - SB, sgn * (T)xs_readlane(ctl.w, q), sg, v, o, xm, dacc);
    }
  }
}

// a group = one LDS class and phase: its diagonals without a row factor of the sign, then (signed tables) those with one
template <class T, int NT, int RPT, int SIGN, bool DOT, int RM, bool PERM, bool PREF, int PH, int SB = 0>
__device__ __forceinline__ void xs_lds_group(const XsCtl& ctl, int q0, int n, unsigned sg, const typename Cx<T>::type* __restrict__ src,
                                             int t, T sgn, typename Cx<T>::type* o, const typename Cx<T>::type* xm, double* dacc) {
  constexpr int G = (PERM ? 4 : 0) + PH * 2;   // the group's two loops in XsOrder::gmask
  if (((unsigned)(ctl.gmask >> G) & 3u) == 0u) return;
  if constexpr (SIGN == 2 && !XS_NO_RS) {
    int nrs = __builtin_amdgcn_readlane(ctl.rsn, (PERM ? 2 : 0) + PH);
    // (a caller may pass a part of the group -- the pairing sweeps split at accumulator 16: the row-signed ones are the group's last)
    const int g_end = (PERM ? ctl.n0 + ctl.n1 : ctl.n0) - (PH ? 0 : (PERM ? ctl.n1p1 : ctl.n0p1));
    const int rs0 = g_end - nrs;                         // first row-signed position of the group
    const int qe = q0 + n, cut = rs0 < q0 ? q0 : (rs0 > qe ? qe : rs0);
    if ((unsigned)(ctl.gmask >> G) & 1u) xs_lds_class<T, NT, RPT, SIGN, DOT, RM, PERM, PREF, PH, SB, false>(ctl, q0, cut - q0, sg, src, t, sgn, o, xm, dacc);
    if ((unsigned)(ctl.gmask >> G) & 2u) xs_lds_class<T, NT, RPT, SIGN, DOT, RM, PERM, PREF, PH, SB, true>(ctl, cut, qe - cut, sg, src, t, sgn, o, xm, dacc);
  } else {
    xs_lds_class<T, NT, RPT, SIGN, DOT, RM, PERM, PREF, PH, SB, false>(ctl, q0, n, sg, src, t, sgn, o, xm, dacc);
  }
}

// a class of a pairing sweep with up to 32 diagonals: positions below 16 accumulate into the first accumulator vector, the others
// into the second -- two instantiations of the loop, no per-diagonal choice of the vector
template <class T, int NT, int RPT, int SIGN, int RM, bool PERM, int PH>
__device__ __forceinline__ void xs_pair_class(const XsCtl& ctl, int q0, int n, unsigned sg,
                                              const typename Cx<T>::type* __restrict__ src, int t, T sgn, typename Cx<T>::type* o,
                                              const typename Cx<T>::type* xm, xs_acc16* dacc) {
  const int qe = q0 + n;
  const int lo_end = qe < 16 ? qe : 16;
  xs_lds_group<T, NT, RPT, SIGN, true, RM, PERM, false, PH, 0>(ctl, q0, lo_end - q0, sg, src, t, sgn, o, xm, (double*)&dacc[0]);
  if constexpr (RM > 16) {
    const int hi0 = q0 > 16 ? q0 : 16;
    xs_lds_group<T, NT, RPT, SIGN, true, RM, PERM, false, PH, 16>(ctl, hi0, qe - hi0, sg, src, t, sgn, o, xm, (double*)&dacc[1]);
  }
}

// the pairing sweep (Horner on lambda): o += sgn (A - diag) v with every partner from LDS, in the order of ctl (no register class)
template <class T, int NT, int RPT, int SIGN, int RM, bool CPLX>
__device__ __forceinline__ void xs_gather_pairings(const XsCtl& ctl, unsigned sg, const typename Cx<T>::type* __restrict__ src,
                                                   int t, T sgn, typename Cx<T>::type* o, const typename Cx<T>::type* xm,
                                                   xs_acc16* dacc) {
  xs_pair_class<T, NT, RPT, SIGN, RM, false, 0>(ctl, 0, ctl.n0 - ctl.n0p1, sg, src, t, sgn, o, xm, dacc);
  if constexpr (CPLX) xs_pair_class<T, NT, RPT, SIGN, RM, false, 1>(ctl, ctl.n0 - ctl.n0p1, ctl.n0p1, sg, src, t, sgn, o, xm, dacc);
  xs_pair_class<T, NT, RPT, SIGN, RM, true, 0>(ctl, ctl.n0, ctl.n1 - ctl.n1p1, sg, src, t, sgn, o, xm, dacc);
  if constexpr (CPLX) xs_pair_class<T, NT, RPT, SIGN, RM, true, 1>(ctl, ctl.n0 + ctl.n1 - ctl.n1p1, ctl.n1p1, sg, src, t, sgn, o, xm, dacc);
}

// the diagonals whose partners are the thread's own rows r ^ XH (register copies), class by class, phase 0 then phase 1
template <class T, int RPT, int XH, int SIGN, bool DOT, int PH, bool RS = false, bool ACCA = false>
__device__ __forceinline__ int xs_local_run(const XsCtl& ctl, int q, int n, unsigned sg, const typename Cx<T>::type* cur, T sgn,
                                            typename Cx<T>::type* o, const typename Cx<T>::type* xm, double* dacc) {
#pragma clang loop unroll(disable)
  for (int i = 0; i < n; ++i, ++q) {
    const int e = __builtin_amdgcn_readlane(ctl.ent, q);
    typename Cx<T>::type v[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) v[r] = cur[r ^ XH];
    xs_use<T, RPT, SIGN, DOT, PH, RS, ACCA>(e >> 20, q, sgn * (T)xs_readlane(ctl.w, q), sg, v, o, xm, dacc);
  }
  return q;
}
// one register class and phase: the diagonals without a row factor of the sign, then those with one
template <class T, int RPT, int XH, int SIGN, bool DOT, int PH, bool ACCA = false>
__device__ __forceinline__ int xs_local_group(const XsCtl& ctl, int q, int n, unsigned sg, const typename Cx<T>::type* cur, T sgn,
                                              typename Cx<T>::type* o, const typename Cx<T>::type* xm, double* dacc) {
  constexpr int G = 8 + (XH - 1) * 4 + PH * 2;
  if constexpr (SIGN == 2 && !XS_NO_RS) {
    if (((unsigned)(ctl.gmask >> G) & 3u) == 0u) return q;
    int nrs = __builtin_amdgcn_readlane(ctl.rsn, 4 + 2 * (XH - 1) + PH);
    nrs = nrs > n ? n : nrs;
    if ((unsigned)(ctl.gmask >> G) & 1u) q = xs_local_run<T, RPT, XH, SIGN, DOT, PH, false, ACCA>(ctl, q, n - nrs, sg, cur, sgn, o, xm, dacc);
    if ((unsigned)(ctl.gmask >> G) & 2u) q = xs_local_run<T, RPT, XH, SIGN, DOT, PH, true, ACCA>(ctl, q, nrs, sg, cur, sgn, o, xm, dacc);
    return q;
  } else {
    return xs_local_run<T, RPT, XH, SIGN, DOT, PH, false, ACCA>(ctl, q, n, sg, cur, sgn, o, xm, dacc);
  }
}
template <class T, int RPT, int XH, int SIGN, bool DOT, bool CPLX, bool ACCA = false>
__device__ __forceinline__ void xs_local_classes(const XsCtl& ctl, int q, unsigned sg, const typename Cx<T>::type* cur, T sgn,
                                                 typename Cx<T>::type* o, const typename Cx<T>::type* xm, double* dacc) {
  if constexpr (XH < RPT && XH < 8) {
    if (((unsigned)(ctl.gmask >> (8 + (XH - 1) * 4)) & 15u) != 0u) {   // (an empty class: one test)
      q = xs_local_group<T, RPT, XH, SIGN, DOT, 0, ACCA>(ctl, q, ctl.nloc[XH] - ctl.nlocp1[XH], sg, cur, sgn, o, xm, dacc);
      if constexpr (CPLX) q = xs_local_group<T, RPT, XH, SIGN, DOT, 1, ACCA>(ctl, q, ctl.nlocp1[XH], sg, cur, sgn, o, xm, dacc);
    }
    xs_local_classes<T, RPT, XH + 1, SIGN, DOT, CPLX, ACCA>(ctl, q, sg, cur, sgn, o, xm, dacc);
  }
}

// o += sgn * (A - diag) v,  v = cur (the thread's rows, also in the LDS image `src`), class by class and phase by phase: no
// per-diagonal branching.  LOCAL = false: the caller's order (xs_make_order with local = false) has no register class -- every
// partner is read from LDS and the caller's copy of v need not stay in registers.  DOT: the sweep carries the gradient pairings; dacc is then an xs_acc16 indexed by the diagonal's SORTED
// position (xs_finish_kernel maps back).
// CPLX = false (real tables: every phase 0): the phase-1 loops are not even instantiated (their code alone cost the real kernels 4 %).
template <class T, int NT, int RPT, int SIGN, bool DOT, int RM, bool PREF, bool LOCAL, bool CPLX, bool STAGGER = false>
__device__ __forceinline__ void xs_gather(const XsCtl& ctl, unsigned sg, const typename Cx<T>::type* __restrict__ src,
                                          const typename Cx<T>::type* cur, int t, T sgn, typename Cx<T>::type* o,
                                          const typename Cx<T>::type* xm, double* dacc) {
  // STAGGER: the two waves that share a SIMD (w and w + 4 of a 512-thread workgroup) walk a plain sweep in opposite order of its two
  // kinds of work -- one starts with the register partners (vector work only) while the other issues its LDS reads.  Measured at
  // d = B = 4096 (tools/statecol_probe.hip): phase 1 of the quadrature adjoint -2.4 % per launch, the forward kernel +1.5 % (not used there)
  const bool local_first = STAGGER && SIGN == 0 && LOCAL && !DOT && NT >= 512 && ((t >> 8) & 1);   // (unsigned tables only: the second copy of the register classes is code the signed kernels have enough of)
  if (local_first) xs_local_classes<T, RPT, 1, SIGN, DOT, CPLX>(ctl, ctl.n0 + ctl.n1, sg, cur, sgn, o, xm, dacc);
  xs_lds_group<T, NT, RPT, SIGN, DOT, RM, false, PREF, 0>(ctl, 0, ctl.n0 - ctl.n0p1, sg, src, t, sgn, o, xm, dacc);
  if constexpr (CPLX) xs_lds_group<T, NT, RPT, SIGN, DOT, RM, false, PREF, 1>(ctl, ctl.n0 - ctl.n0p1, ctl.n0p1, sg, src, t, sgn, o, xm, dacc);
  xs_lds_group<T, NT, RPT, SIGN, DOT, RM, true, PREF, 0>(ctl, ctl.n0, ctl.n1 - ctl.n1p1, sg, src, t, sgn, o, xm, dacc);
  if constexpr (CPLX) xs_lds_group<T, NT, RPT, SIGN, DOT, RM, true, PREF, 1>(ctl, ctl.n0 + ctl.n1 - ctl.n1p1, ctl.n1p1, sg, src, t, sgn, o, xm, dacc);
  if (LOCAL && !local_first) xs_local_classes<T, RPT, 1, SIGN, DOT, CPLX>(ctl, ctl.n0 + ctl.n1, sg, cur, sgn, o, xm, dacc);
}

// the two halves of xs_gather on their own, for sweeps that run the register classes BEHIND the publish of their input (the writes of
// the image drain while the vector pipes work on the thread's own rows; the LDS classes follow the barrier)
template <class T, int NT, int RPT, int SIGN, int RM, bool PREF, bool CPLX>
__device__ __forceinline__ void xs_gather_lds(const XsCtl& ctl, unsigned sg, const typename Cx<T>::type* __restrict__ src, int t, T sgn,
                                              typename Cx<T>::type* o) {
  xs_lds_group<T, NT, RPT, SIGN, false, RM, false, PREF, 0>(ctl, 0, ctl.n0 - ctl.n0p1, sg, src, t, sgn, o, nullptr, nullptr);
  if constexpr (CPLX) xs_lds_group<T, NT, RPT, SIGN, false, RM, false, PREF, 1>(ctl, ctl.n0 - ctl.n0p1, ctl.n0p1, sg, src, t, sgn, o, nullptr, nullptr);
  xs_lds_group<T, NT, RPT, SIGN, false, RM, true, PREF, 0>(ctl, ctl.n0, ctl.n1 - ctl.n1p1, sg, src, t, sgn, o, nullptr, nullptr);
  if constexpr (CPLX) xs_lds_group<T, NT, RPT, SIGN, false, RM, true, PREF, 1>(ctl, ctl.n0 + ctl.n1 - ctl.n1p1, ctl.n1p1, sg, src, t, sgn, o, nullptr, nullptr);
}
template <class T, int RPT, int SIGN, bool CPLX>
__device__ __forceinline__ void xs_gather_local(const XsCtl& ctl, unsigned sg, const typename Cx<T>::type* cur, T sgn, typename Cx<T>::type* o) {
  xs_local_classes<T, RPT, 1, SIGN, false, CPLX>(ctl, ctl.n0 + ctl.n1, sg, cur, sgn, o, nullptr, nullptr);
}

template <class T, int NT, int RPT>
__device__ __forceinline__ void xs_put(typename Cx<T>::type* img, int t, const typename Cx<T>::type* v) {
#pragma unroll
  for (int r = 0; r < RPT; ++r) img[r * NT + t] = v[r];
}

}  // namespace

namespace {

// 16 per-lane values -> lane l < 16 holds the wave's total of value l (17 exchanges instead of 96).  Every stage has its own
// constant-indexed array, so that the values stay in registers.
// The four stages inside a row of 16 lanes take their partner through DPP (a register move in the vector pipe: no LDS round trip,
// where __shfl_xor is two ds_bpermute_b32 and their latency per exchange).  gfx9 DPP has no XOR patterns beyond the quad, but a
// reduction does not need them: the masks 8 (row_ror:8), 7 (row_half_mirror), 3 and 1 (quad_perm) generate the row's 16 lanes as
// well as 8, 4, 2, 1 do, and lane bit 3 / 2 / 1 / 0 still tells the two partners of a stage apart (every mask flips its bit).
#ifndef XS_DPP_SUMS
#define XS_DPP_SUMS 1
#endif
template <int CTRL>
__device__ __forceinline__ double xs_dpp_move(double v) {
  // (mov_dpp: no `old` operand -- update_dpp cost a v_mov 0 and two wait states in front of every move)
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double xs_wave_sums16_dpp(const double* a, int lane) {
  double b8[8], b4[4], b2[2];
  {
    const bool up = (lane & 8) != 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) b8[i] = (up ? a[i + 8] : a[i]) + xs_dpp_move<0x128>(up ? a[i] : a[i + 8]);   // row_ror:8 = lane ^ 8
  }
  {
    const bool up = (lane & 4) != 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) b4[i] = (up ? b8[i + 4] : b8[i]) + xs_dpp_move<0x141>(up ? b8[i] : b8[i + 4]);   // row_half_mirror = lane ^ 7
  }
  {
    const bool up = (lane & 2) != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) b2[i] = (up ? b4[i + 2] : b4[i]) + xs_dpp_move<0x1b>(up ? b4[i] : b4[i + 2]);   // quad_perm:[3,2,1,0] = lane ^ 3
  }
  const bool up = (lane & 1) != 0;
  double v = (up ? b2[1] : b2[0]) + xs_dpp_move<0xb1>(up ? b2[0] : b2[1]);   // quad_perm:[1,0,3,2] = lane ^ 1
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;  // lane l: the total of value (l & 15)
}
__device__ __forceinline__ double xs_wave_sums16(const double* a, int lane) {
  if (XS_DPP_SUMS) return xs_wave_sums16_dpp(a, lane);
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

// A wave's step sums into its slots of `part` (lanes < 16; a slot belongs to one wave, which adds its columns in order).  The
// read-modify-write had its load latency exposed at the end of every step (nothing else is left to do there); the add without a
// return value (global_atomic_add_f64) leaves the wave at once, and adds of one wave to one address keep their order.
#ifndef XS_W_PREFETCH
#define XS_W_PREFETCH 1
#endif
#ifndef XS_SETPRIO
#define XS_SETPRIO 0   // probe: 1 = the second of the two waves that share a SIMD (waves w and w + 4 of a 512-thread workgroup) runs at a lower issue priority
#endif
__device__ __forceinline__ void xs_stagger_priority(int t, int nt) {
#if XS_SETPRIO
  if (nt >= 512) {
    if ((t >> 8) & 1) __builtin_amdgcn_s_setprio(0);
    else __builtin_amdgcn_s_setprio(XS_SETPRIO);
  }
#else
  (void)t; (void)nt;
#endif
}
#ifndef XS_BWD_SPLIT
#define XS_BWD_SPLIT 1   // quadrature adjoint kernel: the same for both of its phases (statecol_bwdq.hip, publish_begin)
#endif
#ifndef XS_FWD_SPLIT
#define XS_FWD_SPLIT 1   // forward kernel: the register classes of a sweep run between the writes of its input image and the barrier
#endif
#ifndef XS_PART_ATOMIC
#define XS_PART_ATOMIC 1
#endif
__device__ __forceinline__ void xs_part_add(double* slot, double v, bool first) {
  if (first) {
    *slot = v;
  } else if (XS_PART_ATOMIC) {
    (void)unsafeAtomicAdd(slot, v);
  } else {
    *slot += v;
  }
}

// Diagonal controls that are Z-strings (XrTab::dstring): the step's sums sum_i D_jd[i] rho[i] over the thread's rows, rho = rowacc.
// D_jd[i] = dmag (-1)^parity(i & z), i = r NT + t: the rows' factor parity(r & (z / NT)) makes the sum over r a Walsh-Hadamard
// coefficient of rowacc (all RPT of them: RPT log2 RPT additions), the thread's factor parity(t & z) is one sign flip per source
// (bit jd of dsg).  dv[jd] = this thread's share of source jd.  Replaces nd x RPT table reads per thread and step (they were 19 k of
// the 22 k cycles a ring-QAOA step spent after its sweeps).
// The result leaves the magnitudes out: dv[jd] = (thread's sign) w[zh_jd], and xs_finish_kernel multiplies the step's total by
// dmag[jd] -- what the kernel needs per source is then 3 bits, all 16 sources in one 64-bit scalar (XsDiagCtl::zpack).  (Round 5's
// first form read dzmask / dmag / dkid from the kernel argument per source: three scalar loads and a wait each, in sequence; tables
// one per lane were spilled by the allocator, and so were the 16 sign masks it hoisted out of the step loop: 5.5 k of the 8.5 k
// cycles a ring-QAOA step spent behind its sweeps.)
struct XsDiagCtl {
  unsigned long long zpack;   // bits 4 jd .. 4 jd + 2: (dzmask[jd] / NT) & 7; bit 4 jd + 3: the source is a control
  __device__ __forceinline__ void load(const XrTab& tab, int log_nt) {
    unsigned long long z = 0;
    for (int jd = 0; jd < XR_DMAX; ++jd)
      if (jd < tab.nd && tab.dkid[jd] >= 0) z |= (unsigned long long)(((tab.dzmask[jd] >> log_nt) & 7u) | 8u) << (4 * jd);
    zpack = z;
  }
};
template <int RPT>
__device__ __forceinline__ void xs_diag_string_sums(const XsDiagCtl& dct, const double* rowacc, unsigned dsg, double* dv) {
  typedef double wvec __attribute__((ext_vector_type(8)));
  static_assert(RPT <= 8, "rows per thread");
  wvec w = 0.0;
  {
    double a[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) a[r] = r < RPT ? rowacc[r] : 0.0;
#pragma unroll
    for (int h = 1; h < 8; h <<= 1) {
      double b[8];
#pragma unroll
      for (int r = 0; r < 8; ++r) b[r] = (r & h) ? a[r ^ h] - a[r] : a[r] + a[r ^ h];
#pragma unroll
      for (int r = 0; r < 8; ++r) a[r] = b[r];
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) w[r] = a[r];   // w[zh] = sum_r (-1)^parity(r & zh) rowacc[r]
  }
  asm volatile("" : "+v"(dsg));   // (the sign masks are formed here, per step: hoisted, they were 32 more live registers -- spilled)
  const unsigned zlo = __builtin_amdgcn_readfirstlane((unsigned)dct.zpack), zhi = __builtin_amdgcn_readfirstlane((unsigned)(dct.zpack >> 32));
#pragma unroll
  for (int jd = 0; jd < XR_DMAX; ++jd) {
    const unsigned e = ((jd < 8 ? zlo : zhi) >> (4 * (jd & 7))) & 15u;   // wave-uniform: the vector is indexed through s_set_gpr_idx
    dv[jd] = (e & 8u) ? xs_flip(w[e & 7u], (dsg >> jd) & 1u) : 0.0;
  }
}
// bit jd: this thread's factor of Z-string source jd, parity(t & dzmask[jd])
__device__ __forceinline__ unsigned xs_diag_thread_signs(const XrTab& tab, int t) {
  unsigned w = 0;
  for (int jd = 0; jd < tab.nd; ++jd) w |= (unsigned)(__popc((unsigned)t & tab.dzmask[jd]) & 1) << jd;
  return w;
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

// what the adjoint kernels' launchers share
inline bool xs_table_is_complex(const XrTab& tab) {
  bool cplx = false;
  for (int j = 0; j < tab.R; ++j) cplx = cplx || tab.phase[j] != 0;
  return cplx;
}

}  // namespace

// (statecol.hip) LDS bytes / threads of a workgroup, diagonal controls present
size_t xs_lds_bytes(int d, size_t elem);
int xs_threads(int d, int variant);
bool xs_has_diag_controls(const XrTab& tab);
XsOrder xs_make_order(int d, const XrTab& tab, int variant, bool local);
bool xs_order_row_signs(const XsOrder& o);
// the adjoint kernels by form (statecol_bwd.hip / statecol_bwdq.hip); `variant` as in xs_backward_launch
template <class T>
hipError_t xs_backward_exact_launch(typename Cx<T>::type* psi, typename Cx<T>::type* lam, int d, const XrTab& tab, const double* wt,
                                    const double* dgs, long dgs_step, int S, int B, int L, int nblk, typename Cx<T>::type* park,
                                    double* part, const XsPoly& pol, int variant, double pr, double pi, hipStream_t st);
template <class T>
hipError_t xs_backward_quad_launch(typename Cx<T>::type* psi, typename Cx<T>::type* lam, int d, const XrTab& tab, const double* wt,
                                   const double* dgs, long dgs_step, int S, int B, int L, int nblk, const XsQuad& qd, double* part,
                                   int variant, double pr, double pi, hipStream_t st);

}  // namespace dq
