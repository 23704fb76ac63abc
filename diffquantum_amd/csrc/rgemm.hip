// Batched REAL GEMM on the gfx950 matrix cores (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32), with a
// symmetric-output form. It exists for the real-symmetric fast path of the propagator chain:
//
//   when H0 and every H_k are real (the reference's own problems are: Z/ZZ drifts and X controls,
//   /root/reference/demo_maxcut.py:36-79; BASELINE's random Ising H), the generator of a step is A = iX with
//   X = -dt (H0 + sum_k u_k H_k) REAL SYMMETRIC, and exp(A) = cos X + i sin X  (diffqc.cc:198 `dH.exp()`,
//   sim_plain.py:145 `scipy.linalg.expm(dH)`). cos X and sin X are polynomials in X evaluated with real
//   products of commuting symmetric matrices -- each a quarter of the MFMA work of a complex product, and
//   half of that again because the result is symmetric (only the tiles on and above the diagonal are computed;
//   the mirror image is written by the same workgroup).
//
//   C[z] = alpha * op(A[z]) * op(B[z]) + beta * C[z]        op in {N, T};  row-major, explicit leading dimensions
//   C[z] = alpha * (A[z] B[z] + B[z] A[z]) + beta * C[z]     symmetric A, B (not necessarily commuting), symmetric
//                                                              output form only (`two`): the reverse mode's
//                                                              sym(Wbar V) products without a full-matrix round trip
//
// Tiling follows zgemm.hip: BM x BN tile per workgroup, WGM x WGN waves, 16x16 MFMA tiles, K streamed in
// BK-deep slabs through double-buffered LDS images stored k-major; next slab's global loads in flight during
// the MFMAs; XCD-aware workgroup -> tile remap. Operands that are stored k-major already (op(A) = A^T with A
// stored [k][m]; B stored [k][n]) take the LINEAR form: one base pointer per thread, branch-free unchecked loads
// for every slab that lies inside K (rows/columns beyond M/N are clamped -- they only feed outputs that are never
// stored), linear LDS writes, pitch +16 elements so that the fragment reads (16 consecutive elements from each of
// 4 k-rows) are bank-conflict free. The transposing forms keep bounds-checked loads and pitch +1 (conflict-free
// transposing writes). The engine only issues the linear form (its operands are symmetric, so A = A^T).
//
// Symmetric form and the tail: the triangle has nt(nt+1)/2 tiles, which rarely fills the chip's workgroup slots
// evenly (d = 4096: 528 tiles on 512 slots = two rounds for 3 % more work than one). The tiles beyond the last
// full round are therefore split along K into `split_f` pieces each (one launch: full tiles first, then pieces
// writing raw partial tiles to a workspace) and summed in a fixed order by rgemm_fixup_kernel.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include "dq_internal.h"

// Ablation switches for tools/rgemm_probe.hip only (never set in the library build): 1 = no barriers in the K loop,
// 2 = no global loads / LDS stores after the prologue, 4 = fragments read from LDS once, 8 = no global loads (stale
// registers are stored), 16 = no LDS stores (loads are issued and waited for). Results are then wrong;
// the timings say what each part of the loop costs next to the MFMAs.
#ifndef DQ_RGEMM_ABLATE
#define DQ_RGEMM_ABLATE 0
#endif

namespace dq {

typedef double rdouble4_t __attribute__((ext_vector_type(4)));
typedef float rfloat4_t __attribute__((ext_vector_type(4)));

template <class T>
struct RMfma;
template <>
struct RMfma<double> {
  typedef rdouble4_t acc_t;
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct RMfma<float> {
  typedef rfloat4_t acc_t;
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

template <class T, int BM, int BN, int BK, int WGM, int WGN, int OPA, int OPB, int SYM, int MINW, int GLDS = 0>
__global__ void __launch_bounds__(WGM* WGN * 64, MINW) rgemm_kernel(RgemmArgs<T> g) {
  typedef typename RMfma<T>::acc_t acc_t;
  constexpr int NT = WGM * WGN * 64;
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr bool LINEAR = (OPA == 1 && OPB == 0);
  constexpr int PA = BM + (OPA == 1 ? 16 : 1), PB = BN + (OPB == 0 ? 16 : 1);  // LDS pitch in elements
  constexpr int A_PER_T = (BM * BK) / NT, B_PER_T = (BN * BK) / NT;
  constexpr int KA = NT / BM, KB = NT / BN;  // k-rows covered by one pass of the workgroup (linear form)
  static_assert(BK % 4 == 0, "BK must be a multiple of the MFMA k (4)");
  static_assert((BM * BK) % NT == 0 && (BN * BK) % NT == 0, "tile/threads mismatch");
  static_assert(NT % BM == 0 && NT % BN == 0, "linear staging needs whole k-rows per pass");
  static_assert(!SYM || (BM == BN && LINEAR), "the symmetric-output form needs square tiles and the linear form");
  // GLDS: the slabs go global -> LDS directly (global_load_lds_dwordx4: 16 B per lane, one wave instruction fills one
  // 1-KiB k-row of the image; no staging registers, no ds_write). Needs whole tiles and 16-B aligned rows (host checks).
  static_assert(!GLDS || (LINEAR && BM * sizeof(T) == 1024 && BN * sizeof(T) == 1024), "LDS-DMA staging: 1-KiB rows only");
  constexpr int NW = NT / 64;

  __shared__ T lds[2 * BK * (PA + PB)];
  T* As = lds;                // [2][BK][PA]
  T* Bs = lds + 2 * BK * PA;  // [2][BK][PB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave-uniform by construction: keep it in an SGPR
  const int wm = wave / WGN, wn = wave % WGN;

  // ---- workgroup -> tile (and, for tail pieces of the symmetric form, a K range) ---------------
  const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
  const int nkt = (g.K + BK - 1) / BK;
  const int nk_all = (SYM && g.two) ? 2 * nkt : nkt;  // two-product form: slabs nkt.. are those of B A
  int tile_m, tile_n, kt0 = 0, kt1 = nk_all, piece = -1;
  if (SYM) {
    const int ntri = ntm * (ntm + 1) / 2;
    const int nfull = g.n_full >= 0 ? g.n_full : ntri;
    int pid = blockIdx.x;
    if (pid >= nfull) {
      piece = pid - nfull;
      pid = nfull + piece / g.split_f;
      const int per = (nk_all + g.split_f - 1) / g.split_f;
      kt0 = (piece % g.split_f) * per;
      kt1 = min(nk_all, kt0 + per);
    } else if ((nfull & 7) == 0) {
      pid = (pid & 7) * (nfull >> 3) + (pid >> 3);
    }
    tri_tile(pid, ntm, tile_m, tile_n);
  } else {
    const int nwg = ntm * ntn;
    int pid = blockIdx.x;
    if ((nwg & 7) == 0) pid = (pid & 7) * (nwg >> 3) + (pid >> 3);
    constexpr int GROUP_M = 4;
    const int in_group = GROUP_M * ntn;
    const int group = pid / in_group;
    const int first_m = group * GROUP_M;
    const int gsz = min(ntm - first_m, GROUP_M);
    tile_m = first_m + (pid % in_group) % gsz;
    tile_n = (pid % in_group) / gsz;
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  const long z = blockIdx.z;
  // current operand pair: (A, B) for slabs < nkt, (B, A) from slab nkt on (two-product form)
  const T* A = g.A + z * g.sA;
  const T* B = g.B + z * g.sB;
  long lda = g.lda, ldb = g.ldb;
  int kt_shift = 0;  // slab index of the current pair's k = 0
  if (SYM && kt0 >= nkt) {
    A = g.B + z * g.sB; B = g.A + z * g.sA; lda = g.ldb; ldb = g.lda; kt_shift = nkt;
  }
  T* __restrict__ C = g.C + z * g.sC;

  acc_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = {0, 0, 0, 0};

  T ra[GLDS ? 1 : A_PER_T], rb[GLDS ? 1 : B_PER_T];
  constexpr int EPL = 16 / sizeof(T);  // elements per lane of one LDS-DMA instruction
  const T* ga = A + (long)((kt0 - kt_shift) * BK) * lda + m0 + lane * EPL;
  const T* gb = B + (long)((kt0 - kt_shift) * BK) * ldb + n0 + lane * EPL;
  // one wave instruction per k-row: row r of the slab -> LDS row r of buffer buf_ (base is wave-uniform, + lane*16 B)
  // (DQ_RGLDS_ASM, default: the DMA is issued through inline asm and the wait for it sits in front of the slab's barrier. With
  //  __builtin_amdgcn_global_load_lds hipcc puts an `s_waitcnt vmcnt(0)` in front of the next LDS read -- it cannot tell that the
  //  DMA writes the OTHER slab buffer -- which exposes the load latency once per slab; see zgemm_lin.hip.)
#ifndef DQ_RGLDS_ASM
#define DQ_RGLDS_ASM 1
#endif
#if DQ_RGLDS_ASM
#define DQ_RGLDS_ONE(gp_, lp_)                                                                                 \
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp_),                   \
               "s"((unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) void*)(lp_)) : "memory")
#define DQ_RGLDS_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define DQ_RGLDS_ONE(gp_, lp_)                                                                                 \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp_),                       \
                                   (__attribute__((address_space(3))) void*)(lp_), 16, 0, 0)
#define DQ_RGLDS_LANDED()
#endif
#define DQ_RGLDS(kt_, buf_)                                                                                    \
  {                                                                                                            \
    if (SYM && (kt_) == nkt && kt_shift == 0) { /* switch to the pair (B, A) */                                \
      const T* t_ = A; A = B; B = t_;                                                                          \
      const long l_ = lda; lda = ldb; ldb = l_;                                                                \
      kt_shift = nkt;                                                                                          \
      ga = A + m0 + lane * EPL;                                                                                \
      gb = B + n0 + lane * EPL;                                                                                \
    }                                                                                                          \
    _Pragma("unroll") for (int r = wave; r < BK; r += NW) {                                                    \
      DQ_RGLDS_ONE(ga + (long)r * lda, As + ((buf_) * BK + r) * PA);                                           \
      DQ_RGLDS_ONE(gb + (long)r * ldb, Bs + ((buf_) * BK + r) * PB);                                           \
    }                                                                                                          \
    ga += (long)BK * lda;                                                                                      \
    gb += (long)BK * ldb;                                                                                      \
  }

  // linear form: thread -> (k-row within a pass, column); columns beyond M/N are clamped (never stored)
  const int am = tid % BM, ak = tid / BM, bn = tid % BN, bk = tid / BN;
  const T* pa = A + (long)((kt0 - kt_shift) * BK + ak) * lda + min(m0 + am, g.M - 1);
  const T* pb = B + (long)((kt0 - kt_shift) * BK + bk) * ldb + min(n0 + bn, g.N - 1);
  long a_step = (long)KA * lda, b_step = (long)KB * ldb;

  // ---- staging (macros, not lambdas: by-reference captures of ra/rb end up in scratch memory, see zgemm.hip) ----
  // fast: the slab lies inside K; pa/pb point at its first row for this thread and are advanced afterwards
#define DQ_RLOAD_FAST()                                                                      \
  {                                                                                          \
    _Pragma("unroll") for (int j = 0; j < A_PER_T; ++j) ra[j] = pa[j * a_step];              \
    _Pragma("unroll") for (int j = 0; j < B_PER_T; ++j) rb[j] = pb[j * b_step];              \
    pa += (long)BK * lda;                                                                    \
    pb += (long)BK * ldb;                                                                    \
  }
  // checked: any op form, any position. op(A) is M x K:  OPA == 0: stored [m][k] (lanes along k, transposing
  // LDS write); OPA == 1: stored [k][m] (lanes along m). op(B) is K x N:  OPB == 0: stored [k][n]; 1: stored [n][k].
#define DQ_RLOAD_CHECKED(k0_)                                                                \
  {                                                                                          \
    _Pragma("unroll") for (int j = 0; j < A_PER_T; ++j) {                                    \
      const int e = tid + j * NT;                                                            \
      const int m = (OPA == 0) ? e / BK : e % BM;                                            \
      const int k = (OPA == 0) ? e % BK : e / BM;                                            \
      const int gm = m0 + m, gk = (k0_) + k;                                                 \
      const long off = (OPA == 0) ? ((long)gm * lda + gk) : ((long)gk * lda + gm);           \
      T v = 0;                                                                               \
      if ((gm < g.M) && (gk < g.K)) v = A[off];                                              \
      ra[j] = v;                                                                             \
    }                                                                                        \
    _Pragma("unroll") for (int j = 0; j < B_PER_T; ++j) {                                    \
      const int e = tid + j * NT;                                                            \
      const int n = (OPB == 0) ? e % BN : e / BK;                                            \
      const int k = (OPB == 0) ? e / BN : e % BK;                                            \
      const int gn = n0 + n, gk = (k0_) + k;                                                 \
      const long off = (OPB == 0) ? ((long)gk * ldb + gn) : ((long)gn * ldb + gk);           \
      T v = 0;                                                                               \
      if ((gn < g.N) && (gk < g.K)) v = B[off];                                              \
      rb[j] = v;                                                                             \
    }                                                                                        \
  }
#define DQ_RLOAD(kt_)                                                                        \
  {                                                                                          \
    if (SYM && (kt_) == nkt && kt_shift == 0) { /* switch to the pair (B, A) */              \
      const T* t_ = A; A = B; B = t_;                                                        \
      const long l_ = lda; lda = ldb; ldb = l_;                                              \
      kt_shift = nkt;                                                                        \
      pa = A + (long)ak * lda + min(m0 + am, g.M - 1);                                       \
      pb = B + (long)bk * ldb + min(n0 + bn, g.N - 1);                                       \
      a_step = (long)KA * lda; b_step = (long)KB * ldb;                                      \
    }                                                                                        \
    if (LINEAR && ((kt_) - kt_shift + 1) * BK <= g.K) DQ_RLOAD_FAST()                        \
    else DQ_RLOAD_CHECKED(((kt_) - kt_shift) * BK)                                           \
  }
#define DQ_RSTORE_SLAB(buf_)                                                                 \
  {                                                                                          \
    T* as_ = As + (buf_) * BK * PA;                                                          \
    T* bs_ = Bs + (buf_) * BK * PB;                                                          \
    _Pragma("unroll") for (int j = 0; j < A_PER_T; ++j) {                                    \
      const int e = tid + j * NT;                                                            \
      const int m = (OPA == 0) ? e / BK : e % BM;                                            \
      const int k = (OPA == 0) ? e % BK : e / BM;                                            \
      as_[k * PA + m] = ra[j];                                                               \
    }                                                                                        \
    _Pragma("unroll") for (int j = 0; j < B_PER_T; ++j) {                                    \
      const int e = tid + j * NT;                                                            \
      const int n = (OPB == 0) ? e % BN : e / BK;                                            \
      const int k = (OPB == 0) ? e / BN : e % BK;                                            \
      bs_[k * PB + n] = rb[j];                                                               \
    }                                                                                        \
  }
#define DQ_RCOMPUTE_SLAB(buf_)                                                               \
  {                                                                                          \
    const T* as_ = As + (buf_) * BK * PA + wm * TM + (lane & 15);                            \
    const T* bs_ = Bs + (buf_) * BK * PB + wn * TN + (lane & 15);                            \
    _Pragma("unroll") for (int ks = 0; ks < BK / 4; ++ks) {                                  \
      const int kk = ks * 4 + (lane >> 4);                                                   \
      T a[MI], b[NI];                                                                        \
      if (!(DQ_RGEMM_ABLATE & 4) || kt == kt0) {                                             \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) a[i] = as_[kk * PA + i * 16];         \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) b[j] = bs_[kk * PB + j * 16];         \
        if (DQ_RGEMM_ABLATE & 4) {                                                           \
          _Pragma("unroll") for (int i = 0; i < MI; ++i) fa[ks][i] = a[i];                   \
          _Pragma("unroll") for (int j = 0; j < NI; ++j) fb[ks][j] = b[j];                   \
        }                                                                                    \
      } else {                                                                               \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) a[i] = fa[ks][i];                     \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) b[j] = fb[ks][j];                     \
      }                                                                                      \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j) \
        acc[i][j] = RMfma<T>::run(a[i], b[j], acc[i][j]);                                    \
    }                                                                                        \
  }

  T fa[(DQ_RGEMM_ABLATE & 4) ? BK / 4 : 1][MI], fb[(DQ_RGEMM_ABLATE & 4) ? BK / 4 : 1][NI];  // ablation only
  if (GLDS) {
    if (kt0 < kt1) {
      DQ_RGLDS(kt0, 0);
      DQ_RGLDS_LANDED();
      __syncthreads();  // (builtin form: hipcc drains vmcnt before the barrier)
      for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (kt + 1 < kt1 && !(DQ_RGEMM_ABLATE & 2)) DQ_RGLDS(kt + 1, cur ^ 1);
        DQ_RCOMPUTE_SLAB(cur);
        DQ_RGLDS_LANDED();
        if (!(DQ_RGEMM_ABLATE & 1)) __syncthreads();
      }
    }
  } else if (kt0 < kt1) {
    DQ_RLOAD(kt0);
    DQ_RSTORE_SLAB(0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      const bool more = kt + 1 < kt1 && !(DQ_RGEMM_ABLATE & 2);  // uniform
      if (more && !(DQ_RGEMM_ABLATE & 8)) DQ_RLOAD(kt + 1);
      DQ_RCOMPUTE_SLAB(cur);
      if (more && !(DQ_RGEMM_ABLATE & 16)) DQ_RSTORE_SLAB(cur ^ 1);
      if (DQ_RGEMM_ABLATE & 16) {
        _Pragma("unroll") for (int j = 0; j < A_PER_T; ++j) asm volatile("" ::"v"(ra[j]));
        _Pragma("unroll") for (int j = 0; j < B_PER_T; ++j) asm volatile("" ::"v"(rb[j]));
      }
      if (!(DQ_RGEMM_ABLATE & 1)) __syncthreads();
    }
  }
#undef DQ_RGLDS
#undef DQ_RGLDS_ONE
#undef DQ_RGLDS_LANDED
#undef DQ_RLOAD_FAST
#undef DQ_RLOAD_CHECKED
#undef DQ_RLOAD
#undef DQ_RSTORE_SLAB
#undef DQ_RCOMPUTE_SLAB

  // ---- epilogue ---------------------------------------------------------------------------------
  if (SYM && piece >= 0) {  // raw partial tile -> workspace [piece][BM][BN]
    T* __restrict__ W = g.ws + (long)piece * (BM * BN);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int col = wn * TN + j * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; ++r) W[(wm * TM + i * 16 + RMfma<T>::row(lane, r)) * BN + col] = acc[i][j][r];
      }
    return;
  }
  // C = alpha*acc + beta*C; symmetric form: elements on/above the diagonal, each written to both triangles
  const T alpha = (T)g.alpha, beta = (T)g.beta;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int col = n0 + wn * TN + j * 16 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = m0 + wm * TM + i * 16 + RMfma<T>::row(lane, r);
        if (row < g.M && col < g.N && (!SYM || col >= row)) {
          const long off = (long)row * g.ldc + col;
          T v = alpha * acc[i][j][r];
          if (g.beta != 0.0) v += beta * C[off];
          C[off] = v;
          if (SYM && col != row) C[(long)col * g.ldc + row] = v;  // beta != 0 requires a symmetric C on entry
        }
      }
    }
}

// sums the split_f partial tiles of every tail tile in a fixed order and writes C (both triangles)
template <class T, int BM, int BN>
__global__ void __launch_bounds__(256) rgemm_fixup_kernel(RgemmArgs<T> g) {
  const int ntm = (g.M + BM - 1) / BM;
  int tile_m, tile_n;
  tri_tile(g.n_full + blockIdx.y, ntm, tile_m, tile_n);
  const T* __restrict__ W = g.ws + (long)blockIdx.y * g.split_f * (BM * BN);
  const T alpha = (T)g.alpha, beta = (T)g.beta;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < BM * BN; e += gridDim.x * 256) {
    const int row = tile_m * BM + e / BN, col = tile_n * BN + e % BN;
    if (row < g.M && col < g.N && col >= row) {
      T s = 0;
      for (int f = 0; f < g.split_f; ++f) s += W[(long)f * (BM * BN) + e];
      const long off = (long)row * g.ldc + col;
      T v = alpha * s;
      if (g.beta != 0.0) v += beta * g.C[off];
      g.C[off] = v;
      if (col != row) g.C[(long)col * g.ldc + row] = v;
    }
  }
}

template <class T, int BM, int BN, int BK, int WGM, int WGN, int MINW>
static void rlaunch_cfg(RgemmArgs<T> g, hipStream_t st, int n_cu) {
  const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
  dim3 block(WGM * WGN * 64);
  g.n_full = -1;
  g.split_f = 1;
  // LDS-DMA staging: whole tiles, 1-KiB tile rows, 16-B aligned row starts in every batch member
  constexpr int GL = (BM * sizeof(T) == 1024 && BN * sizeof(T) == 1024) ? 1 : 0;
  constexpr long EPL = 16 / sizeof(T);
  if (g.two && !g.sym) g.two = 0;
  const bool glds = GL && g.glds_ok && (g.sym || (g.opA == 1 && g.opB == 0)) && g.M % BM == 0 && g.N % BN == 0 &&
                    g.K % BK == 0 && g.lda % EPL == 0 && g.ldb % EPL == 0 && g.sA % EPL == 0 && g.sB % EPL == 0 &&
                    ((size_t)g.A % 16) == 0 && ((size_t)g.B % 16) == 0;
  if (g.sym) {
    static std::atomic<int> wg_per_cu{0};  // resident workgroups per CU of this configuration (registers + LDS)
    if (wg_per_cu.load() == 0) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, rgemm_kernel<T, BM, BN, BK, WGM, WGN, 1, 0, 1, MINW>,
                                                       WGM * WGN * 64, 0) != hipSuccess || nb <= 0) {
        (void)hipGetLastError();
        nb = 1;
      }
      wg_per_cu.store(nb);
    }
    const int slots = n_cu * wg_per_cu.load();
    // symmetric output: the operands commute and are symmetric, so op() is irrelevant -- always the linear form
    const int ntri = ntm * (ntm + 1) / 2;
    const int nkt = ((g.K + BK - 1) / BK) * (g.two ? 2 : 1);
    int grid_x = ntri;
    const int rem = slots > 0 ? ntri % slots : 0;
    // tail tiles that would occupy less than half of one more round are split along K
    if (g.batch == 1 && g.ws && ntri > slots && rem > 0 && 2 * rem <= slots) {
      int f = std::min(std::min(slots / rem, nkt), 64);
      while (f > 1 && (size_t)rem * f * BM * BN > g.ws_elems) --f;
      if (f > 1) {
        g.n_full = ntri - rem;
        g.split_f = f;
        grid_x = g.n_full + rem * f;
      }
    }
    if (glds)
      hipLaunchKernelGGL((rgemm_kernel<T, BM, BN, BK, WGM, WGN, 1, 0, 1, MINW, GL>), dim3((unsigned)grid_x, 1, (unsigned)g.batch),
                         block, 0, st, g);
    else
      hipLaunchKernelGGL((rgemm_kernel<T, BM, BN, BK, WGM, WGN, 1, 0, 1, MINW>), dim3((unsigned)grid_x, 1, (unsigned)g.batch),
                         block, 0, st, g);
    if (g.n_full >= 0)
      hipLaunchKernelGGL((rgemm_fixup_kernel<T, BM, BN>), dim3((unsigned)(BM * BN / 256), (unsigned)rem), dim3(256),
                         0, st, g);
    return;
  }
  dim3 grid((unsigned)(ntm * ntn), 1, (unsigned)g.batch);
  const int op = (g.opA ? 2 : 0) | (g.opB ? 1 : 0);
  switch (op) {
    case 0: hipLaunchKernelGGL((rgemm_kernel<T, BM, BN, BK, WGM, WGN, 0, 0, 0, MINW>), grid, block, 0, st, g); break;
    case 1: hipLaunchKernelGGL((rgemm_kernel<T, BM, BN, BK, WGM, WGN, 0, 1, 0, MINW>), grid, block, 0, st, g); break;
    case 2:
      if (glds) hipLaunchKernelGGL((rgemm_kernel<T, BM, BN, BK, WGM, WGN, 1, 0, 0, MINW, GL>), grid, block, 0, st, g);
      else hipLaunchKernelGGL((rgemm_kernel<T, BM, BN, BK, WGM, WGN, 1, 0, 0, MINW>), grid, block, 0, st, g);
      break;
    default: hipLaunchKernelGGL((rgemm_kernel<T, BM, BN, BK, WGM, WGN, 1, 1, 0, MINW>), grid, block, 0, st, g); break;
  }
}

// cfg: 0 = auto; 128x128 tile with 1: 4 waves of 64x64, 2: 8 waves of 64x32, 3: 16 waves of 32x32; 64 / 32 = small tiles.
// n_cu: compute units of the device (workgroup slots = n_cu * resident workgroups per CU of the configuration).
template <class T>
hipError_t rgemm_launch(const RgemmArgs<T>& g, hipStream_t st, int cfg, int n_cu) {
  if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return hipSuccess;
  if (g.sym && g.M != g.N) return hipErrorInvalidValue;
  if (cfg == 0) {
    const int t = zgemm_pick_tile(g.M, g.N, g.batch);
    cfg = t == 128 ? 3 : t;
  }
  if (cfg == 1) rlaunch_cfg<T, 128, 128, 16, 2, 2, 2>(g, st, n_cu);
  else if (cfg == 2) rlaunch_cfg<T, 128, 128, 16, 2, 4, 4>(g, st, n_cu);
  else if (cfg == 3) rlaunch_cfg<T, 128, 128, 16, 4, 4, 8>(g, st, n_cu);
  else if (cfg == 64) rlaunch_cfg<T, 64, 64, 16, 2, 2, 2>(g, st, n_cu);
  else if (cfg == 32) rlaunch_cfg<T, 32, 32, 16, 1, 1, 2>(g, st, n_cu);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}
template hipError_t rgemm_launch<double>(const RgemmArgs<double>&, hipStream_t, int, int);
template hipError_t rgemm_launch<float>(const RgemmArgs<float>&, hipStream_t, int, int);

}  // namespace dq
