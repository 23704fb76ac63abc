// Batched complex GEMM on the gfx950 matrix cores, fp64 (v_mfma_f64_16x16x4_f64) and fp32
// (v_mfma_f32_16x16x4_f32) complex.
//
//   C[z] = alpha * op(A[z]) * op(B[z]) + beta * C[z]       op in {N, C (conjugate transpose)}
//
// This one kernel is every O(d^3)/O(d^2 B) operation of the propagator hot path:
//   * propagator . state      U_t Psi_t                /root/reference/diffqc.cc:198, sim_plain.py:146
//   * the products inside exp(A_t) (polynomial evaluation; replaces Eigen's MatrixBase::exp()
//     diffqc.cc:198 and scipy.linalg.expm sim_plain.py:145)
//   * the adjoint sweep's U_t^H Lambda, Lambda Psi^H and the reverse-mode products.
// The reference's `typedef double Scalar` / commented `typedef float Scalar` (diffqc.cc:13-19) is the
// template parameter T here.
//
// Data layout: interleaved complex (re, im) pairs, row-major, explicit leading dimension (in complex
// elements) -- the memory layout of numpy complex128/complex64 and of std::complex<T>, so the C-ABI takes
// the reference's buffers as they are. One 16-byte (8-byte for fp32) global load / LDS read moves a whole
// complex number and yields both the real and the imaginary MFMA operand of a lane.
//
// Tiling (CDNA4): a BM x BN output tile per workgroup, 64-lane waves arranged WGM x WGN, each wave owning
// (BM/WGM) x (BN/WGN) complex outputs as 16x16 MFMA tiles. Two arithmetic forms:
//   4M: separate real/imag accumulators, 4 real MFMAs per complex tile and k-step; conjugation and the
//       minus sign of the imag*imag term ride on the f64 MFMA's neg bits (fp64 only);
//   3M: accumulators P1 = sum ar*br, P2 = sum ai*bi, P3 = sum (ar+ai)(br+bi): 3 real MFMAs, Cr = P1 - P2,
//       Ci = P3 - P1 - P2 formed in the epilogue (normwise- rather than componentwise-stable; measured
//       end-to-end error equal to 4M, tests/test_gpu_chain.py). Default on the 128 tile, always for fp32.
// <=256 VGPRs so two waves share each SIMD and hipcc emits the VGPR-form MFMA (with a 512-register budget
// it selects the AGPR form and brackets every MFMA with v_accvgpr copies: measured 138 vs 64 cycles per
// f64 MFMA, tools/mfma_f64_probe.hip). K is streamed in BK-deep slabs through double-buffered LDS images
// stored k-major ([k][m] and [k][n], pitch +1 element) so that fragment reads are 16 consecutive elements;
// the next slab's global loads are in flight while the MFMAs run (the big fp64 3M tile keeps four slab images and takes one
// barrier per two slabs: DEEP). Workgroup ids are remapped so that the 32
// workgroups resident on one XCD cover a 4 x 8 block of tiles and share their A/B panels through that
// XCD's L2.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include "dq_internal.h"

// Ablation switches for tools/zgemm_probe.hip only (never set in the library build): 1 = no barriers in the K loop,
// 2 = no global loads / LDS stores after the prologue, 32 = no operand sums before the third 3M product. Results are
// then wrong; the timings say what each part of the loop costs next to the MFMAs.
#ifndef DQ_ZGEMM_ABLATE
#define DQ_ZGEMM_ABLATE 0
#endif

namespace dq {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

// f64 MFMA neg bits (blgp field), decoded with tools/mfma_f64_probe.hip
#define DQ_NEG_NONE 0
#define DQ_NEG_A 1
#define DQ_NEG_B 2

template <class T>
struct Mfma;
template <>
struct Mfma<double> {
  typedef double4_t acc_t;
  template <int NEG>
  static __device__ __forceinline__ acc_t run(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, NEG);
  }
  // C/D: col = lane & 15, row = (lane >> 4) + 4 * reg   (f64 form only)
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <>
struct Mfma<float> {
  typedef float4_t acc_t;
  template <int NEG>
  static __device__ __forceinline__ acc_t run(float a, float b, acc_t c) {
    static_assert(NEG == 0, "the f32 MFMA has no neg bits (blgp permutes lanes there)");
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // C/D: col = lane & 15, row = 4 * (lane >> 4) + reg   (the dtype-independent 16x16 map)
  static __device__ __forceinline__ int row(int lane, int r) { return (lane >> 4) * 4 + r; }
};

template <class T, int BM, int BN, int BK, int WGM, int WGN, int OPA, int OPB, int ALG3M, int HERM = 0, int DEEP = 0>
__global__ void __launch_bounds__(WGM* WGN * 64, 2) zgemm_kernel(ZgemmArgs<T> g) {
  typedef typename Cx<T>::type T2;
  typedef typename Mfma<T>::acc_t acc_t;
  constexpr int NT = WGM * WGN * 64;
  constexpr int TM = BM / WGM, TN = BN / WGN;
  constexpr int MI = TM / 16, NI = TN / 16;
  constexpr int PA = BM + 1, PB = BN + 1;  // LDS pitch in complex elements
  constexpr int A_PER_T = (BM * BK + NT - 1) / NT;
  constexpr int B_PER_T = (BN * BK + NT - 1) / NT;
  static_assert(BK % 4 == 0, "BK must be a multiple of the MFMA k (4)");
  static_assert((BM * BK) % NT == 0 && (BN * BK) % NT == 0, "tile/threads mismatch");
  static_assert(ALG3M || sizeof(T) == 8, "fp32 uses the 3M form (no MFMA neg bits)");

  // DEEP: four slab images instead of two and ONE barrier per two slabs (the registers only allow one workgroup per CU at
  // this tile, so the LDS is free: 132 KB); the register staging stays one slab deep
  static_assert(!DEEP || !HERM, "the two-slabs-per-barrier loop is for full-output tiles (no K-split pieces)");
  constexpr int NBUF = DEEP ? 4 : 2;
  __shared__ T2 lds[NBUF * BK * (PA + PB)];
  T2* As = lds;                    // [NBUF][BK][PA]
  T2* Bs = lds + NBUF * BK * PA;   // [NBUF][BK][PB]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;

  // ---- workgroup -> tile mapping (XCD-aware, grouped along M) ------------------
  const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
  const int nkt = (g.K + BK - 1) / BK;
  int tile_m, tile_n, kt0 = 0, kt1 = nkt, piece = -1;
  if (HERM) {
    // (anti-)Hermitian output: tiles on and above the diagonal only; the tiles beyond the last full round of
    // workgroup slots are split along K (piece >= 0: raw partial tile -> workspace, summed by zgemm_fixup_kernel)
    static_assert(!HERM || BM == BN, "the (anti-)Hermitian output form needs square tiles");
    const int ntri = ntm * (ntm + 1) / 2;
    const int nfull = g.n_full >= 0 ? g.n_full : ntri;
    int pid = blockIdx.x;
    if (pid >= nfull) {
      piece = pid - nfull;
      pid = nfull + piece / g.split_f;
      const int per = (nkt + g.split_f - 1) / g.split_f;
      kt0 = (piece % g.split_f) * per;
      kt1 = min(nkt, kt0 + per);
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
  const T2* __restrict__ A = g.A + z * g.sA;
  const T2* __restrict__ B = g.B + z * g.sB;
  T2* __restrict__ C = g.C + z * g.sC;

  // 4M: accr = Re, acci = Im.   3M: accr = sum ar*br, acci = sum ai*bi, acc3 = sum (ar+ai')(br+bi')
  acc_t accr[MI][NI], acci[MI][NI], acc3[ALG3M ? MI : 1][ALG3M ? NI : 1];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      accr[i][j] = {0, 0, 0, 0};
      acci[i][j] = {0, 0, 0, 0};
      if (ALG3M) acc3[i][j] = {0, 0, 0, 0};
    }

  T2 ra[A_PER_T], rb[B_PER_T];

  // Stage one BK-deep slab into registers. op(A) is M x K.
  //   OPA == N: stored [m][k] (k contiguous)  -> lanes run along k, LDS write transposes
  //   OPA == C: stored [k][m] (m contiguous)  -> lanes run along m, LDS write is linear
  // op(B) is K x N.
  //   OPB == N: stored [k][n] (n contiguous)  -> linear
  //   OPB == C: stored [n][k] (k contiguous)  -> transposing
  // (Macros, not lambdas: a by-reference capture of ra/rb demoted them to scratch memory --
  //  4 scratch stores + 4 loads per lane per slab, visible as 3-8x WRITE_SIZE in the PMC profile.)
#define DQ_LOAD_SLAB(k0_)                                                                    \
  {                                                                                          \
    _Pragma("unroll") for (int j = 0; j < A_PER_T; ++j) {                                    \
      const int e = tid + j * NT;                                                            \
      const int m = (OPA == 0) ? e / BK : e % BM;                                            \
      const int k = (OPA == 0) ? e % BK : e / BM;                                            \
      const int gm = m0 + m, gk = (k0_) + k;                                                 \
      const long off = (OPA == 0) ? ((long)gm * g.lda + gk) : ((long)gk * g.lda + gm);       \
      T2 v;                                                                                  \
      v.x = 0;                                                                               \
      v.y = 0;                                                                               \
      if ((gm < g.M) && (gk < g.K)) v = A[off];                                              \
      ra[j] = v;                                                                             \
    }                                                                                        \
    _Pragma("unroll") for (int j = 0; j < B_PER_T; ++j) {                                    \
      const int e = tid + j * NT;                                                            \
      const int n = (OPB == 0) ? e % BN : e / BK;                                            \
      const int k = (OPB == 0) ? e / BN : e % BK;                                            \
      const int gn = n0 + n, gk = (k0_) + k;                                                 \
      const long off = (OPB == 0) ? ((long)gk * g.ldb + gn) : ((long)gn * g.ldb + gk);       \
      T2 v;                                                                                  \
      v.x = 0;                                                                               \
      v.y = 0;                                                                               \
      if ((gn < g.N) && (gk < g.K)) v = B[off];                                              \
      rb[j] = v;                                                                             \
    }                                                                                        \
  }
#define DQ_STORE_SLAB(buf_)                                                                  \
  {                                                                                          \
    T2* as_ = As + (buf_) * BK * PA;                                                         \
    T2* bs_ = Bs + (buf_) * BK * PB;                                                         \
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

  // sign bookkeeping (4M): a = (ar, sa*ai), b = (br, sb*bi)
  //   Cr += ar*br - (sa*sb) ai*bi ;  Ci += sb * ar*bi + sa * ai*br
  constexpr int NEG_II = (ALG3M || ((OPA != 0) != (OPB != 0))) ? DQ_NEG_NONE : DQ_NEG_A;  // -(sa*sb)
  constexpr int NEG_RI = (!ALG3M && OPB != 0) ? DQ_NEG_B : DQ_NEG_NONE;
  constexpr int NEG_IR = (!ALG3M && OPA != 0) ? DQ_NEG_A : DQ_NEG_NONE;

#define DQ_COMPUTE_SLAB(buf_)                                                                              \
  {                                                                                                          \
    const T2* as_ = As + (buf_) * BK * PA + wm * TM + (lane & 15);                                           \
    const T2* bs_ = Bs + (buf_) * BK * PB + wn * TN + (lane & 15);                                           \
    _Pragma("unroll") for (int ks = 0; ks < BK / 4; ++ks) {                                                  \
      const int kk = ks * 4 + (lane >> 4);                                                                   \
      T2 a[MI], b[NI];                                                                                       \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) a[i] = as_[kk * PA + i * 16];                           \
      _Pragma("unroll") for (int j = 0; j < NI; ++j) b[j] = bs_[kk * PB + j * 16];                           \
      if (ALG3M) {                                                                                           \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j) {      \
          accr[i][j] = Mfma<T>::template run<0>(a[i].x, b[j].x, accr[i][j]);                                 \
          acci[i][j] = Mfma<T>::template run<0>(a[i].y, b[j].y, acci[i][j]);                                 \
        }                                                                                                    \
        /* sums formed after the P1/P2 issue so they can reuse the fragment registers */                     \
        if (!(DQ_ZGEMM_ABLATE & 32)) {                                                                       \
          _Pragma("unroll") for (int i = 0; i < MI; ++i) a[i].x = (OPA != 0) ? a[i].x - a[i].y : a[i].x + a[i].y; \
          _Pragma("unroll") for (int j = 0; j < NI; ++j) b[j].x = (OPB != 0) ? b[j].x - b[j].y : b[j].x + b[j].y; \
        }                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j)        \
          acc3[i][j] = Mfma<T>::template run<0>(a[i].x, b[j].x, acc3[i][j]);                                 \
      } else {                                                                                               \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j) {      \
          accr[i][j] = Mfma<T>::template run<DQ_NEG_NONE>(a[i].x, b[j].x, accr[i][j]);                       \
          acci[i][j] = Mfma<T>::template run<NEG_RI>(a[i].x, b[j].y, acci[i][j]);                            \
          accr[i][j] = Mfma<T>::template run<NEG_II>(a[i].y, b[j].y, accr[i][j]);                            \
          acci[i][j] = Mfma<T>::template run<NEG_IR>(a[i].y, b[j].x, acci[i][j]);                            \
        }                                                                                                    \
      }                                                                                                      \
    }                                                                                                        \
  }

  if (DEEP) {
    // slabs in pairs: images (2p, 2p + 1) hold the pair being computed, the other two receive the next pair; slabs beyond K
    // load zeros through the bounds check (an odd slab count computes one zero slab)
    DQ_LOAD_SLAB(kt0 * BK);
    DQ_STORE_SLAB(0);
    DQ_LOAD_SLAB((kt0 + 1) * BK);
    DQ_STORE_SLAB(1);
    __syncthreads();
    for (int kt = kt0; kt < kt1; kt += 2) {
      const int cur = ((kt - kt0) >> 1) & 1;
      DQ_LOAD_SLAB((kt + 2) * BK);
      DQ_COMPUTE_SLAB(cur * 2);
      DQ_STORE_SLAB((cur ^ 1) * 2);
      DQ_LOAD_SLAB((kt + 3) * BK);
      DQ_COMPUTE_SLAB(cur * 2 + 1);
      DQ_STORE_SLAB((cur ^ 1) * 2 + 1);
      __syncthreads();
    }
  } else {
    DQ_LOAD_SLAB(kt0 * BK);
    DQ_STORE_SLAB(0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      // the tail slab (k >= K) loads zeros through the bounds check, so no branch is needed here
      if (!(DQ_ZGEMM_ABLATE & 2)) DQ_LOAD_SLAB((kt + 1) * BK);
      DQ_COMPUTE_SLAB((kt - kt0) & 1);
      if (!(DQ_ZGEMM_ABLATE & 2)) DQ_STORE_SLAB((kt - kt0 + 1) & 1);
      if (!(DQ_ZGEMM_ABLATE & 1)) __syncthreads();
    }
  }
#undef DQ_LOAD_SLAB
#undef DQ_STORE_SLAB
#undef DQ_COMPUTE_SLAB

  // ---- epilogue: C = alpha*acc + beta*C -----------------------------------------
  const T alpha = (T)g.alpha, beta = (T)g.beta;
  T2* __restrict__ W = (HERM && piece >= 0) ? g.ws + (long)piece * (BM * BN) : nullptr;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int lc = wn * TN + j * 16 + (lane & 15);
      const int col = n0 + lc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = wm * TM + i * 16 + Mfma<T>::row(lane, r);
        const int row = m0 + lr;
        T cr, ci;
        if (ALG3M) {  // Cr = P1 - s P2, Ci = P3 - P1 - s P2, s = +1 unless exactly one operand is conjugated
          const T p2 = ((OPA != 0) != (OPB != 0)) ? -acci[i][j][r] : acci[i][j][r];
          cr = accr[i][j][r] - p2;
          ci = acc3[i][j][r] - accr[i][j][r] - p2;
        } else {
          cr = accr[i][j][r];
          ci = acci[i][j][r];
        }
        if (HERM && piece >= 0) {  // raw partial tile -> workspace [piece][BM][BN]
          T2 v;
          v.x = cr;
          v.y = ci;
          W[lr * BN + lc] = v;
        } else if (row < g.M && col < g.N && (!HERM || col >= row)) {
          const long off = (long)row * g.ldc + col;
          T2 v;
          v.x = alpha * cr;
          v.y = alpha * ci;
          if (g.beta != 0.0) {
            const T2 c0 = C[off];
            v.x += beta * c0.x;
            v.y += beta * c0.y;
          }
          if (HERM && col == row) {  // exactly (anti-)Hermitian diagonal
            if (g.herm > 0) v.y = 0;
            else v.x = 0;
          }
          C[off] = v;
          if (HERM && col != row) {  // mirror image: +-conj (beta != 0 needs an (anti-)Hermitian C on entry)
            T2 m;
            m.x = g.herm > 0 ? v.x : -v.x;
            m.y = g.herm > 0 ? -v.y : v.y;
            C[(long)col * g.ldc + row] = m;
          }
        }
      }
    }
}

// sums the split_f partial tiles of every tail tile of the herm form in a fixed order and writes both triangles
template <class T, int BM, int BN>
__global__ void __launch_bounds__(256) zgemm_fixup_kernel(ZgemmArgs<T> g) {
  typedef typename Cx<T>::type T2;
  const int ntm = (g.M + BM - 1) / BM;
  int tile_m, tile_n;
  tri_tile(g.n_full + blockIdx.y, ntm, tile_m, tile_n);
  const T2* __restrict__ W = g.ws + (long)blockIdx.y * g.split_f * (BM * BN);
  const T alpha = (T)g.alpha, beta = (T)g.beta;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < BM * BN; e += gridDim.x * 256) {
    const int row = tile_m * BM + e / BN, col = tile_n * BN + e % BN;
    if (row < g.M && col < g.N && col >= row) {
      T sr = 0, si = 0;
      for (int f = 0; f < g.split_f; ++f) {
        const T2 w = W[(long)f * (BM * BN) + e];
        sr += w.x;
        si += w.y;
      }
      const long off = (long)row * g.ldc + col;
      T2 v;
      v.x = alpha * sr;
      v.y = alpha * si;
      if (g.beta != 0.0) {
        const T2 c0 = g.C[off];
        v.x += beta * c0.x;
        v.y += beta * c0.y;
      }
      if (col == row) {
        if (g.herm > 0) v.y = 0;
        else v.x = 0;
      }
      g.C[off] = v;
      if (col != row) {
        T2 m;
        m.x = g.herm > 0 ? v.x : -v.x;
        m.y = g.herm > 0 ? -v.y : v.y;
        g.C[(long)col * g.ldc + row] = m;
      }
    }
  }
}

// fix-up pass for the 128-tile herm form of zgemm_lin.hip (same workspace layout)
void zgemm_fixup_launch_128(const ZgemmArgs<double>& g, int rem, hipStream_t st) {
  hipLaunchKernelGGL((zgemm_fixup_kernel<double, 128, 128>), dim3(128 * 128 / 256, (unsigned)rem), dim3(256), 0, st, g);
}

template <class T, int BM, int BN, int BK, int WGM, int WGN, int ALG3M>
static void launch_cfg(ZgemmArgs<T> g, hipStream_t st, int n_cu) {
  const int ntm = (g.M + BM - 1) / BM, ntn = (g.N + BN - 1) / BN;
  dim3 block(WGM * WGN * 64);
  const int op = (g.opA ? 2 : 0) | (g.opB ? 1 : 0);
  g.n_full = -1;
  g.split_f = 1;
  if (g.herm) {
    const int ntri = ntm * (ntm + 1) / 2;
    const int nkt = (g.K + BK - 1) / BK;
    static std::atomic<int> wg_per_cu{0};  // resident workgroups per CU of this configuration
    if (wg_per_cu.load() == 0) {
      int nb = 0;
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, zgemm_kernel<T, BM, BN, BK, WGM, WGN, 1, 0, ALG3M, 1>,
                                                       WGM * WGN * 64, 0) != hipSuccess || nb <= 0) {
        (void)hipGetLastError();
        nb = 1;
      }
      wg_per_cu.store(nb);
    }
    const int slots = n_cu * wg_per_cu.load();
    const int rem = slots > 0 ? ntri % slots : 0;
    int grid_x = ntri;
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
    dim3 grid((unsigned)grid_x, 1, (unsigned)g.batch);
    switch (op) {
      case 0: hipLaunchKernelGGL((zgemm_kernel<T, BM, BN, BK, WGM, WGN, 0, 0, ALG3M, 1>), grid, block, 0, st, g); break;
      case 1: hipLaunchKernelGGL((zgemm_kernel<T, BM, BN, BK, WGM, WGN, 0, 1, ALG3M, 1>), grid, block, 0, st, g); break;
      case 2: hipLaunchKernelGGL((zgemm_kernel<T, BM, BN, BK, WGM, WGN, 1, 0, ALG3M, 1>), grid, block, 0, st, g); break;
      default: hipLaunchKernelGGL((zgemm_kernel<T, BM, BN, BK, WGM, WGN, 1, 1, ALG3M, 1>), grid, block, 0, st, g); break;
    }
    if (g.n_full >= 0)
      hipLaunchKernelGGL((zgemm_fixup_kernel<T, BM, BN>), dim3((unsigned)(BM * BN / 256), (unsigned)rem), dim3(256), 0, st, g);
    return;
  }
  dim3 grid((unsigned)(ntm * ntn), 1, (unsigned)g.batch);
  // two slabs per barrier on the big fp64 3M tile (its 256 VGPRs leave one workgroup per CU, so 132 KB of LDS are free)
  constexpr int DEEP = (sizeof(T) == 8 && BM == 128 && BN == 128 && ALG3M) ? 1 : 0;
  switch (op) {
    case 0: hipLaunchKernelGGL((zgemm_kernel<T, BM, BN, BK, WGM, WGN, 0, 0, ALG3M, 0, DEEP>), grid, block, 0, st, g); break;
    case 1: hipLaunchKernelGGL((zgemm_kernel<T, BM, BN, BK, WGM, WGN, 0, 1, ALG3M, 0, DEEP>), grid, block, 0, st, g); break;
    case 2: hipLaunchKernelGGL((zgemm_kernel<T, BM, BN, BK, WGM, WGN, 1, 0, ALG3M, 0, DEEP>), grid, block, 0, st, g); break;
    default: hipLaunchKernelGGL((zgemm_kernel<T, BM, BN, BK, WGM, WGN, 1, 1, ALG3M, 0, DEEP>), grid, block, 0, st, g); break;
  }
}

// Tile choice: the 128x128 tile needs >= ~1 wave of workgroups to pay; small problems
// (d <= 64, or a thin batch) take the 64x64 / 32x32 tiles so that more CUs get work.
#ifndef XS_ZGEMM_SMALL
#define XS_ZGEMM_SMALL 1
#endif
int zgemm_pick_tile(int M, int N, int batch) {
  const long t128 = (long)((M + 127) / 128) * ((N + 127) / 128) * batch;
  if (M >= 128 && N >= 128 && t128 >= 64) return 128;
  if (M > 32 && N > 32) return 64;
  return 32;
}

// alg: 0 = 4M (four real MFMAs per complex tile step), 1 = 3M (three; P3 trick, normwise-stable)
template <>
hipError_t zgemm_launch<double>(const ZgemmArgs<double>& g, hipStream_t st, int force_tile, int alg, int n_cu) {
  if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return hipSuccess;
  if (g.herm && g.M != g.N) return hipErrorInvalidValue;
  int tile = force_tile ? force_tile : zgemm_pick_tile(g.M, g.N, g.batch);
  // few tiles (the last levels of a product tree, one d = 256 product: 16 tiles of 64 x 64): the launch lasts as long as ONE wave's
  // chain of MFMAs over K, whatever the tile -- a 64 x 64 tile on four waves and a 32 x 32 tile on one are the same 32 x 32 per wave.
  // 32 x 32 tiles on four waves of 16 x 16 quarter that chain (code 16; complex128 only, not the Hermitian-output form)
  if (XS_ZGEMM_SMALL && !force_tile && tile == 64 && !g.herm && (long)((g.M + 63) / 64) * ((g.N + 63) / 64) * g.batch <= 128) tile = 16;
  if (tile == 128 && alg == 2 && g.opB == 0 && (g.opA == 1 || g.hintA != 0) && zgemm_lin_eligible(g)) {
    // left operand readable k-major: A^H as asked, or A = A^T / A^H / -A^H by the caller's hint
    if (g.opA == 1 || g.hintA == 2) return zgemm_lin_launch(g, 1, st, n_cu);
    if (g.hintA == 1 || g.hintA == 5) return zgemm_lin_launch(g, 0, st, n_cu);  // (5: the buffer already holds A^T, [K][M])
    ZgemmArgs<double> h = g;
    h.alpha = -g.alpha;
    return zgemm_lin_launch(h, 1, st, n_cu);
  }
  if (tile == 128) {
    if (alg >= 1) launch_cfg<double, 128, 128, 8, 2, 4, 1>(g, st, n_cu);
    else launch_cfg<double, 128, 128, 8, 2, 4, 0>(g, st, n_cu);
  } else if (tile == 64) launch_cfg<double, 64, 64, 8, 2, 2, 0>(g, st, n_cu);
  else if (tile == 16) launch_cfg<double, 32, 32, 8, 2, 2, 0>(g, st, n_cu);   // 32 x 32 tiles on four waves of 16 x 16: a quarter of the MFMA chain per wave
  else launch_cfg<double, 32, 32, 8, 1, 1, 0>(g, st, n_cu);
  return hipGetLastError();
}

template <>
hipError_t zgemm_launch<float>(const ZgemmArgs<float>& g, hipStream_t st, int force_tile, int /*alg*/, int n_cu) {
  if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return hipSuccess;
  if (g.herm && g.M != g.N) return hipErrorInvalidValue;
  const int tile = force_tile ? force_tile : zgemm_pick_tile(g.M, g.N, g.batch);
  if (tile == 128) launch_cfg<float, 128, 128, 8, 2, 4, 1>(g, st, n_cu);
  else if (tile == 64) launch_cfg<float, 64, 64, 8, 2, 2, 1>(g, st, n_cu);
  else launch_cfg<float, 32, 32, 8, 1, 1, 1>(g, st, n_cu);
  return hipGetLastError();
}

}  // namespace dq
