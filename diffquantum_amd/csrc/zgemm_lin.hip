// Complex fp64 GEMM, "linear" form, for the big products of the propagator chain at d >= 1024:
//
//   C[z] = alpha * op(A[z]) * B[z] + beta * C[z]      op(A) read K-MAJOR: A is stored [k][m] (m contiguous), i.e.
//                                                      op(A) = A^T (CONJA = 0) or A^H (CONJA = 1); B stored [k][n]
//
// Every product of the headline step is of this kind once the structure of its left operand is used
// (/root/reference/diffqc.cc:198 `psi0 = dH.exp() * psi0`, sim_plain.py:146; the adjoint sweep's W U and U^H G):
//   U Psi      U = cos X + i sin X of a REAL symmetric X is complex symmetric: U = U^T          -> CONJA = 0
//   G = W U    W anti-Hermitian: W = -W^H                                                        -> CONJA = 1, alpha = -1
//   W' = U^H G                                                                                   -> CONJA = 1, HERM output
// and so are the products of the general complex tape whose left operand is a (anti-)Hermitian polynomial variable.
//
// Why a second kernel beside zgemm.hip: that one keeps 64x32 complex outputs per wave (192 accumulator VGPRs in the
// 3M form), so only 2 waves fit a SIMD, and its measured ceiling is the one of every 2-waves-per-SIMD f64 MFMA loop on
// this chip (~81 % of the matrix pipe; the real GEMM shows the same 80.5 % at 2 waves and 88 % at 4-8 waves per SIMD,
// rgemm.hip / DESIGN.md).  Here: 128 x 128 tile, SIXTEEN waves (4 x 4), 32 x 32 complex outputs per wave = 96
// accumulator VGPRs, no staging registers at all -> <= 128 VGPRs -> 4 waves per SIMD.
//   * Staging is LDS-DMA only (global_load_lds_dwordx4: one wave instruction moves one 1-KiB half row of the k-major
//     image straight into LDS; no VGPRs, no ds_write).  That is what makes the register budget: the register-staged
//     16-wave variant of the old kernel spills (6 VGPRs) and was slower than the 8-wave one.
//   * LDS images [k][128] complex with NO padding: a fragment read is 16 consecutive complex numbers (256 B = all 64
//     banks) from each of 4 k-rows whose pitch (2 KiB) is a multiple of 256 B -> every 16-lane group of a
//     ds_read_b128 covers the 64 banks exactly once.
//   * BK = 16: one barrier per 48 MFMAs per wave; two slab buffers (128 KB of the CU's 160 KB; the 1024-thread
//     workgroup owns the CU anyway).
//   * 3M product (P1 = sum ar br, P2 = sum ai bi, P3 = sum (ar + s ai)(br + bi), s = -1 for A^H), as zgemm.hip.
//   * (Anti-)Hermitian output form with the K-split tail and the fixed-order fix-up pass, as zgemm.hip.
// Shapes it takes: M, N multiples of 128, K a multiple of 16, 16-B aligned rows; everything else stays on zgemm.hip.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <atomic>
#include "dq_internal.h"

namespace dq {

typedef double zl_double4_t __attribute__((ext_vector_type(4)));

// Ablation switches for tools/zgemm_lin_probe.hip only (never set in the library build; results are then wrong): 1 = no barrier
// in the K loop, 2 = no DMA after the prologue, 4 = no wait for the DMA in front of the barrier.
#ifndef DQ_ZL_ABLATE
#define DQ_ZL_ABLATE 0
#endif
#ifndef DQ_ZL_DMA_AFTER
#define DQ_ZL_DMA_AFTER 0  // k-steps of a slab computed before the next slab's DMA is issued (0 = DMA first)
#endif
#define DQ_ZL_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f64_16x16x4f64((a_), (b_), (c_), 0, 0, 0)

template <int CONJA, int HERM>
__global__ void __launch_bounds__(1024, 4) zgemm_lin_kernel(ZgemmArgs<double> g) {
  typedef zl_double4_t acc_t;
  constexpr int BM = 128, BN = 128, BK = 16;
  constexpr int WGN = 4;
  constexpr int TM = 32, TN = 32, MI = 2, NI = 2;
  constexpr int NW = 16;

  __shared__ double2 lds[2 * BK * (BM + BN)];  // 128 KB
  double2* As = lds;                 // [2][BK][BM]
  double2* Bs = lds + 2 * BK * BM;   // [2][BK][BN]

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;

  // ---- workgroup -> tile (and, for the tail pieces of the herm form, a K range) -----------------
  const int ntm = g.M / BM, ntn = g.N / BN;
  const int nkt = g.K / BK;
  int tile_m, tile_n, kt0 = 0, kt1 = nkt, piece = -1;
  if (HERM) {
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
    if ((nwg & 7) == 0) pid = (pid & 7) * (nwg >> 3) + (pid >> 3);  // the 32 workgroups of one XCD: a 4 x 8 block of tiles
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
  const double2* __restrict__ A = g.A + z * g.sA;
  const double2* __restrict__ B = g.B + z * g.sB;
  double2* __restrict__ C = g.C + z * g.sC;

  // accr = sum ar*br, acci = sum ai*bi, acc3 = sum (ar + s ai)(br + bi)
  acc_t accr[MI][NI], acci[MI][NI], acc3[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      accr[i][j] = {0, 0, 0, 0};
      acci[i][j] = {0, 0, 0, 0};
      acc3[i][j] = {0, 0, 0, 0};
    }

  // ---- LDS-DMA staging: a slab is BK rows x 2 halves (1 KiB each) per operand = 64 wave instructions, 4 per wave ----
  // wave w moves half h = w & 1 of the rows (w >> 1) and (w >> 1) + 8 of both operands
  const int half = wave & 1, row0 = wave >> 1;
  const double2* ga = A + (long)(kt0 * BK + row0) * g.lda + m0 + half * 64 + lane;
  const double2* gb = B + (long)(kt0 * BK + row0) * g.ldb + n0 + half * 64 + lane;
  const long a_step8 = 8 * g.lda, b_step8 = 8 * g.ldb;
  // (inline asm, not __builtin_amdgcn_global_load_lds: for the builtin hipcc puts an `s_waitcnt vmcnt(0)` in front of the
  //  next LDS read -- it cannot tell that the DMA writes the OTHER slab buffer -- which exposes the whole load latency once
  //  per slab.  The asm form is invisible to that pass; the one wait that is needed sits in front of the slab's barrier.)
  const unsigned lds_a = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) void*)(As + row0 * BM + half * 64);
  const unsigned lds_b = (unsigned)(__UINTPTR_TYPE__)(__attribute__((address_space(3))) void*)(Bs + row0 * BN + half * 64);
#define DQ_ZL_DMA1(gp_, lds_)                                                                                            \
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gp_), "s"(lds_) : "memory")
#define DQ_ZL_DMA(buf_)                                                                                                  \
  {                                                                                                                      \
    const unsigned oa_ = lds_a + (unsigned)((buf_) * BK * BM * 16), ob_ = lds_b + (unsigned)((buf_) * BK * BN * 16);     \
    DQ_ZL_DMA1(ga, oa_);                                                                                                 \
    DQ_ZL_DMA1(gb, ob_);                                                                                                 \
    DQ_ZL_DMA1(ga + a_step8, oa_ + 8 * BM * 16);                                                                         \
    DQ_ZL_DMA1(gb + b_step8, ob_ + 8 * BN * 16);                                                                         \
    ga += 2 * a_step8;                                                                                                   \
    gb += 2 * b_step8;                                                                                                   \
  }
#define DQ_ZL_LANDED() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")

#define DQ_ZL_COMPUTE(buf_, ks0_, ks1_)                                                                                  \
  {                                                                                                                      \
    const double2* as_ = As + (buf_) * BK * BM + (lane >> 4) * BM + wm * TM + (lane & 15);                               \
    const double2* bs_ = Bs + (buf_) * BK * BN + (lane >> 4) * BN + wn * TN + (lane & 15);                               \
    _Pragma("unroll") for (int ks = (ks0_); ks < (ks1_); ++ks) {                                                         \
      double2 a[MI], b[NI];                                                                                              \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) a[i] = as_[ks * 4 * BM + i * 16];                                   \
      _Pragma("unroll") for (int j = 0; j < NI; ++j) b[j] = bs_[ks * 4 * BN + j * 16];                                   \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j) {                    \
        accr[i][j] = DQ_ZL_MFMA(a[i].x, b[j].x, accr[i][j]);                                                             \
        acci[i][j] = DQ_ZL_MFMA(a[i].y, b[j].y, acci[i][j]);                                                             \
      }                                                                                                                  \
      /* operand sums formed after the P1/P2 issue so that they reuse the fragment registers */                          \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) a[i].x = CONJA ? a[i].x - a[i].y : a[i].x + a[i].y;                 \
      _Pragma("unroll") for (int j = 0; j < NI; ++j) b[j].x = b[j].x + b[j].y;                                           \
      _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < NI; ++j)                      \
        acc3[i][j] = DQ_ZL_MFMA(a[i].x, b[j].x, acc3[i][j]);                                                             \
    }                                                                                                                    \
  }

  if (kt0 < kt1) {
    DQ_ZL_DMA(0);
    DQ_ZL_LANDED();
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
      const int cur = (kt - kt0) & 1;
      // the next slab's DMA is issued BEHIND the first k-step's fragment reads and MFMAs: right after the barrier every wave
      // of the workgroup is at the same point, and four DMA issues per wave in front of the first MFMA left the matrix pipe
      // idle for that long once per slab (SQ_VALU_MFMA_BUSY 91.6 % of the kernel's cycles before, tools/profile_sq.sh)
      DQ_ZL_COMPUTE(cur, 0, DQ_ZL_DMA_AFTER);
      if (kt + 1 < kt1 && !(DQ_ZL_ABLATE & 2)) DQ_ZL_DMA(cur ^ 1);  // in flight during the rest of this slab's 48 MFMAs per wave
      DQ_ZL_COMPUTE(cur, DQ_ZL_DMA_AFTER, BK / 4);
      if (!(DQ_ZL_ABLATE & 4)) DQ_ZL_LANDED();
      if (!(DQ_ZL_ABLATE & 1)) __syncthreads();
    }
  }
#undef DQ_ZL_DMA
#undef DQ_ZL_DMA1
#undef DQ_ZL_LANDED
#undef DQ_ZL_COMPUTE

  // ---- epilogue: Cr = P1 - s P2, Ci = P3 - P1 - s P2 (s = -1 for A^H); C = alpha*acc + beta*C ----------------
  const double alpha = g.alpha, beta = g.beta;
  double2* __restrict__ W = (HERM && piece >= 0) ? g.ws + (long)piece * (BM * BN) : nullptr;
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) {
      const int lc = wn * TN + j * 16 + (lane & 15);
      const int col = n0 + lc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = wm * TM + i * 16 + (lane >> 4) + 4 * r;
        const int row = m0 + lr;
        const double p2 = CONJA ? -acci[i][j][r] : acci[i][j][r];
        const double cr = accr[i][j][r] - p2;
        const double ci = acc3[i][j][r] - accr[i][j][r] - p2;
        if (HERM && piece >= 0) {  // raw partial tile -> workspace [piece][BM][BN]
          W[lr * BN + lc] = make_double2(cr, ci);
        } else if (!HERM || col >= row) {
          const long off = (long)row * g.ldc + col;
          double2 v = make_double2(alpha * cr, alpha * ci);
          if (beta != 0.0) {
            const double2 c0 = C[off];
            v.x += beta * c0.x;
            v.y += beta * c0.y;
          }
          if (HERM && col == row) {  // exactly (anti-)Hermitian diagonal
            if (g.herm > 0) v.y = 0;
            else v.x = 0;
          }
          C[off] = v;
          if (HERM && col != row)  // mirror image: +-conj (beta != 0 needs an (anti-)Hermitian C on entry)
            C[(long)col * g.ldc + row] = make_double2(g.herm > 0 ? v.x : -v.x, g.herm > 0 ? -v.y : v.y);
        }
      }
    }
}

// the tail fix-up pass of the herm form is zgemm.hip's (same workspace layout)
void zgemm_fixup_launch_128(const ZgemmArgs<double>& g, int rem, hipStream_t st);

bool zgemm_lin_eligible(const ZgemmArgs<double>& g) {
  return g.M > 0 && g.N > 0 && g.K > 0 && g.M % 128 == 0 && g.N % 128 == 0 && g.K % 16 == 0 &&
         ((size_t)g.A % 16) == 0 && ((size_t)g.B % 16) == 0 && (!g.herm || g.M == g.N) &&
         (long)(g.M / 128) * (g.N / 128) * g.batch >= 64;
}

// conj_a: op(A) = A^H (else A^T); A is read k-major in both cases
hipError_t zgemm_lin_launch(ZgemmArgs<double> g, int conj_a, hipStream_t st, int n_cu) {
  const int ntm = g.M / 128, ntn = g.N / 128;
  g.n_full = -1;
  g.split_f = 1;
  if (g.herm) {
    const int ntri = ntm * (ntm + 1) / 2;
    const int nkt = g.K / 16;
    const int slots = n_cu;  // one 1024-thread workgroup per CU
    const int rem = slots > 0 ? ntri % slots : 0;
    int grid_x = ntri;
    // tail tiles that would occupy less than half of one more round are split along K
    if (g.batch == 1 && g.ws && ntri > slots && rem > 0 && 2 * rem <= slots) {
      int f = std::min(std::min(slots / rem, nkt), 64);
      while (f > 1 && (size_t)rem * f * 128 * 128 > g.ws_elems) --f;
      if (f > 1) {
        g.n_full = ntri - rem;
        g.split_f = f;
        grid_x = g.n_full + rem * f;
      }
    }
    dim3 grid((unsigned)grid_x, 1, (unsigned)g.batch);
    if (conj_a) hipLaunchKernelGGL((zgemm_lin_kernel<1, 1>), grid, dim3(1024), 0, st, g);
    else hipLaunchKernelGGL((zgemm_lin_kernel<0, 1>), grid, dim3(1024), 0, st, g);
    if (g.n_full >= 0) zgemm_fixup_launch_128(g, rem, st);
    return hipGetLastError();
  }
  dim3 grid((unsigned)(ntm * ntn), 1, (unsigned)g.batch);
  if (conj_a) hipLaunchKernelGGL((zgemm_lin_kernel<1, 0>), grid, dim3(1024), 0, st, g);
  else hipLaunchKernelGGL((zgemm_lin_kernel<0, 0>), grid, dim3(1024), 0, st, g);
  return hipGetLastError();
}

}  // namespace dq
