// Sparse generator with an XOR-regular pattern: the step exponential and its reverse mode column by column in LDS.
//
// The sparse-generator scheme (realpath.hip, engine.hip real_forward / real_backward_spx; the step itself is the reference's
// expm(-1j * dt * H) of sim_plain.py:145-146 / diffqc.cc:192-198) evaluates U = cos X + i sin X with X = -dt H sparse from the
// powers X_{m+1} = X X_m.  Every column of every power is an independent chain of sparse matrix-vector products,
//     x_0 = e_c,  x_{m+1} = X x_m,     U[:, c] = sum_j cc_j x_{2j} + i sum_j sc_j x_{2j+1},
// and when X's pattern is XOR-regular -- besides its diagonal, row i holds exactly the entries (i, i ^ x_j), j < R, the same
// offsets x_j in every row, each of constant magnitude along its XOR-diagonal: sums of Pauli strings, e.g. an Ising drift with
// X_k controls -- the column needs no index arrays at all: the vector lives in LDS (d * sizeof(T) bytes, two images), a thread
// owns the RPT rows r * NT + t, the partner of row i under offset x_j is a conflict-free LDS read at i ^ x_j, the weight
// w_j = -dt u[kid_j] cmag_j is wave-uniform (a sign bit per row and diagonal where entries are negative), and all diagonals at
// offset 0 (the drift's, Z-type controls') collapse into one per-row weight.  HBM traffic of a whole step exponential = U
// written once (the dense powers, the dense X and the linear-combination pass no longer exist); the kernel is bound by LDS
// reads (R per row and power).
//
// Reverse mode (xr_reverse_kernel), per column c with (cbar | sbar) = (Re | Im) of ROW c of G = Ubar (the cotangents enter
// through the transposed matrices: X and the controls are symmetric, so <Xi^T, H_k> = <Xi, H_k>):
//     xbar_{2q+1} = sc_q sbar,   xbar_m = coef_m (cbar | sbar) + X xbar_{m+1}   (m = 2q .. 1),
//     dL/du_k    = -dt sum_{j: kid_j = k} sum_i val_ij sum_{m=0}^{2q} xbar_{m+1}[i ^ x_j] x_m[i]
// -- the partner value fetched for the product is the one the dot product needs (as in rsp_mm_dot_kernel); x_2 .. x_2q of the
// column are recomputed first and parked in a per-workgroup scratch that stays in L2, so from HBM nothing but G is read and a
// few sums per workgroup are written.  Results differ from the row-wise kernels by rounding only (different summation order).
#include "dq_internal.h"

namespace dq {

__device__ inline double xr_flip(double v, unsigned bit) {
  return __longlong_as_double(__double_as_longlong(v) ^ ((long long)bit << 63));
}
__device__ inline float xr_flip(float v, unsigned bit) { return __int_as_float(__float_as_int(v) ^ (int)(bit << 31)); }
__device__ inline double xr_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ inline float xr_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// per-thread constants of a launch: the weights of the XOR-diagonals (wave-uniform), the rows' sign words and diagonal weights;
// in LDS: jof[x] = the (last) diagonal whose offset is x (R for x = 0, R + 1 for none), nxt[j] = the next one with the same
// offset (>= R: none), and the weights
template <class T, int NT, int RPT, bool SIGN>
struct XrRows {
  T dg[RPT];  // sum over the offset-0 diagonals of -dt u val
  unsigned sg[RPT];
  int R;
  const unsigned char *jof, *nxt;
  const T* Wl;
  __device__ __forceinline__ void load(const XrTab& tab, const double* __restrict__ uz, double mdt, int t, unsigned char* s_jof,
                                       T* s_w) {
    unsigned char* s_nxt = (unsigned char*)(s_w + XR_RMAX + 2);
    constexpr int D = NT * RPT;
    R = tab.R;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      sg[r] = SIGN ? tab.sgn[r * NT + t] : 0u;
      dg[r] = 0;
    }
    for (int j = 0; j < tab.nd; ++j) {
      const int k = tab.dkid[j];
      const double w = mdt * (k < 0 ? 1.0 : uz[k]);
#pragma unroll
      for (int r = 0; r < RPT; ++r) dg[r] += (T)(w * tab.dval[(long)j * D + r * NT + t]);
    }
    for (int i = t; i < D; i += NT) s_jof[i] = (unsigned char)(tab.R + 1);
    __syncthreads();
    if (t < tab.R) {
      const int k = tab.kid[t];
      s_w[t] = (T)(mdt * (k < 0 ? 1.0 : uz[k]) * tab.cmag[t]);
    }
    if (t == 0) {  // diagonals that share an offset (X_a and X_a Z_b, say) are chained: jof -> the last one, nxt -> the others
      for (int j = 0; j < tab.R; ++j) {
        s_nxt[j] = s_jof[tab.xoff[j]];
        s_jof[tab.xoff[j]] = (unsigned char)j;
      }
      s_jof[0] = (unsigned char)tab.R;
      s_w[tab.R] = 0;
      s_w[tab.R + 1] = 0;
    }
    __syncthreads();
    jof = s_jof;
    nxt = s_nxt;
    Wl = s_w;
  }
  // x_1 = X e_c (column c of X) at the thread's rows: the diagonal weight at row c, +-w_j at the rows c ^ x_j
  __device__ __forceinline__ void first(int t, int c, T* x1) const {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int i = r * NT + t;
      T w = 0;
      for (unsigned j = jof[i ^ c]; j < (unsigned)R; j = nxt[j]) w += SIGN ? xr_flip(Wl[j], (sg[r] >> j) & 1u) : Wl[j];
      x1[r] = i == c ? dg[r] : w;
    }
  }
};

// dacc[j] += s for a wave-uniform j (register arrays need static indices: a binary tree of scalar compares)
template <int LO, int HI>
__device__ __forceinline__ void xr_acc(double* dacc, int j, double s) {
  if constexpr (HI - LO == 1) {
    asm volatile("" : "+v"(dacc[LO]));  // (keeps the tree from being turned into an indexed access to a scratch array)
    dacc[LO] += s;
  } else {
    constexpr int MID = (LO + HI) / 2;
    if (j < MID) xr_acc<LO, MID>(dacc, j, s);
    else xr_acc<MID, HI>(dacc, j, s);
  }
}

// one sweep  o += (X - diag) v  with v in LDS image `src`; optional dot products with xm[] (DOT).  The loop over the
// diagonals is a runtime loop on purpose: unrolled, the compiler hoists sign masks and LDS addresses of all R diagonals out of
// the column loop, prefetches across diagonals and spills hundreds of registers.
template <class T, int NT, int RPT, bool SIGN, bool DOT, bool PAIR>
__device__ __forceinline__ void xr_gather(const XrTab& tab, const XrRows<T, NT, RPT, SIGN>& rw, const T* __restrict__ src, int t, T* o,
                                          const T* xm, double* dacc) {
  // PAIR: two diagonals per trip (16 LDS reads in flight per lane) -- not where that spills (reverse kernel with sign words)
  // (volatile reads: hipcc otherwise merges neighbours into ds_read2st64_b64, which the LDS array serves at half the rate of
  // ds_read_b64 -- 128 against 256 B/clk/CU -- and this loop lives on that rate)
  auto fetch = [&](int j, T* vs) {
    const int xo = tab.xoff[j];
    typedef const volatile __attribute__((address_space(3))) T* lds_ptr;
    lds_ptr base = (lds_ptr)(src + (t ^ (xo & (NT - 1))));
    if (xo < NT) {  // (wave-uniform) the partner rows keep their place in the thread's row list: immediate offsets
#pragma unroll
      for (int r = 0; r < RPT; ++r) vs[r] = base[r * NT];
    } else {
      const int xh = xo / NT;
#pragma unroll
      for (int r = 0; r < RPT; ++r) vs[r] = base[(r ^ xh) * NT];
    }
  };
  auto use = [&](int j, T* vs) {
    if (SIGN) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) vs[r] = xr_flip(vs[r], (rw.sg[r] >> j) & 1u);
    }
    const T w = rw.Wl[j];
#pragma unroll
    for (int r = 0; r < RPT; ++r) o[r] = xr_fma(w, vs[r], o[r]);
    if (DOT) {
      T s = 0;
#pragma unroll
      for (int r = 0; r < RPT; ++r) s = xr_fma(vs[r], xm[r], s);
      xr_acc<0, XR_RMAX>(dacc, j, (double)s);
    }
  };
  int j = 0;
#pragma clang loop unroll(disable)
  for (; PAIR && j + 1 < tab.R; j += 2) {
    T va[RPT], vb[RPT];
    fetch(j, va);
    fetch(j + 1, vb);
    use(j, va);
    use(j + 1, vb);
  }
#pragma clang loop unroll(disable)
  for (; j < tab.R; ++j) {
    T va[RPT];
    fetch(j, va);
    use(j, va);
  }
}

// level 0 of the gradient, by lane 0 of the workgroup: xbar_1 (LDS image `src`) pairs with x_0 = e_c, i.e. the entries
// (c, c ^ x_j) -- sign taken at row c ^ x_j (the diagonals are symmetric)
template <class T, int J>
__device__ __forceinline__ void xr_level0(const XrTab& tab, const T* __restrict__ src, int c, double* dacc) {
  if constexpr (J < XR_RMAX) {
    if (J < tab.R) {
      const int row = c ^ tab.xoff[J];
      const unsigned bit = tab.sgn ? (tab.sgn[row] >> J) & 1u : 0u;
      dacc[J] += (double)xr_flip(src[row], bit);
    }
    xr_level0<T, J + 1>(tab, src, c, dacc);
  }
}

// LDS of both kernels: `images` vectors of D, then jof (D bytes), then XR_RMAX + 2 weights, then nxt (XR_RMAX bytes)
template <class T, int D>
constexpr size_t xr_lds_bytes(int images) {
  return (size_t)images * sizeof(T) * D + (size_t)D + sizeof(T) * (XR_RMAX + 2) + XR_RMAX;
}

// U[z][c][:] for the columns c = blockIdx.x, blockIdx.x + gridDim.x, ...   (U symmetric: row c = column c)
template <class T, int NT, int RPT, bool SIGN>
__global__ void __launch_bounds__(NT, (NT * 2 + 255) / 256) xr_forward_kernel(typename Cx<T>::type* __restrict__ U, long stride, XrTab tab,
                                                        const double* __restrict__ u, int K, double mdt, XrPoly pol) {
  typedef typename Cx<T>::type T2;
  constexpr int D = NT * RPT;
  extern __shared__ __align__(16) unsigned char xr_lds_raw[];
  T* lds = (T*)xr_lds_raw;  // two images of D
  const int t = threadIdx.x;
  const long z = blockIdx.y;
  const int np = 2 * pol.q + 1;
  XrRows<T, NT, RPT, SIGN> rw;
  rw.load(tab, u + z * K, mdt, t, xr_lds_raw + 2 * sizeof(T) * D, (T*)(xr_lds_raw + 2 * sizeof(T) * D + D));
  int p = 1;  // image read last
  for (int c = blockIdx.x; c < D; c += gridDim.x) {
    T cur[RPT], are[RPT], aim[RPT];
    rw.first(t, c, cur);
    p ^= 1;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      lds[p * D + r * NT + t] = cur[r];
      are[r] = (r * NT + t) == c ? (T)pol.cc[0] : (T)0;
      aim[r] = (T)pol.sc[0] * cur[r];
    }
    __syncthreads();
    for (int m = 2; m <= np; ++m) {
      T o[RPT];
#pragma unroll
      for (int r = 0; r < RPT; ++r) o[r] = rw.dg[r] * cur[r];
      xr_gather<T, NT, RPT, SIGN, false, true>(tab, rw, lds + p * D, t, o, nullptr, nullptr);
      const T cf = (T)((m & 1) ? pol.sc[(m - 1) / 2] : pol.cc[m / 2]);
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        cur[r] = o[r];
        if (m & 1) aim[r] = xr_fma(cf, o[r], aim[r]);
        else are[r] = xr_fma(cf, o[r], are[r]);
      }
      if (m < np) {
        p ^= 1;
#pragma unroll
        for (int r = 0; r < RPT; ++r) lds[p * D + r * NT + t] = o[r];
        __syncthreads();
      }
    }
    T2* urow = U + z * stride + (long)c * D;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      T2 v;
      v.x = are[r];
      v.y = aim[r];
      urow[r * NT + t] = v;
    }
  }
}

// part[(z * gridDim.x + blockIdx.x) * XR_SLOTS + s]: this workgroup's share, over its columns, of
//   s < XR_RMAX:  sum_i sign_ij Xi[i ^ x_j, i]  (before the factors cmag_j and -dt);   s = XR_RMAX + j: the offset-0 diagonal j,
//   sum_i dval_j[i] Xi[i, i]  (DIAGCTL: some control has diagonal entries).
// The column's powers x_2 .. x_2q are recomputed first and parked in a scratch buffer in device memory (xpow: per workgroup
// 2q - 1 vectors of D, each element written by the thread that reads it back: no synchronisation; they live in L2 / Infinity
// Cache) -- in registers they cost 2 (2q - 1) RPT VGPRs, pin the kernel to 8 waves per CU and q to a template parameter
// (measured: the same speed).  x_1 is rebuilt from the lookup table.  LDS: four images (the two of the sweeps, cbar, sbar);
// NT = D / 4 threads (16 waves per CU at D = 4096), one workgroup per CU.
template <class T, int NT, int RPT, bool SIGN, bool DIAGCTL>
__global__ void __launch_bounds__(NT) xr_reverse_kernel(const typename Cx<T>::type* __restrict__ G, long stride, XrTab tab,
                                                               const double* __restrict__ u, int K, double mdt, XrPoly pol,
                                                               double* __restrict__ part, T* __restrict__ xpow) {
  typedef typename Cx<T>::type T2;
  constexpr int D = NT * RPT;
  extern __shared__ __align__(16) unsigned char xr_lds_raw[];
  T* lds = (T*)xr_lds_raw;
  T* lcb = lds + 2 * D;
  T* lsb = lds + 3 * D;
  const int t = threadIdx.x;
  const long z = blockIdx.y;
  const int npm1 = 2 * pol.q;
  XrRows<T, NT, RPT, SIGN> rw;
  rw.load(tab, u + z * K, mdt, t, xr_lds_raw + 4 * sizeof(T) * D, (T*)(xr_lds_raw + 4 * sizeof(T) * D + D));
  T* xp = xpow + (z * gridDim.x + blockIdx.x) * (long)(npm1 - 1) * D + t;  // [m - 2][r * NT + t]
  double dacc[XR_RMAX], rowacc[RPT];
#pragma unroll
  for (int j = 0; j < XR_RMAX; ++j) dacc[j] = 0.0;
#pragma unroll
  for (int r = 0; r < RPT; ++r) rowacc[r] = 0.0;
  int p = 1;
  for (int c = blockIdx.x; c < D; c += gridDim.x) {
    const T2* grow = G + z * stride + (long)c * D;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const T2 g = grow[r * NT + t];
      lcb[r * NT + t] = g.x;
      lsb[r * NT + t] = g.y;
    }
    T cur[RPT];
    rw.first(t, c, cur);
    p ^= 1;
#pragma unroll
    for (int r = 0; r < RPT; ++r) lds[p * D + r * NT + t] = cur[r];
    __syncthreads();
    for (int m = 2; m <= npm1; ++m) {
      T o[RPT];
#pragma unroll
      for (int r = 0; r < RPT; ++r) o[r] = rw.dg[r] * cur[r];
      xr_gather<T, NT, RPT, SIGN, false, true>(tab, rw, lds + p * D, t, o, nullptr, nullptr);
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        cur[r] = o[r];
        xp[(long)(m - 2) * D + r * NT] = o[r];
      }
      if (m < npm1) {
        p ^= 1;
#pragma unroll
        for (int r = 0; r < RPT; ++r) lds[p * D + r * NT + t] = o[r];
        __syncthreads();
      }
    }
    p ^= 1;
    {
      const T cf = (T)pol.sc[pol.q];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        cur[r] = cf * lsb[r * NT + t];
        lds[p * D + r * NT + t] = cur[r];
      }
    }
    __syncthreads();
    for (int m = npm1; m >= 1; --m) {
      T o[RPT], xmm[RPT];
      if (m == 1) {
        rw.first(t, c, xmm);
      } else {
#pragma unroll
        for (int r = 0; r < RPT; ++r) xmm[r] = xp[(long)(m - 2) * D + r * NT];
      }
      const T cf = (T)((m & 1) ? pol.sc[(m - 1) / 2] : pol.cc[m / 2]);
      const T* add = (m & 1) ? lsb : lcb;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        o[r] = xr_fma(cf, add[r * NT + t], rw.dg[r] * cur[r]);
        if (DIAGCTL) rowacc[r] = __builtin_fma((double)cur[r], (double)xmm[r], rowacc[r]);
      }
      xr_gather<T, NT, RPT, SIGN, true, !DIAGCTL>(tab, rw, lds + p * D, t, o, xmm, dacc);
      p ^= 1;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        cur[r] = o[r];
        lds[p * D + r * NT + t] = o[r];
      }
      __syncthreads();
    }
    if (t == 0) xr_level0<T, 0>(tab, lds + p * D, c, dacc);
    if (DIAGCTL) {
#pragma unroll
      for (int r = 0; r < RPT; ++r)
        if ((r * NT + t) == c) rowacc[r] += (double)cur[r];
    }
  }
  __syncthreads();
  double* red = (double*)xr_lds_raw;  // [NT / 64][XR_SLOTS]
  const int lane = t & 63, wave = t >> 6;
  auto put = [&](int slot, double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave * XR_SLOTS + slot] = v;
  };
#pragma unroll
  for (int j = 0; j < XR_RMAX; ++j) put(j, dacc[j]);
  for (int j = 0; j < XR_DMAX; ++j) {
    double v = 0.0;
    if (DIAGCTL && j < tab.nd && tab.dkid[j] >= 0) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) v = __builtin_fma((double)tab.dval[(long)j * D + r * NT + t], rowacc[r], v);
    }
    put(XR_RMAX + j, v);
  }
  __syncthreads();
  if (t < XR_SLOTS) {
    double s = 0.0;
    for (int w = 0; w < NT / 64; ++w) s += red[w * XR_SLOTS + t];
    part[(z * gridDim.x + blockIdx.x) * XR_SLOTS + t] = s;
  }
}

// grad[z][k] (+)= scale * (sum_j [kid_j == k] cmag_j sum_b part[z][b][j] + sum_j [dkid_j == k] sum_b part[z][b][XR_RMAX + j])
// One workgroup per step: 32 lanes per slot, lane l adds the workgroups l, l + 32, ... in order, then a fixed butterfly.
__global__ void __launch_bounds__(XR_SLOTS * 32) xr_finish_kernel(const double* __restrict__ part, int nblk, XrTab tab, int K,
                                                                  double scale, double* __restrict__ grad, int accumulate) {
  const long z = blockIdx.x;
  const int slot = threadIdx.x >> 5, l = threadIdx.x & 31;
  __shared__ double s_j[XR_SLOTS];
  const bool live = slot < XR_RMAX ? slot < tab.R : (slot - XR_RMAX < tab.nd && tab.dkid[slot - XR_RMAX] >= 0);
  double s = 0.0;
  if (live)
    for (int b = l; b < nblk; b += 32) s += part[(z * nblk + b) * XR_SLOTS + slot];
  for (int off = 16; off > 0; off >>= 1) s += __shfl_down(s, off, 32);
  if (l == 0) s_j[slot] = live ? s * (slot < XR_RMAX ? tab.cmag[slot] : 1.0) : 0.0;
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += XR_SLOTS * 32) {
    double g = accumulate ? grad[z * K + k] : 0.0;
    for (int q = 0; q < tab.R; ++q)
      if (tab.kid[q] == k) g += scale * s_j[q];
    for (int q = 0; q < tab.nd; ++q)
      if (tab.dkid[q] == k) g += scale * s_j[XR_RMAX + q];
    grad[z * K + k] = g;
  }
}

bool xr_supported(int d, int q, int R, int nd, size_t elem) {
  if (elem != sizeof(double) && elem != sizeof(float)) return false;
  if (R < 0 || R > XR_RMAX || nd < 0 || nd > XR_DMAX || q < 1 || q > XR_QMAX) return false;
  return d == 256 || d == 512 || d == 1024 || d == 2048 || d == 4096;
}

// workgroups per step: 512 for a single step (two per CU), fewer when many steps are stacked in one launch (about 4096
// workgroups in all), so that a workgroup's set-up (diagonal weights, lookup table) is spread over many columns
int xr_blocks(int d, int batch) {
  int nb = 512;
  while (nb > 1 && (long)nb * batch > 4096) nb >>= 1;
  return nb < d ? nb : d;
}

template <class T, int NT, int RPT>
static hipError_t xr_forward_go(typename Cx<T>::type* U, long stride, const XrTab& tab, const double* u, int K, double mdt,
                                const XrPoly& pol, int batch, hipStream_t st) {
  const size_t lds = xr_lds_bytes<T, NT * RPT>(2);
  const dim3 grid((unsigned)xr_blocks(NT * RPT, batch), (unsigned)batch);
  void (*kf)(typename Cx<T>::type*, long, XrTab, const double*, int, double, XrPoly) =
      tab.sgn ? xr_forward_kernel<T, NT, RPT, true> : xr_forward_kernel<T, NT, RPT, false>;
  (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kf, grid, dim3(NT), lds, st, U, stride, tab, u, K, mdt, pol);
  return hipGetLastError();
}

// U[z] = cos X_z + i sin X_z for `batch` stacked steps (X_z = -dt H(u[z])), U row-major d x d complex, stride elements apart
template <class T>
static hipError_t xr_forward_t(typename Cx<T>::type* U, long stride, int d, const XrTab& tab, const double* u, int K, double dt,
                               const XrPoly& pol, int batch, hipStream_t st) {
  if (batch <= 0) return hipSuccess;
  if (!xr_supported(d, pol.q, tab.R, tab.nd, sizeof(T))) return hipErrorInvalidValue;
  switch (d) {
    case 4096: return xr_forward_go<T, 512, 8>(U, stride, tab, u, K, -dt, pol, batch, st);
    case 2048: return xr_forward_go<T, 256, 8>(U, stride, tab, u, K, -dt, pol, batch, st);
    case 1024: return xr_forward_go<T, 128, 8>(U, stride, tab, u, K, -dt, pol, batch, st);
    case 512: return xr_forward_go<T, 64, 8>(U, stride, tab, u, K, -dt, pol, batch, st);
    default: return xr_forward_go<T, 64, 4>(U, stride, tab, u, K, -dt, pol, batch, st);
  }
}
hipError_t xr_forward_launch(double2* U, long stride, int d, const XrTab& tab, const double* u, int K, double dt,
                             const XrPoly& pol, int batch, hipStream_t st) {
  return xr_forward_t<double>(U, stride, d, tab, u, K, dt, pol, batch, st);
}
hipError_t xr_forward_launch(float2* U, long stride, int d, const XrTab& tab, const double* u, int K, double dt, const XrPoly& pol,
                             int batch, hipStream_t st) {
  return xr_forward_t<float>(U, stride, d, tab, u, K, dt, pol, batch, st);
}

template <class T, int NT, int RPT>
static hipError_t xr_reverse_go(const typename Cx<T>::type* G, long stride, const XrTab& tab, const double* u, int K,
                                       double mdt, const XrPoly& pol, int batch, double* part, const unsigned* zero_sgn, T* xpow,
                                       hipStream_t st) {
  const size_t lds = xr_lds_bytes<T, NT * RPT>(4);
  const dim3 grid((unsigned)xr_blocks(NT * RPT, batch), (unsigned)batch);
  bool full = tab.sgn != nullptr;
  for (int j = 0; j < tab.nd; ++j) full = full || tab.dkid[j] >= 0;
  XrTab tb = tab;
  if (full && !tb.sgn) tb.sgn = zero_sgn;
  if (full && !tb.sgn) return hipErrorInvalidValue;
  void (*kf)(const typename Cx<T>::type*, long, XrTab, const double*, int, double, XrPoly, double*, T*) =
      full ? xr_reverse_kernel<T, NT, RPT, true, true> : xr_reverse_kernel<T, NT, RPT, false, false>;
  (void)hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kf, grid, dim3(NT), lds, st, G, stride, tb, u, K, mdt, pol, part, xpow);
  return hipGetLastError();
}

// scratch of the reverse kernel: the powers x_2 .. x_2q of one column per workgroup and step
size_t xr_pow_elems(int d, int q, int batch) { return (size_t)xr_blocks(d, batch) * (size_t)(2 * q - 1) * (size_t)d; }

// grad[z][k] (+)= dL/du_k of step z given G[z] = Ubar_z; part: batch * xr_blocks(d, batch) * XR_SLOTS doubles of scratch, xpow: batch *
// xr_pow_elems(d, q, batch) doubles of scratch; zero_sgn: d zero words (used when a control has diagonal entries but the table has no
// sign words)
template <class T>
static hipError_t xr_reverse_t(const typename Cx<T>::type* G, long stride, int d, const XrTab& tab, const double* u, int K, double dt,
                               const XrPoly& pol, int batch, double* part, const unsigned* zero_sgn, T* xpow, double* grad,
                               int accumulate, hipStream_t st) {
  if (batch <= 0) return hipSuccess;
  if (!xr_supported(d, pol.q, tab.R, tab.nd, sizeof(T)) || !xpow) return hipErrorInvalidValue;
  hipError_t e;
  switch (d) {
    case 4096: e = xr_reverse_go<T, 1024, 4>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st); break;
    case 2048: e = xr_reverse_go<T, 512, 4>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st); break;
    case 1024: e = xr_reverse_go<T, 256, 4>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st); break;
    case 512: e = xr_reverse_go<T, 128, 4>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st); break;
    default: e = xr_reverse_go<T, 64, 4>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st);
  }
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(xr_finish_kernel, dim3((unsigned)batch), dim3(XR_SLOTS * 32), 0, st, part, xr_blocks(d, batch), tab, K, -dt, grad,
                     accumulate);
  return hipGetLastError();
}
hipError_t xr_reverse_launch(const double2* G, long stride, int d, const XrTab& tab, const double* u, int K, double dt,
                             const XrPoly& pol, int batch, double* part, const unsigned* zero_sgn, double* xpow, double* grad,
                             int accumulate, hipStream_t st) {
  return xr_reverse_t<double>(G, stride, d, tab, u, K, dt, pol, batch, part, zero_sgn, xpow, grad, accumulate, st);
}
hipError_t xr_reverse_launch(const float2* G, long stride, int d, const XrTab& tab, const double* u, int K, double dt,
                             const XrPoly& pol, int batch, double* part, const unsigned* zero_sgn, float* xpow, double* grad,
                             int accumulate, hipStream_t st) {
  return xr_reverse_t<float>(G, stride, d, tab, u, K, dt, pol, batch, part, zero_sgn, xpow, grad, accumulate, st);
}

}  // namespace dq
