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
template <class T, int NT, int RPT, bool SIGN, bool DOT, bool PAIR, int RM = XR_RSMALL>
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
      xr_acc<0, RM>(dacc, j, (double)s);
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
template <class T, int RM, int J>
__device__ __forceinline__ void xr_level0(const XrTab& tab, const T* __restrict__ src, int c, double* dacc) {
  if constexpr (J < RM) {
    if (J < tab.R) {
      const int row = c ^ tab.xoff[J];
      const unsigned bit = tab.sgn ? (tab.sgn[row] >> J) & 1u : 0u;
      dacc[J] += (double)xr_flip(src[row], bit);
    }
    xr_level0<T, RM, J + 1>(tab, src, c, dacc);
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
template <class T, int NT, int RPT, bool SIGN, bool DIAGCTL, int RM>
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
  double dacc[RM], rowacc[RPT];
#pragma unroll
  for (int j = 0; j < RM; ++j) dacc[j] = 0.0;
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
      xr_gather<T, NT, RPT, SIGN, true, !DIAGCTL, RM>(tab, rw, lds + p * D, t, o, xmm, dacc);
      p ^= 1;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        cur[r] = o[r];
        lds[p * D + r * NT + t] = o[r];
      }
      __syncthreads();
    }
    if (t == 0) xr_level0<T, RM, 0>(tab, lds + p * D, c, dacc);
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
  for (int j = 0; j < RM; ++j) put(j, dacc[j]);
  for (int j = RM; j < XR_RMAX; ++j) put(j, 0.0);
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
__global__ void __launch_bounds__(1024) xr_finish_kernel(const double* __restrict__ part, int nblk, XrTab tab, int K, double scale,
                                                         double* __restrict__ grad, int accumulate) {
  const long z = blockIdx.x;
  const int l = threadIdx.x & 31;
  __shared__ double s_j[XR_SLOTS];
  for (int slot = threadIdx.x >> 5; slot < XR_SLOTS; slot += 32) {
    const bool live = slot < XR_RMAX ? slot < tab.R : (slot - XR_RMAX < tab.nd && tab.dkid[slot - XR_RMAX] >= 0);
    double s = 0.0;
    if (live)
      for (int b = l; b < nblk; b += 32) s += part[(z * nblk + b) * XR_SLOTS + slot];
    for (int off = 16; off > 0; off >>= 1) s += __shfl_down(s, off, 32);
    if (l == 0) s_j[slot] = live ? s * (slot < XR_RMAX ? tab.cmag[slot] : 1.0) : 0.0;
  }
  __syncthreads();
  for (int k = threadIdx.x; k < K; k += 1024) {
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
  {
    const hipError_t ea = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;  // (a device or partition mode with less LDS: the caller falls back)
  }
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

template <class T, int NT, int RPT, int RM>
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
      full ? xr_reverse_kernel<T, NT, RPT, true, true, RM> : xr_reverse_kernel<T, NT, RPT, false, false, RM>;
  {
    const hipError_t ea = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;  // (a device or partition mode with less LDS: the caller falls back)
  }
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
  if (tab.R <= XR_RSMALL) {  // 16 accumulators: 4 rows per thread, d / 4 threads
    switch (d) {
      case 4096: e = xr_reverse_go<T, 1024, 4, XR_RSMALL>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st); break;
      case 2048: e = xr_reverse_go<T, 512, 4, XR_RSMALL>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st); break;
      case 1024: e = xr_reverse_go<T, 256, 4, XR_RSMALL>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st); break;
      case 512: e = xr_reverse_go<T, 128, 4, XR_RSMALL>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st); break;
      default: e = xr_reverse_go<T, 64, 4, XR_RSMALL>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st);
    }
  } else {  // up to 32 accumulators (64 VGPRs): 8 rows per thread, d / 8 threads, so that a lane may use 256 registers
    switch (d) {
      case 4096: e = xr_reverse_go<T, 512, 8, XR_RMAX>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st); break;
      case 2048: e = xr_reverse_go<T, 256, 8, XR_RMAX>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st); break;
      case 1024: e = xr_reverse_go<T, 128, 8, XR_RMAX>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st); break;
      case 512: e = xr_reverse_go<T, 64, 8, XR_RMAX>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st); break;
      default: e = xr_reverse_go<T, 64, 4, XR_RMAX>(G, stride, tab, u, K, -dt, pol, batch, part, zero_sgn, xpow, st);
    }
  }
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(xr_finish_kernel, dim3((unsigned)batch), dim3(1024), 0, st, part, xr_blocks(d, batch), tab, K, -dt, grad,
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

// =====================================================================================================================
// Complex Hermitian XOR-regular generators (X and Y drives: H = sum of Pauli strings with an odd number of Y's among them).
// No cos / sin split exists for them; the step is U = exp(A) = sum_{m <= L} A^m / m!, A = -i dt H anti-Hermitian, and the same
// column idea applies with complex vectors (two LDS images of 16 D bytes).  The kernels work in the TRANSPOSED world,
// A~ = A^T (entries A~[i, i ^ x_j] = -i dt conj(H[i, i ^ x_j])):  x_m = A~ x_{m-1} / m from x_0 = e_c gives ROW c of U, which is
// what the chain's products read; and, with g = row c of G = Ubar and H~_k = conj(H_k),
//     dL/du_k = dt Im sum_c sum_{a < L} < y_a , H~_k x_a >,     y_{L-1} = g / L,   y_a = (g - A~ y_{a+1}) / (a + 1)
// (from dL = Re tr(G^H dU), dU = sum_l 1/l! sum_{a+b=l-1} A^a dA A^b, dA = -i dt H_k du_k, transposed under the trace; A~^H = -A~).
// The partner value y_{a+1}[i ^ x_j] fetched for the product is the factor of the level-(a+1) dot product (Hermitian symmetry
// of H~_k moves the conjugate onto it); an XOR-diagonal with entries +-c contributes sigma (v_r x_i - v_i x_r), one with entries
// +-i c contributes sigma (v_r x_r + v_i x_i) -- one real accumulator per diagonal, as in the real kernels.
// Table: XrTab with phase[j]; sign words as in the real case; offset-0 diagonals real and from the drift only.

template <int NT, int RPT, bool SIGN>
struct ZxrRows {
  double dg[RPT];  // dt * sum over the offset-0 diagonals of u val (real)
  unsigned sg[RPT];
  int R;
  const unsigned char *jof, *nxt;
  const double* Wl;  // dt u_j c_j
  const int* ph;
  __device__ __forceinline__ void load(const XrTab& tab, const double* __restrict__ uz, double dt, int t, unsigned char* s_jof,
                                       double* s_w) {
    constexpr int D = NT * RPT;
    unsigned char* s_nxt = (unsigned char*)(s_w + XR_RMAX + 2);
    int* s_ph = (int*)(s_nxt + XR_RMAX);
    R = tab.R;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      sg[r] = SIGN ? tab.sgn[r * NT + t] : 0u;
      dg[r] = 0;
    }
    for (int j = 0; j < tab.nd; ++j) {
      const int k = tab.dkid[j];
      const double w = dt * (k < 0 ? 1.0 : uz[k]);
#pragma unroll
      for (int r = 0; r < RPT; ++r) dg[r] += w * tab.dval[(long)j * D + r * NT + t];
    }
    for (int i = t; i < D; i += NT) s_jof[i] = (unsigned char)(tab.R + 1);
    __syncthreads();
    if (t < XR_RMAX) s_ph[t] = t < tab.R ? tab.phase[t] : 0;
    if (t < tab.R) {
      const int k = tab.kid[t];
      s_w[t] = dt * (k < 0 ? 1.0 : uz[k]) * tab.cmag[t];
    }
    if (t == 0) {
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
    ph = s_ph;
  }
  // x_1 = A~ e_c at the thread's rows: A~[i, c] = -i w sigma (phase 0) or -w sigma (phase 1) where i ^ x_j = c; -i dg at i = c
  __device__ __forceinline__ void first(int t, int c, double2* x1) const {
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      const int i = r * NT + t;
      double2 v;
      v.x = 0;
      v.y = 0;
      for (unsigned j = jof[i ^ c]; j < (unsigned)R; j = nxt[j]) {
        const double w = SIGN ? xr_flip(Wl[j], (sg[r] >> j) & 1u) : Wl[j];
        if (ph[j]) v.x -= w;
        else v.y -= w;
      }
      if (i == c) v.y -= dg[r];
      x1[r] = v;
    }
  }
};

// o += s * (A~ - diag) v  with v in the LDS image `src` (interleaved re, im); DOT: dacc[j] += sigma (v_r x_i - v_i x_r) or
// sigma (v_r x_r + v_i x_i) with x = xm (the thread's own rows)
template <int NT, int RPT, bool SIGN, bool DOT, bool PAIR = false, int RM = XR_RSMALL>
__device__ __forceinline__ void zxr_gather(const XrTab& tab, const ZxrRows<NT, RPT, SIGN>& rw, const double2* __restrict__ src, int t,
                                           double s, double2* o, const double2* xm, double* dacc) {
  typedef double zxr_v2 __attribute__((ext_vector_type(2)));
  typedef const volatile __attribute__((address_space(3))) zxr_v2* lds_ptr;  // one ds_read_b128 per partner value
  auto fetch = [&](int j, double* vr, double* vi) {
    const int xo = tab.xoff[j];
    lds_ptr base = (lds_ptr)(src + (t ^ (xo & (NT - 1))));
    if (xo < NT) {  // (wave-uniform) immediate offsets
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const zxr_v2 v = base[r * NT];
        vr[r] = v.x;
        vi[r] = v.y;
      }
    } else {
      const int xh = xo / NT;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const zxr_v2 v = base[(r ^ xh) * NT];
        vr[r] = v.x;
        vi[r] = v.y;
      }
    }
  };
  auto use = [&](int j, double* vr, double* vi) {
    const double w = s * rw.Wl[j];
    const int phj = rw.ph[j];
    if (SIGN) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const unsigned b = (rw.sg[r] >> j) & 1u;
        vr[r] = xr_flip(vr[r], b);
        vi[r] = xr_flip(vi[r], b);
      }
    }
    if (phj) {  // entries -w sigma
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        o[r].x = __builtin_fma(-w, vr[r], o[r].x);
        o[r].y = __builtin_fma(-w, vi[r], o[r].y);
      }
    } else {  // entries -i w sigma
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        o[r].x = __builtin_fma(w, vi[r], o[r].x);
        o[r].y = __builtin_fma(-w, vr[r], o[r].y);
      }
    }
    if (DOT) {
      double d = 0;
      if (phj) {
#pragma unroll
        for (int r = 0; r < RPT; ++r) d = __builtin_fma(vr[r], xm[r].x, __builtin_fma(vi[r], xm[r].y, d));
      } else {
#pragma unroll
        for (int r = 0; r < RPT; ++r) d = __builtin_fma(vr[r], xm[r].y, __builtin_fma(-vi[r], xm[r].x, d));
      }
      xr_acc<0, RM>(dacc, j, d);
    }
  };
  int j = 0;
#pragma clang loop unroll(disable)
  for (; PAIR && j + 1 < rw.R; j += 2) {
    double ar[RPT], ai[RPT], br[RPT], bi[RPT];
    fetch(j, ar, ai);
    fetch(j + 1, br, bi);
    use(j, ar, ai);
    use(j + 1, br, bi);
  }
#pragma clang loop unroll(disable)
  for (; j < rw.R; ++j) {
    double ar[RPT], ai[RPT];
    fetch(j, ar, ai);
    use(j, ar, ai);
  }
}

constexpr size_t zxr_lds_bytes(int D) {
  return (size_t)2 * 16 * D + (size_t)D + 8 * (XR_RMAX + 2) + XR_RMAX + 4 * XR_RMAX;
}

// U[z][c][:] = row c of exp(A_z), rows c = blockIdx.x, blockIdx.x + gridDim.x, ...
template <int NT, int RPT, bool SIGN>
__global__ void __launch_bounds__(NT) zxr_forward_kernel(double2* __restrict__ U, long stride, XrTab tab, const double* __restrict__ u,
                                                         int K, double dt, int L) {
  constexpr int D = NT * RPT;
  extern __shared__ __align__(16) unsigned char xr_lds_raw[];
  double2* lds = (double2*)xr_lds_raw;  // two images of D
  const int t = threadIdx.x;
  const long z = blockIdx.y;
  ZxrRows<NT, RPT, SIGN> rw;
  rw.load(tab, u + z * K, dt, t, xr_lds_raw + 32 * D, (double*)(xr_lds_raw + 32 * D + D));
  int p = 1;
  for (int c = blockIdx.x; c < D; c += gridDim.x) {
    double2 cur[RPT], acc[RPT];
    rw.first(t, c, cur);
    p ^= 1;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      lds[p * D + r * NT + t] = cur[r];
      acc[r].x = cur[r].x + ((r * NT + t) == c ? 1.0 : 0.0);
      acc[r].y = cur[r].y;
    }
    __syncthreads();
    for (int m = 2; m <= L; ++m) {
      double2 o[RPT];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {  // the diagonal: -i dg v
        o[r].x = rw.dg[r] * cur[r].y;
        o[r].y = -rw.dg[r] * cur[r].x;
      }
      zxr_gather<NT, RPT, SIGN, false, true>(tab, rw, lds + p * D, t, 1.0, o, nullptr, nullptr);
      const double inv = 1.0 / m;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        o[r].x *= inv;
        o[r].y *= inv;
        cur[r] = o[r];
        acc[r].x += o[r].x;
        acc[r].y += o[r].y;
      }
      if (m < L) {
        p ^= 1;
#pragma unroll
        for (int r = 0; r < RPT; ++r) lds[p * D + r * NT + t] = o[r];
        __syncthreads();
      }
    }
    double2* urow = U + z * stride + (long)c * D;
#pragma unroll
    for (int r = 0; r < RPT; ++r) urow[r * NT + t] = acc[r];
  }
}

// part[(z * gridDim.x + blockIdx.x) * XR_SLOTS + j] = this workgroup's share of sum_c sum_a of the level dot products of
// XOR-diagonal j (before the factors cmag_j and dt); xpow: per workgroup the powers x_1 .. x_{L-1} of the current row
template <int NT, int RPT, bool SIGN, bool DIAGCTL, int RM>
__global__ void __launch_bounds__(NT) zxr_reverse_kernel(const double2* __restrict__ G, long stride, XrTab tab,
                                                         const double* __restrict__ u, int K, double dt, int L,
                                                         double* __restrict__ part, double2* __restrict__ xpow) {
  constexpr int D = NT * RPT;
  extern __shared__ __align__(16) unsigned char xr_lds_raw[];
  double2* lds = (double2*)xr_lds_raw;
  const int t = threadIdx.x;
  const long z = blockIdx.y;
  ZxrRows<NT, RPT, SIGN> rw;
  rw.load(tab, u + z * K, dt, t, xr_lds_raw + 32 * D, (double*)(xr_lds_raw + 32 * D + D));
  double2* xp = xpow + (z * gridDim.x + blockIdx.x) * (long)(L - 1) * D + t;  // [a - 1][r * NT + t], a = 1 .. L - 1
  double dacc[RM], rowacc[DIAGCTL ? RPT : 1];  // rowacc: sum over levels and rows c of Im(conj(y_a[i]) x_a[i]) (diagonal controls)
#pragma unroll
  for (int j = 0; j < RM; ++j) dacc[j] = 0.0;
#pragma unroll
  for (int r = 0; r < (DIAGCTL ? RPT : 1); ++r) rowacc[r] = 0.0;
  int p = 1;
  for (int c = blockIdx.x; c < D; c += gridDim.x) {
    const double2* grow = G + z * stride + (long)c * D + t;
    // x_1 .. x_{L-1}
    double2 cur[RPT];
    rw.first(t, c, cur);
    p ^= 1;
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
      lds[p * D + r * NT + t] = cur[r];
      if (L > 1) xp[r * NT] = cur[r];
    }
    __syncthreads();
    for (int m = 2; m < L; ++m) {
      double2 o[RPT];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        o[r].x = rw.dg[r] * cur[r].y;
        o[r].y = -rw.dg[r] * cur[r].x;
      }
      zxr_gather<NT, RPT, SIGN, false, true>(tab, rw, lds + p * D, t, 1.0, o, nullptr, nullptr);
      const double inv = 1.0 / m;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        o[r].x *= inv;
        o[r].y *= inv;
        cur[r] = o[r];
        xp[(long)(m - 1) * D + r * NT] = o[r];
      }
      if (m < L - 1) {
        p ^= 1;
#pragma unroll
        for (int r = 0; r < RPT; ++r) lds[p * D + r * NT + t] = o[r];
        __syncthreads();
      }
    }
    // y_{L-1} = g / L
    p ^= 1;
    {
      const double inv = 1.0 / L;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        const double2 g = grow[r * NT];
        cur[r].x = g.x * inv;
        cur[r].y = g.y * inv;
        lds[p * D + r * NT + t] = cur[r];
      }
    }
    __syncthreads();
    // the sweep that forms y_{a-1} from y_a pairs y_a with x_a (a = L-1 .. 1); y_0 is left in image p
    for (int a = L - 1; a >= 1; --a) {
      double2 o[RPT], xm[RPT];
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        xm[r] = xp[(long)(a - 1) * D + r * NT];
        const double2 g = grow[r * NT];
        o[r].x = g.x - rw.dg[r] * cur[r].y;   // g - (-i dg y)
        o[r].y = g.y + rw.dg[r] * cur[r].x;
        if (DIAGCTL) rowacc[r] += cur[r].x * xm[r].y - cur[r].y * xm[r].x;
      }
      zxr_gather<NT, RPT, SIGN, true, false, RM>(tab, rw, lds + p * D, t, -1.0, o, xm, dacc);
      const double inv = 1.0 / a;
      p ^= 1;
#pragma unroll
      for (int r = 0; r < RPT; ++r) {
        cur[r].x = o[r].x * inv;
        cur[r].y = o[r].y * inv;
        lds[p * D + r * NT + t] = cur[r];
      }
      __syncthreads();
    }
    // level 0: x_0 = e_c pairs with y_0 at the rows c ^ x_j (sign and phase of entry (c, c ^ x_j))
    if (t == 0) {
      for (int j = 0; j < tab.R; ++j) {
        const double2 v = lds[p * D + (c ^ tab.xoff[j])];
        const unsigned b = tab.sgn ? (tab.sgn[c] >> j) & 1u : 0u;
        const double d = tab.phase[j] ? v.x : -v.y;
        xr_acc<0, RM>(dacc, j, b ? -d : d);
      }
    }
    if (DIAGCTL) {  // level 0 of the diagonal controls: x_0 = e_c, Im(conj(y_0[c])) = -Im y_0[c]
#pragma unroll
      for (int r = 0; r < RPT; ++r)
        if ((r * NT + t) == c) rowacc[r] -= cur[r].y;
    }
  }
  __syncthreads();
  double* red = (double*)xr_lds_raw;  // [NT / 64][XR_SLOTS]
  const int lane = t & 63, wave = t >> 6;
#pragma unroll
  for (int j = 0; j < XR_RMAX; ++j) {
    double v = j < RM ? dacc[j < RM ? j : 0] : 0.0;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave * XR_SLOTS + j] = v;
  }
  for (int j = 0; j < XR_DMAX; ++j) {
    double v = 0.0;
    if (DIAGCTL && j < tab.nd && tab.dkid[j] >= 0) {
#pragma unroll
      for (int r = 0; r < RPT; ++r) v = __builtin_fma(tab.dval[(long)j * D + r * NT + t], rowacc[r], v);
    }
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave * XR_SLOTS + XR_RMAX + j] = v;
  }
  __syncthreads();
  if (t < XR_SLOTS) {
    double sum = 0.0;
    for (int w = 0; w < NT / 64; ++w) sum += red[w * XR_SLOTS + t];
    part[(z * gridDim.x + blockIdx.x) * XR_SLOTS + t] = sum;
  }
}

bool zxr_supported(int d, int L, int R, int nd) {
  if (R < 1 || R > XR_RMAX || nd < 0 || nd > XR_DMAX || L < 1 || L > 16) return false;
  return d == 256 || d == 512 || d == 1024 || d == 2048 || d == 4096;
}

size_t zxr_pow_elems(int d, int L, int batch) { return (size_t)xr_blocks(d, batch) * (size_t)(L > 1 ? L - 1 : 1) * (size_t)d; }

template <int NT, int RPT>
static hipError_t zxr_forward_go(double2* U, long stride, const XrTab& tab, const double* u, int K, double dt, int L, int batch,
                                 hipStream_t st) {
  const size_t lds = zxr_lds_bytes(NT * RPT);
  const dim3 grid((unsigned)xr_blocks(NT * RPT, batch), (unsigned)batch);
  void (*kf)(double2*, long, XrTab, const double*, int, double, int) =
      tab.sgn ? zxr_forward_kernel<NT, RPT, true> : zxr_forward_kernel<NT, RPT, false>;
  {
    const hipError_t ea = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;  // (a device or partition mode with less LDS: the caller falls back)
  }
  hipLaunchKernelGGL(kf, grid, dim3(NT), lds, st, U, stride, tab, u, K, dt, L);
  return hipGetLastError();
}

hipError_t zxr_forward_launch(double2* U, long stride, int d, const XrTab& tab, const double* u, int K, double dt, int L, int batch,
                              hipStream_t st) {
  if (batch <= 0) return hipSuccess;
  if (!zxr_supported(d, L, tab.R, tab.nd)) return hipErrorInvalidValue;
  switch (d) {
    case 4096: return zxr_forward_go<512, 8>(U, stride, tab, u, K, dt, L, batch, st);
    case 2048: return zxr_forward_go<256, 8>(U, stride, tab, u, K, dt, L, batch, st);
    case 1024: return zxr_forward_go<128, 8>(U, stride, tab, u, K, dt, L, batch, st);
    case 512: return zxr_forward_go<64, 8>(U, stride, tab, u, K, dt, L, batch, st);
    default: return zxr_forward_go<64, 4>(U, stride, tab, u, K, dt, L, batch, st);
  }
}

template <int NT, int RPT, int RM>
static hipError_t zxr_reverse_go(const double2* G, long stride, const XrTab& tab, const double* u, int K, double dt, int L, int batch,
                                 double* part, double2* xpow, hipStream_t st) {
  const size_t lds = zxr_lds_bytes(NT * RPT);
  const dim3 grid((unsigned)xr_blocks(NT * RPT, batch), (unsigned)batch);
  bool dc = false;
  for (int j = 0; j < tab.nd; ++j) dc = dc || tab.dkid[j] >= 0;
  if (dc && !tab.sgn) return hipErrorInvalidValue;  // (the full variant reads the sign words; complex tables always carry them)
  void (*kf)(const double2*, long, XrTab, const double*, int, double, int, double*, double2*) =
      dc ? zxr_reverse_kernel<NT, RPT, true, true, RM>
         : (tab.sgn ? zxr_reverse_kernel<NT, RPT, true, false, RM> : zxr_reverse_kernel<NT, RPT, false, false, RM>);
  {
    const hipError_t ea = hipFuncSetAttribute((const void*)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (ea != hipSuccess) return ea;  // (a device or partition mode with less LDS: the caller falls back)
  }
  hipLaunchKernelGGL(kf, grid, dim3(NT), lds, st, G, stride, tab, u, K, dt, L, part, xpow);
  return hipGetLastError();
}

hipError_t zxr_reverse_launch(const double2* G, long stride, int d, const XrTab& tab, const double* u, int K, double dt, int L,
                              int batch, double* part, const unsigned* zero_sgn, double2* xpow, double* grad, int accumulate,
                              hipStream_t st) {
  (void)zero_sgn;
  if (batch <= 0) return hipSuccess;
  if (!zxr_supported(d, L, tab.R, tab.nd) || !xpow) return hipErrorInvalidValue;
  hipError_t e;
  if (tab.R <= XR_RSMALL) {
    switch (d) {
      case 4096: e = zxr_reverse_go<1024, 4, XR_RSMALL>(G, stride, tab, u, K, dt, L, batch, part, xpow, st); break;
      case 2048: e = zxr_reverse_go<512, 4, XR_RSMALL>(G, stride, tab, u, K, dt, L, batch, part, xpow, st); break;
      case 1024: e = zxr_reverse_go<256, 4, XR_RSMALL>(G, stride, tab, u, K, dt, L, batch, part, xpow, st); break;
      case 512: e = zxr_reverse_go<128, 4, XR_RSMALL>(G, stride, tab, u, K, dt, L, batch, part, xpow, st); break;
      default: e = zxr_reverse_go<64, 4, XR_RSMALL>(G, stride, tab, u, K, dt, L, batch, part, xpow, st);
    }
  } else {
    switch (d) {
      case 4096: e = zxr_reverse_go<512, 8, XR_RMAX>(G, stride, tab, u, K, dt, L, batch, part, xpow, st); break;
      case 2048: e = zxr_reverse_go<256, 8, XR_RMAX>(G, stride, tab, u, K, dt, L, batch, part, xpow, st); break;
      case 1024: e = zxr_reverse_go<128, 8, XR_RMAX>(G, stride, tab, u, K, dt, L, batch, part, xpow, st); break;
      case 512: e = zxr_reverse_go<64, 8, XR_RMAX>(G, stride, tab, u, K, dt, L, batch, part, xpow, st); break;
      default: e = zxr_reverse_go<64, 4, XR_RMAX>(G, stride, tab, u, K, dt, L, batch, part, xpow, st);
    }
  }
  if (e != hipSuccess) return e;
  // dL/du_k = +dt * sum_j [kid_j == k] cmag_j * (sum of the workgroups' parts)
  hipLaunchKernelGGL(xr_finish_kernel, dim3((unsigned)batch), dim3(1024), 0, st, part, xr_blocks(d, batch), tab, K, dt, grad,
                     accumulate);
  return hipGetLastError();
}

// ---- nonzeros of complex matrices (the complex table is found from them): counts, then (column, re, im) per (source, row) ----
template <class T2>
__global__ void __launch_bounds__(64) zsp_count_kernel(const T2* __restrict__ H, int d, int* __restrict__ cnt) {
  const int i = blockIdx.x, k = blockIdx.y, lane = threadIdx.x;
  const T2* row = H + ((long)k * d + i) * d;
  int n = 0;
  for (int j0 = 0; j0 < d; j0 += 64) {
    const int j = j0 + lane;
    const bool nz = j < d && (row[j].x != 0 || row[j].y != 0);
    n += __popcll(__ballot(nz));
  }
  if (lane == 0) cnt[(long)k * d + i] = n;
}
template <class T2>
__global__ void __launch_bounds__(64) zsp_fill_kernel(const T2* __restrict__ H, int d, const long* __restrict__ base,
                                                      int* __restrict__ col, double* __restrict__ re, double* __restrict__ im) {
  const int i = blockIdx.x, k = blockIdx.y, lane = threadIdx.x;
  const T2* row = H + ((long)k * d + i) * d;
  long at = base[(long)k * d + i];
  for (int j0 = 0; j0 < d; j0 += 64) {
    const int j = j0 + lane;
    T2 v;
    v.x = 0;
    v.y = 0;
    if (j < d) v = row[j];
    const bool nz = v.x != 0 || v.y != 0;
    const unsigned long long m = __ballot(nz);
    if (nz) {
      const long e = at + __popcll(m & ((1ull << lane) - 1ull));
      col[e] = j;
      re[e] = (double)v.x;
      im[e] = (double)v.y;
    }
    at += __popcll(m);
  }
}
hipError_t zsp_count_launch(const double2* H, int d, int K, int* cnt, hipStream_t st) {
  if (d <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(zsp_count_kernel<double2>, dim3((unsigned)d, (unsigned)K), dim3(64), 0, st, H, d, cnt);
  return hipGetLastError();
}
hipError_t zsp_fill_launch(const double2* H, int d, int K, const long* base, int* col, double* re, double* im, hipStream_t st) {
  if (d <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(zsp_fill_kernel<double2>, dim3((unsigned)d, (unsigned)K), dim3(64), 0, st, H, d, base, col, re, im);
  return hipGetLastError();
}
hipError_t zsp_count_launch(const float2* H, int d, int K, int* cnt, hipStream_t st) {
  if (d <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(zsp_count_kernel<float2>, dim3((unsigned)d, (unsigned)K), dim3(64), 0, st, H, d, cnt);
  return hipGetLastError();
}
hipError_t zsp_fill_launch(const float2* H, int d, int K, const long* base, int* col, double* re, double* im, hipStream_t st) {
  if (d <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(zsp_fill_kernel<float2>, dim3((unsigned)d, (unsigned)K), dim3(64), 0, st, H, d, base, col, re, im);
  return hipGetLastError();
}

}  // namespace dq
