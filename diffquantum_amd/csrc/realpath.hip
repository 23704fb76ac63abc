// Streaming kernels of the real-symmetric fast path (see rgemm.hip for when it applies): everything here is
// HBM-bound, one real scalar (8 B fp64 / 4 B fp32) per lane and access, consecutive lanes on consecutive
// elements; the transposing kernels go through a 32x33 LDS tile so both the direct and the transposed access
// are coalesced. Reductions accumulate in fp64 in a fixed order (bitwise reproducible gradients).
//
//   step generator   A = -i dt (H0 + sum_k u_k H_k) = i X,  X = -dt (H0 + sum_k u_k H_k)   (diffqc.cc:192-197)
//   propagator       U = exp(A) = cos X + i sin X = C + i S                                 (diffqc.cc:198)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include "dq_internal.h"

namespace dq {

// ---- Hr = (Re H + Re H^T)/2 ; imag_max = max |Im H| -----------------------------------------
template <class T>
__global__ void __launch_bounds__(256) real_part_kernel(const typename Cx<T>::type* __restrict__ H, T* __restrict__ Hr,
                                                        int d, unsigned long long* __restrict__ imag_max) {
  typedef typename Cx<T>::type T2;
  __shared__ T tl[32][33];
  const long base = (long)blockIdx.z * d * d;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    T v = 0;
    if (j0 + r < d && i0 + tx < d) v = H[base + (long)(j0 + r) * d + i0 + tx].x;
    tl[r][tx] = v;
  }
  __syncthreads();
  double mx = 0.0;
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, j = j0 + tx;
    if (i < d && j < d) {
      const T2 h = H[base + (long)i * d + j];
      Hr[base + (long)i * d + j] = (T)0.5 * (h.x + tl[tx][r]);
      mx = fmax(mx, fabs((double)h.y));
    }
  }
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_down(mx, off, 64));
  if ((threadIdx.x & 63) == 0 && mx > 0.0) atomicMax(imag_max, (unsigned long long)__double_as_longlong(mx));
}

template <class T>
hipError_t real_part_launch(const typename Cx<T>::type* H, T* Hr, int d, int batch, double* imag_max, hipStream_t st) {
  if (d <= 0 || batch <= 0) return hipSuccess;
  dim3 grid((unsigned)((d + 31) / 32), (unsigned)((d + 31) / 32), (unsigned)batch);
  hipLaunchKernelGGL(real_part_kernel<T>, grid, dim3(256), 0, st, H, Hr, d, (unsigned long long*)imag_max);
  return hipGetLastError();
}
template hipError_t real_part_launch<double>(const double2*, double*, int, int, double*, hipStream_t);
template hipError_t real_part_launch<float>(const float2*, float*, int, int, double*, hipStream_t);

// ---- X[z] = -dt (H0r + sum_k u[z][k] Hsr[k]) ------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) rassemble_kernel(T* __restrict__ X, const T* __restrict__ H0r,
                                                        const T* __restrict__ Hsr, const double* __restrict__ u, int K,
                                                        long n_elem, double dt_) {
  const long z = blockIdx.z;
  const double* uz = u + z * K;
  T* Xz = X + z * n_elem;
  const T mdt = (T)(-dt_);
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_elem; e += (long)gridDim.x * blockDim.x) {
    T h = H0r[e];
    for (int k = 0; k < K; ++k) h = fma((T)uz[k], Hsr[(long)k * n_elem + e], h);
    Xz[e] = mdt * h;
  }
}

template <class T>
hipError_t rassemble_launch(T* X, const T* H0r, const T* Hsr, const double* u, int K, long n_elem, int batch, double dt,
                            hipStream_t st) {
  if (n_elem <= 0 || batch <= 0) return hipSuccess;
  long nb = (n_elem + 255) / 256;
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(rassemble_kernel<T>, dim3((unsigned)nb, 1, (unsigned)batch), dim3(256), 0, st, X, H0r, Hsr, u, K,
                     n_elem, dt);
  return hipGetLastError();
}
template hipError_t rassemble_launch<double>(double*, const double*, const double*, const double*, int, long, int,
                                             double, hipStream_t);
template hipError_t rassemble_launch<float>(float*, const float*, const float*, const double*, int, long, int, double,
                                            hipStream_t);

// ---- out_o = [out_o +] cI_o I + sum_j c_oj x_j ------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) rlincomb_kernel(RLincombArgs<T> a) {
  const long z = blockIdx.z;
  const long n_elem = (long)a.d * a.d;
  const long base = z * a.stride;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_elem; e += (long)gridDim.x * blockDim.x) {
    T v0 = 0, v1 = 0;
    if (a.accumulate) {
      v0 = a.out[0][base + e];
      if (a.n_out > 1) v1 = a.out[1][base + e];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < a.n_in) {
        const T x = a.x[j][base + e];
        v0 = fma((T)a.c[0][j], x, v0);
        v1 = fma((T)a.c[1][j], x, v1);
      }
    if ((e / a.d) == (e % a.d)) {
      v0 += (T)a.cI[0];
      v1 += (T)a.cI[1];
    }
    if (a.pack) {
      typename Cx<T>::type u;
      u.x = v0;
      u.y = v1;
      ((typename Cx<T>::type*)a.out[0])[base + e] = u;  // base counts complex elements here (same stride in elements)
    } else {
      a.out[0][base + e] = v0;
      if (a.n_out > 1) a.out[1][base + e] = v1;
    }
  }
}

template <class T>
hipError_t rlincomb_launch(const RLincombArgs<T>& a, hipStream_t st) {
  const long n_elem = (long)a.d * a.d;
  if (n_elem <= 0 || a.batch <= 0) return hipSuccess;
  long nb = (n_elem + 255) / 256;
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(rlincomb_kernel<T>, dim3((unsigned)nb, 1, (unsigned)a.batch), dim3(256), 0, st, a);
  return hipGetLastError();
}
template hipError_t rlincomb_launch<double>(const RLincombArgs<double>&, hipStream_t);
template hipError_t rlincomb_launch<float>(const RLincombArgs<float>&, hipStream_t);

// ---- U = C + i S ------------------------------------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) pack_u_kernel(typename Cx<T>::type* __restrict__ U, const T* __restrict__ C,
                                                     const T* __restrict__ S, long n) {
  typedef typename Cx<T>::type T2;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    T2 v;
    v.x = C[e];
    v.y = S[e];
    U[e] = v;
  }
}

template <class T>
hipError_t pack_u_launch(typename Cx<T>::type* U, const T* C, const T* S, long n, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  long nb = (n + 255) / 256;
  if (nb > 262144) nb = 262144;
  hipLaunchKernelGGL(pack_u_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, U, C, S, n);
  return hipGetLastError();
}
template hipError_t pack_u_launch<double>(double2*, const double*, const double*, long, hipStream_t);
template hipError_t pack_u_launch<float>(float2*, const float*, const float*, long, hipStream_t);

// ---- Cbar = sym(Re G), Sbar = sym(Im G) ---------------------------------------------------------
// dL = Re tr(G^H dU) with dU = dC + i dS (dC, dS real symmetric) = <Re G, dC> + <Im G, dS>.
template <class T>
__global__ void __launch_bounds__(256) split_g_kernel(const typename Cx<T>::type* __restrict__ G, T* __restrict__ Cb,
                                                      T* __restrict__ Sb, int d, long stride) {
  typedef typename Cx<T>::type T2;
  __shared__ T2 tl[32][33];
  const long base = (long)blockIdx.z * stride;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    T2 v;
    v.x = 0;
    v.y = 0;
    if (j0 + r < d && i0 + tx < d) v = G[base + (long)(j0 + r) * d + i0 + tx];
    tl[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, j = j0 + tx;
    if (i < d && j < d) {
      const long off = base + (long)i * d + j;
      const T2 g = G[off];
      const T2 gt = tl[tx][r];
      Cb[off] = (T)0.5 * (g.x + gt.x);
      Sb[off] = (T)0.5 * (g.y + gt.y);
    }
  }
}

template <class T>
hipError_t split_g_launch(const typename Cx<T>::type* G, T* Cbar, T* Sbar, int d, long stride, int batch, hipStream_t st) {
  if (d <= 0 || batch <= 0) return hipSuccess;
  dim3 grid((unsigned)((d + 31) / 32), (unsigned)((d + 31) / 32), (unsigned)batch);
  hipLaunchKernelGGL(split_g_kernel<T>, grid, dim3(256), 0, st, G, Cbar, Sbar, d, stride);
  return hipGetLastError();
}
template hipError_t split_g_launch<double>(const double2*, double*, double*, int, long, int, hipStream_t);
template hipError_t split_g_launch<float>(const float2*, float*, float*, int, long, int, hipStream_t);

// ---- grad[z][k] = scale * <Xbar[z], Hsr[k]> -------------------------------------------------------
constexpr int RKC = 16;  // controls per pass (one pass over Xbar for K <= 16)
constexpr int RGC_THREADS = 256;

template <class T>
__global__ void __launch_bounds__(RGC_THREADS) rgrad_contract_kernel(const T* __restrict__ Xbar, long strideX,
                                                                     const T* __restrict__ Hsr, int K, int k0,
                                                                     long n_elem, double* __restrict__ part) {
  const long z = blockIdx.z;
  const T* Xz = Xbar + z * strideX;
  double acc[RKC];
#pragma unroll
  for (int k = 0; k < RKC; ++k) acc[k] = 0.0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_elem; e += (long)gridDim.x * blockDim.x) {
    const double x = (double)Xz[e];
#pragma unroll
    for (int k = 0; k < RKC; ++k)
      if (k0 + k < K) acc[k] += x * (double)Hsr[(long)(k0 + k) * n_elem + e];
  }
  __shared__ double red[RKC][RGC_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < RKC; ++k) {
    double v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[k][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < RKC && k0 + (int)threadIdx.x < K) {
    double v = 0.0;
    for (int w = 0; w < RGC_THREADS / 64; ++w) v += red[threadIdx.x][w];
    part[(z * gridDim.x + blockIdx.x) * K + k0 + threadIdx.x] = v;
  }
}

template <class T>
hipError_t rgrad_contract_launch(const T* Xbar, long strideX, const T* Hsr, int K, long n_elem, int batch, double scale,
                                 double* part, double* grad, hipStream_t st) {
  if (K <= 0 || batch <= 0) return hipSuccess;
  const int nb = grad_contract_blocks(n_elem);
  dim3 grid((unsigned)nb, 1, (unsigned)batch);
  for (int k0 = 0; k0 < K; k0 += RKC)
    hipLaunchKernelGGL(rgrad_contract_kernel<T>, grid, dim3(RGC_THREADS), 0, st, Xbar, strideX, Hsr, K, k0, n_elem, part);
  return grad_finish_launch(part, nb, K, batch, scale, grad, st);
}
template hipError_t rgrad_contract_launch<double>(const double*, long, const double*, int, long, int, double, double*,
                                                  double*, hipStream_t);
template hipError_t rgrad_contract_launch<float>(const float*, long, const float*, int, long, int, double, double*,
                                                 double*, hipStream_t);

// ---- sparse control Hamiltonians ----------------------------------------------------------------------------------
// Control Hamiltonians are usually sparse (local Paulis: X_k has d nonzeros out of d^2; the reference's own problems,
// demo_maxcut.py:36-79, are ZZ and X terms). When they are, three things on the real path need no dense d x d pass:
//   * assembly   X = -dt H0 + sum_k (-dt u_k) H_k           : one dense pass over H0 + a scatter of the nonzeros
//   * the last level of the reverse mode. The cotangent of X is  Xbar = (Sbar P + P Sbar)/2 + (Ybar_1 X + X Ybar_1)
//     (S = X P, Y_1 = X X), two two-product symmetric launches -- but the gradient only needs <Xbar, H_k>, and for
//     symmetric operands  <Sbar P + P Sbar, H>/2 = sum_{(i,j) in nnz(H)} H_ij <Sbar[i,:], P[j,:]>  (likewise for Ybar_1, X):
//     nnz(H) row-by-row dot products instead of 2 x 2 d^3 flops; Xbar itself is never formed
//   * the control contractions  <Xbar, H_k>                 : folded into the same dot products.
// The nonzeros of all controls live in one CSR by row (entries of a row ordered by control, then column).

// cnt[k*d + i] = number of nonzeros in row i of Hsr[k]
template <class T>
__global__ void __launch_bounds__(64) rsp_count_kernel(const T* __restrict__ Hsr, int d, int* __restrict__ cnt) {
  const int i = blockIdx.x, k = blockIdx.y, lane = threadIdx.x;
  const T* row = Hsr + ((long)k * d + i) * d;
  int n = 0;
  for (int j0 = 0; j0 < d; j0 += 64) {
    const int j = j0 + lane;
    const bool nz = j < d && row[j] != (T)0;
    n += __popcll(__ballot(nz));
  }
  if (lane == 0) cnt[(long)k * d + i] = n;
}

// entries of (k, i) written in column order at base[k*d + i]
template <class T>
__global__ void __launch_bounds__(64) rsp_fill_kernel(const T* __restrict__ Hsr, int d, const long* __restrict__ base,
                                                      int* __restrict__ col, int* __restrict__ kid,
                                                      double* __restrict__ val, int kid0) {
  const int i = blockIdx.x, k = blockIdx.y, lane = threadIdx.x;
  const T* row = Hsr + ((long)k * d + i) * d;
  long at = base[(long)k * d + i];
  for (int j0 = 0; j0 < d; j0 += 64) {
    const int j = j0 + lane;
    const T v = j < d ? row[j] : (T)0;
    const unsigned long long m = __ballot(v != (T)0);
    if (v != (T)0) {
      const long e = at + __popcll(m & ((1ull << lane) - 1ull));
      col[e] = j;
      kid[e] = kid0 + k;
      val[e] = (double)v;
    }
    at += __popcll(m);
  }
}

template <class T>
hipError_t rsp_count_launch(const T* Hsr, int d, int K, int* cnt, hipStream_t st) {
  if (d <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(rsp_count_kernel<T>, dim3((unsigned)d, (unsigned)K), dim3(64), 0, st, Hsr, d, cnt);
  return hipGetLastError();
}
template hipError_t rsp_count_launch<double>(const double*, int, int, int*, hipStream_t);
template hipError_t rsp_count_launch<float>(const float*, int, int, int*, hipStream_t);

template <class T>
hipError_t rsp_fill_launch(const T* Hsr, int d, int K, const long* base, int* col, int* kid, double* val, int kid0,
                           hipStream_t st) {
  if (d <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(rsp_fill_kernel<T>, dim3((unsigned)d, (unsigned)K), dim3(64), 0, st, Hsr, d, base, col, kid, val, kid0);
  return hipGetLastError();
}
template hipError_t rsp_fill_launch<double>(const double*, int, int, const long*, int*, int*, double*, int, hipStream_t);
template hipError_t rsp_fill_launch<float>(const float*, int, int, const long*, int*, int*, double*, int, hipStream_t);

// X[z][i][col] += -dt u[z][kid] val (kid < 0: a drift entry, weight -dt val)  for the nonzeros of row i: one thread per
// (row, step), entries in CSR order
// (deterministic even when two controls share an entry)
template <class T>
__global__ void __launch_bounds__(256) rsp_add_kernel(T* __restrict__ X, long strideX, const long* __restrict__ rowptr,
                                                      const int* __restrict__ col, const int* __restrict__ kid,
                                                      const double* __restrict__ val, const double* __restrict__ u, int K,
                                                      int d, double mdt) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d) return;
  const long z = blockIdx.z;
  T* row = X + z * strideX + (long)i * d;
  const double* uz = u + z * K;
  for (long e = rowptr[i]; e < rowptr[i + 1]; ++e) row[col[e]] += (T)(mdt * (kid[e] < 0 ? 1.0 : uz[kid[e]]) * val[e]);
}

template <class T>
hipError_t rsp_add_launch(T* X, long strideX, const long* rowptr, const int* col, const int* kid, const double* val,
                          const double* u, int K, int d, int batch, double dt, hipStream_t st) {
  if (d <= 0 || batch <= 0 || K <= 0) return hipSuccess;
  hipLaunchKernelGGL(rsp_add_kernel<T>, dim3((unsigned)((d + 255) / 256), 1, (unsigned)batch), dim3(256), 0, st, X, strideX,
                     rowptr, col, kid, val, u, K, d, -dt);
  return hipGetLastError();
}
template hipError_t rsp_add_launch<double>(double*, long, const long*, const int*, const int*, const double*, const double*,
                                           int, int, int, double, hipStream_t);
template hipError_t rsp_add_launch<float>(float*, long, const long*, const int*, const int*, const double*, const double*, int,
                                          int, int, double, hipStream_t);

// nzval[z][e] = val[e] * ( <Sb[z][i,:], P[z][j,:]> + 2 <Yb[z][i,:], X[z][j,:]> )   for the nonzeros e = (i, j) of row i.
// One wave per row: a 512-column strip of the row's Sb and Yb entries stays in registers while the partner rows of up to
// RSP_NZ nonzeros stream past it (partner rows come from L2: the rows of one control's partners are shared by neighbours).
constexpr int RSP_NZ = 16, RSP_U = 8, RSP_WAVES = 4;

template <class T>
__global__ void __launch_bounds__(RSP_WAVES * 64) rsp_contract_kernel(const T* __restrict__ Sb, const T* __restrict__ P,
                                                                      const T* __restrict__ Yb, const T* __restrict__ X,
                                                                      long stride, int d, const long* __restrict__ rowptr,
                                                                      const int* __restrict__ col,
                                                                      const double* __restrict__ val, long nnz,
                                                                      double* __restrict__ nzval) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * RSP_WAVES + wave;
  if (i >= d) return;
  const long z = blockIdx.z;
  const T* s_row = Sb + z * stride + (long)i * d;
  const T* y_row = Yb + z * stride + (long)i * d;
  const T* Pz = P + z * stride;
  const T* Xz = X + z * stride;
  const long e_end = rowptr[i + 1];
  for (long e0 = rowptr[i]; e0 < e_end; e0 += RSP_NZ) {
    const int cnt = (int)min((long)RSP_NZ, e_end - e0);
    double acc[RSP_NZ];
#pragma unroll
    for (int q = 0; q < RSP_NZ; ++q) acc[q] = 0.0;
    for (int m0 = 0; m0 < d; m0 += 64 * RSP_U) {
      T s[RSP_U], y[RSP_U];
#pragma unroll
      for (int r = 0; r < RSP_U; ++r) {
        const int m = m0 + r * 64 + lane;
        s[r] = m < d ? s_row[m] : (T)0;
        y[r] = m < d ? (T)2 * y_row[m] : (T)0;
      }
#pragma unroll
      for (int q = 0; q < RSP_NZ; ++q) {
        if (q < cnt) {  // wave-uniform
          const long j = col[e0 + q];
          const T* p_row = Pz + j * d;
          const T* x_row = Xz + j * d;
          T t = 0;
#pragma unroll
          for (int r = 0; r < RSP_U; ++r) {
            const int m = m0 + r * 64 + lane;
            if (m < d) t += s[r] * p_row[m] + y[r] * x_row[m];
          }
          acc[q] += (double)t;
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RSP_NZ; ++q) {
      if (q < cnt) {
        double v = acc[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) nzval[z * nnz + e0 + q] = v * val[e0 + q];
      }
    }
  }
}

// grad[z][k] = scale * sum of nzval[z][e] over the entries of control k (kent[kptr[k] .. kptr[k+1]), fixed order)
__global__ void __launch_bounds__(256) rsp_reduce_kernel(const double* __restrict__ nzval, long nnz,
                                                         const long* __restrict__ kptr, const long* __restrict__ kent, int K,
                                                         double scale, double* __restrict__ grad) {
  const int k = blockIdx.x;
  const long z = blockIdx.z;
  const double* v = nzval + z * nnz;
  double a = 0.0;
  for (long q = kptr[k] + threadIdx.x; q < kptr[k + 1]; q += 256) a += v[kent[q]];
  __shared__ double red[256];
  red[threadIdx.x] = a;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) grad[z * K + k] = scale * red[0];
}

template <class T>
hipError_t rsp_contract_launch(const T* Sb, const T* P, const T* Yb, const T* X, long stride, int d, const long* rowptr,
                               const int* col, const double* val, long nnz, const long* kptr, const long* kent, int K,
                               int batch, double scale, double* nzval, double* grad, hipStream_t st) {
  if (d <= 0 || batch <= 0 || K <= 0) return hipSuccess;
  if (nnz > 0) {
    hipLaunchKernelGGL(rsp_contract_kernel<T>, dim3((unsigned)((d + RSP_WAVES - 1) / RSP_WAVES), 1, (unsigned)batch),
                       dim3(RSP_WAVES * 64), 0, st, Sb, P, Yb, X, stride, d, rowptr, col, val, nnz, nzval);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(rsp_reduce_kernel, dim3((unsigned)K, 1, (unsigned)batch), dim3(256), 0, st, nzval, nnz, kptr, kent, K,
                     scale, grad);
  return hipGetLastError();
}
template hipError_t rsp_contract_launch<double>(const double*, const double*, const double*, const double*, long, int,
                                                const long*, const int*, const double*, long, const long*, const long*, int,
                                                int, double, double*, double*, hipStream_t);
template hipError_t rsp_contract_launch<float>(const float*, const float*, const float*, const float*, long, int, const long*,
                                               const int*, const double*, long, const long*, const long*, int, int, double,
                                               double*, double*, hipStream_t);

// ---- sparse generator: drift AND controls sparse (e.g. a diagonal Ising drift with local-Pauli controls) --------------
// Then X_t itself is sparse (13 nonzeros per row at n = 12) and every product of the polynomial can have X as its left
// factor: the powers X_{m+1} = X X_m (m = 1 .. 2q) are sparse x dense products -- nnz(X) row AXPYs per output row, HBM/L2
// bound -- instead of symmetric-output MFMA products, and cos X = sum cc_j X_{2j}, sin X = sum sc_j X_{2j+1} are linear
// combinations of the stored powers. Reverse mode: Xbar_m = coef_m (Cbar | Sbar) + X Xbar_{m+1} (the same kernel with an
// AXPY epilogue), and the generator's cotangent is only ever contracted with the controls:
//   <Xi, H_k>,  Xi = Xbar_1 + sum_m Xbar_{m+1} X_m   ->   sum over nnz(H_k) of row dot products, level by level.

// out[z][i][:] = ca * Add[z][i][:] + sum_{e in row i} w_e V[z][col_e][:],   w_e = mdt (kid_e < 0 ? 1 : u[z][kid_e]) val_e
constexpr int RSPMM_U = 8, RSPMM_WAVES = 4;

template <class T>
__global__ void __launch_bounds__(RSPMM_WAVES * 64) rsp_mm_kernel(T* __restrict__ out, const T* __restrict__ V,
                                                                  const T* __restrict__ Add, double ca, long stride, int d,
                                                                  const long* __restrict__ rowptr, const int* __restrict__ col,
                                                                  const int* __restrict__ kid, const double* __restrict__ val,
                                                                  const double* __restrict__ u, int K, double mdt) {
  // the row's nonzeros as (column, weight) pairs, staged once per 64 of them (one per lane) and then read wave-uniformly
  __shared__ int s_col[RSPMM_WAVES][64];
  __shared__ double s_w[RSPMM_WAVES][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * RSPMM_WAVES + wave;
  if (i >= d) return;  // (whole waves leave together; no block-level barrier below)
  const long z = blockIdx.z;
  const T* Vz = V + z * stride;
  const double* uz = u + z * K;
  const long e0 = rowptr[i], e1 = rowptr[i + 1];
  T* orow = out + z * stride + (long)i * d;
  for (long eb = e0; eb < e1 || eb == e0; eb += 64) {
    const int cnt = (int)min((long)64, e1 - eb);
    if (lane < cnt) {
      const int k = kid[eb + lane];
      s_col[wave][lane] = col[eb + lane];
      s_w[wave][lane] = mdt * (k < 0 ? 1.0 : uz[k]) * val[eb + lane];
    }
    __builtin_amdgcn_wave_barrier();
    const bool first = eb == e0;
    for (int m0 = 0; m0 < d; m0 += 64 * RSPMM_U) {
      T acc[RSPMM_U];
#pragma unroll
      for (int r = 0; r < RSPMM_U; ++r) {
        const int m = m0 + r * 64 + lane;
        if (!first) acc[r] = m < d ? orow[m] : (T)0;
        else acc[r] = (Add != nullptr && m < d) ? (T)ca * Add[z * stride + (long)i * d + m] : (T)0;
      }
      for (int q = 0; q < cnt; ++q) {
        const int jc = __builtin_amdgcn_readfirstlane(s_col[wave][q]);
        const T w = (T)s_w[wave][q];
        const T* vrow = Vz + (long)jc * d;
#pragma unroll
        for (int r = 0; r < RSPMM_U; ++r) {
          const int m = m0 + r * 64 + lane;
          if (m < d) acc[r] += w * vrow[m];
        }
      }
#pragma unroll
      for (int r = 0; r < RSPMM_U; ++r) {
        const int m = m0 + r * 64 + lane;
        if (m < d) orow[m] = acc[r];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (e1 - eb <= 64) break;
  }
}

// ---- the same pass in XCD-slab form (whole 16-B column blocks: d a multiple of 8 * 64 * (16 / sizeof(T))) ----------------
// The matrix is cut into 8 column slabs, one per XCD (workgroups b and b + 8 share an XCD under the observed round-robin
// placement -- a speed matter only).  One wave per (row, slab): the row's (column, weight) pairs sit one per lane in registers
// and are broadcast with v_readlane (no LDS); per 64-lane column block (16 B per lane) the partner rows of up to RSPS_NQ
// entries are requested together, then summed in CSR order.  A slab's rows are then fetched by ONE XCD only (every partner row
// of a row lies in the same slab), most re-reads are L2 hits, and each wave keeps up to 13 KiB of loads in flight:
// 0.27 -> 0.135 ms per pass at d = 4096, nnz/row = 13 (tools/rspmm_probe.hip).
constexpr int RSPS_NQ = 16;

template <class T>
__global__ void __launch_bounds__(256) rsp_mm_slab_kernel(T* __restrict__ out, const T* __restrict__ V, const T* __restrict__ Add,
                                                          double ca, long stride, int d, const long* __restrict__ rowptr,
                                                          const int* __restrict__ col, const int* __restrict__ kid,
                                                          const double* __restrict__ val, const double* __restrict__ u, int K,
                                                          double mdt, int ncbw) {
  constexpr int VEC = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7, i = (blockIdx.x >> 3) * 4 + wave;
  if (i >= d) return;
  const long z = blockIdx.z;
  const double* uz = u + z * K;
  const long e0 = rowptr[i], e1 = rowptr[i + 1];
  const long rbase = z * stride + (long)i * d;
  for (long eb = e0; eb < e1 || eb == e0; eb += 64) {
    const int cnt = (int)max(0L, min(64L, e1 - eb));
    int mycol = i;  // (lanes beyond the row's entries name a valid row; their weight is never used)
    double myw = 0.0;
    if (lane < cnt) {
      const int k = kid[eb + lane];
      mycol = col[eb + lane];
      myw = mdt * (k < 0 ? 1.0 : uz[k]) * val[eb + lane];
    }
    const int wl = (int)(__double_as_longlong(myw) & 0xffffffffll), wh = (int)(__double_as_longlong(myw) >> 32);
    const bool first = eb == e0;
    for (int cb = xcd * ncbw; cb < (xcd + 1) * ncbw; ++cb) {
      const int c0 = cb * 64 * VEC + lane * VEC;
      const T* Vz = V + z * stride + c0;
      vec_t acc;
      if (!first) acc = *(const vec_t*)(out + rbase + c0);
      else if (Add) acc = (T)ca * *(const vec_t*)(Add + rbase + c0);
      else acc = 0;
      for (int q0 = 0; q0 < cnt; q0 += RSPS_NQ) {
        vec_t v[RSPS_NQ];
        T w[RSPS_NQ];
#pragma unroll
        for (int k = 0; k < RSPS_NQ; ++k) {
          const int q = q0 + k < cnt ? q0 + k : q0;  // wave-uniform; the tail re-reads a valid row and is skipped below
          const int jc = __builtin_amdgcn_readlane(mycol, q);
          w[k] = (T)__longlong_as_double(((long long)__builtin_amdgcn_readlane(wh, q) << 32) |
                                         (unsigned)__builtin_amdgcn_readlane(wl, q));
          v[k] = *(const vec_t*)(Vz + (long)jc * d);
        }
#pragma unroll
        for (int k = 0; k < RSPS_NQ; ++k)
          if (q0 + k < cnt) acc += w[k] * v[k];
      }
      *(vec_t*)(out + rbase + c0) = acc;
    }
    if (e1 - eb <= 64) break;
  }
}

// whole 16-B column blocks in 8 slabs, 16-B aligned rows
template <class T>
static bool rsp_slab_ok(const void* a, const void* b, const void* c, const void* e, long stride, int d) {
  constexpr int VEC = 16 / sizeof(T);
  auto al = [](const void* p) { return ((size_t)p & 15) == 0; };
  return d % (8 * 64 * VEC) == 0 && stride % VEC == 0 && al(a) && al(b) && al(c) && al(e);
}

template <class T>
hipError_t rsp_mm_launch(T* out, const T* V, const T* Add, double ca, long stride, int d, const long* rowptr, const int* col,
                         const int* kid, const double* val, const double* u, int K, int batch, double dt, int slab,
                         hipStream_t st) {
  if (d <= 0 || batch <= 0) return hipSuccess;
  if (slab && rsp_slab_ok<T>(out, V, Add, nullptr, stride, d)) {
    constexpr int VEC = 16 / sizeof(T);
    hipLaunchKernelGGL(rsp_mm_slab_kernel<T>, dim3((unsigned)(8 * ((d + 3) / 4)), 1, (unsigned)batch), dim3(256), 0, st, out, V, Add,
                       ca, stride, d, rowptr, col, kid, val, u, K, -dt, d / (8 * 64 * VEC));
    return hipGetLastError();
  }
  hipLaunchKernelGGL(rsp_mm_kernel<T>, dim3((unsigned)((d + RSPMM_WAVES - 1) / RSPMM_WAVES), 1, (unsigned)batch),
                     dim3(RSPMM_WAVES * 64), 0, st, out, V, Add, ca, stride, d, rowptr, col, kid, val, u, K, -dt);
  return hipGetLastError();
}
template hipError_t rsp_mm_launch<double>(double*, const double*, const double*, double, long, int, const long*, const int*,
                                          const int*, const double*, const double*, int, int, double, int, hipStream_t);
template hipError_t rsp_mm_launch<float>(float*, const float*, const float*, double, long, int, const long*, const int*,
                                         const int*, const double*, const double*, int, int, double, int, hipStream_t);

// The reverse-mode pass with the gradient level fused in:  out = ca Add + X V  as rsp_mm_kernel, and, while the partner rows
// V[j_e,:] of row i's nonzeros stream through the registers anyway,
//     nzacc[z][e] (+)= < V[z][j_e,:] , Xm[z][i,:] >          (the (j_e, i) entry of V Xm^T; V = Xbar_{m+1}, Xm = X_m)
// -- the level sum of the generator's cotangent at X's own pattern, for the price of one more row read per output row
// instead of a separate pass over 2 (2q) matrices. out may be null (dot products only: the level that has no product).
// Rounds of RSPD_NZ entries (their accumulators live in registers); rows with more entries take several rounds over `out`.
constexpr int RSPD_NZ = 16, RSPD_U = 8, RSPD_WAVES = 4;

template <class T>
__global__ void __launch_bounds__(RSPD_WAVES * 64) rsp_mm_dot_kernel(T* __restrict__ out, const T* __restrict__ V,
                                                                     const T* __restrict__ Add, double ca,
                                                                     const T* __restrict__ Xm, long stride, int d,
                                                                     const long* __restrict__ rowptr,
                                                                     const int* __restrict__ col, const int* __restrict__ kid,
                                                                     const double* __restrict__ val,
                                                                     const double* __restrict__ u, int K, double mdt, long nnz,
                                                                     double* __restrict__ nzacc, int accumulate) {
  __shared__ int s_col[RSPD_WAVES][RSPD_NZ];
  __shared__ double s_w[RSPD_WAVES][RSPD_NZ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int i = blockIdx.x * RSPD_WAVES + wave;
  if (i >= d) return;
  const long z = blockIdx.z;
  const T* Vz = V + z * stride;
  const double* uz = u + z * K;
  const long e0 = rowptr[i], e1 = rowptr[i + 1];
  T* orow = out ? out + z * stride + (long)i * d : nullptr;
  const T* xrow = Xm + z * stride + (long)i * d;
  long eb = e0;
  do {
    const int cnt = (int)max(0L, min((long)RSPD_NZ, e1 - eb));
    if (lane < cnt) {
      const int k = kid[eb + lane];
      s_col[wave][lane] = col[eb + lane];
      s_w[wave][lane] = mdt * (k < 0 ? 1.0 : uz[k]) * val[eb + lane];
    }
    __builtin_amdgcn_wave_barrier();
    const bool first = eb == e0;
    double dacc[RSPD_NZ];
#pragma unroll
    for (int q = 0; q < RSPD_NZ; ++q) dacc[q] = 0.0;
    for (int m0 = 0; m0 < d; m0 += 64 * RSPD_U) {
      T acc[RSPD_U], xm[RSPD_U];
#pragma unroll
      for (int r = 0; r < RSPD_U; ++r) {
        const int m = m0 + r * 64 + lane;
        xm[r] = m < d ? xrow[m] : (T)0;
        if (!orow) acc[r] = 0;
        else if (!first) acc[r] = m < d ? orow[m] : (T)0;
        else acc[r] = (Add != nullptr && m < d) ? (T)ca * Add[z * stride + (long)i * d + m] : (T)0;
      }
#pragma unroll
      for (int q = 0; q < RSPD_NZ; ++q) {
        if (q < cnt) {  // wave-uniform
          const int jc = __builtin_amdgcn_readfirstlane(s_col[wave][q]);
          const T w = (T)s_w[wave][q];
          const T* vrow = Vz + (long)jc * d;
          T t = 0;
#pragma unroll
          for (int r = 0; r < RSPD_U; ++r) {
            const int m = m0 + r * 64 + lane;
            const T v = m < d ? vrow[m] : (T)0;
            acc[r] += w * v;
            t += xm[r] * v;
          }
          dacc[q] += (double)t;
        }
      }
      if (orow) {
#pragma unroll
        for (int r = 0; r < RSPD_U; ++r) {
          const int m = m0 + r * 64 + lane;
          if (m < d) orow[m] = acc[r];
        }
      }
    }
#pragma unroll
    for (int q = 0; q < RSPD_NZ; ++q) {
      if (q < cnt) {
        double v = dacc[q];
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) {
          if (accumulate) nzacc[z * nnz + eb + q] += v;
          else nzacc[z * nnz + eb + q] = v;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    eb += RSPD_NZ;
  } while (eb < e1);
}

// XCD-slab form of the fused pass (see rsp_mm_slab_kernel): the dot products of a row are formed per slab (lane partial
// sums over the slab's column blocks, one wave reduction per entry) into part[z][slab][e]; rsp_dotsum_kernel then adds the 8
// slabs in a fixed order into nzacc -- deterministic, 8 nnz doubles of extra traffic.
template <class T, int NQG>
__global__ void __launch_bounds__(256) rsp_mm_dot_slab_kernel(T* __restrict__ out, const T* __restrict__ V,
                                                              const T* __restrict__ Add, double ca, const T* __restrict__ Xm,
                                                              long stride, int d, const long* __restrict__ rowptr,
                                                              const int* __restrict__ col, const int* __restrict__ kid,
                                                              const double* __restrict__ val, const double* __restrict__ u,
                                                              int K, double mdt, long nnz, double* __restrict__ part, int ncbw) {
  constexpr int VEC = 16 / sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  static_assert(RSPS_NQ % NQG == 0, "groups of NQG loads in flight per round of RSPS_NQ entries");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int xcd = blockIdx.x & 7, i = (blockIdx.x >> 3) * 4 + wave;
  if (i >= d) return;
  const long z = blockIdx.z;
  const double* uz = u + z * K;
  const long e0 = rowptr[i], e1 = rowptr[i + 1];
  const long rbase = z * stride + (long)i * d;
  double* prow = part + (z * 8 + xcd) * nnz;
  long eb = e0;
  do {
    const int cnt = (int)max(0L, min((long)RSPS_NQ, e1 - eb));
    int mycol = i;
    double myw = 0.0;
    if (lane < cnt) {
      const int k = kid[eb + lane];
      mycol = col[eb + lane];
      myw = mdt * (k < 0 ? 1.0 : uz[k]) * val[eb + lane];
    }
    const int wl = (int)(__double_as_longlong(myw) & 0xffffffffll), wh = (int)(__double_as_longlong(myw) >> 32);
    const bool first = eb == e0;
    double dacc[RSPS_NQ];
#pragma unroll
    for (int q = 0; q < RSPS_NQ; ++q) dacc[q] = 0.0;
    for (int cb = xcd * ncbw; cb < (xcd + 1) * ncbw; ++cb) {
      const int c0 = cb * 64 * VEC + lane * VEC;
      const T* Vz = V + z * stride + c0;
      const vec_t xm = *(const vec_t*)(Xm + rbase + c0);
      vec_t acc;
      if (!out) acc = 0;
      else if (!first) acc = *(const vec_t*)(out + rbase + c0);
      else if (Add) acc = (T)ca * *(const vec_t*)(Add + rbase + c0);
      else acc = 0;
#pragma unroll
      for (int g0 = 0; g0 < RSPS_NQ; g0 += NQG) {
        if (g0 < cnt || g0 == 0) {  // wave-uniform
          vec_t v[NQG];
          T w[NQG];
#pragma unroll
          for (int k = 0; k < NQG; ++k) {
            const int q = g0 + k < cnt ? g0 + k : 0;
            const int jc = __builtin_amdgcn_readlane(mycol, q);
            w[k] = (T)__longlong_as_double(((long long)__builtin_amdgcn_readlane(wh, q) << 32) |
                                           (unsigned)__builtin_amdgcn_readlane(wl, q));
            v[k] = *(const vec_t*)(Vz + (long)jc * d);
          }
#pragma unroll
          for (int k = 0; k < NQG; ++k)
            if (g0 + k < cnt) {
              acc += w[k] * v[k];
              T t = 0;
#pragma unroll
              for (int j = 0; j < VEC; ++j) t += xm[j] * v[k][j];
              dacc[g0 + k] += (double)t;
            }
        }
      }
      if (out) *(vec_t*)(out + rbase + c0) = acc;
    }
#pragma unroll
    for (int q = 0; q < RSPS_NQ; ++q)
      if (q < cnt) {
        double s = dacc[q];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) prow[eb + q] = s;
      }
    eb += RSPS_NQ;
  } while (eb < e1);
}

// nzacc[z][e] = (accumulate ? nzacc[z][e] : 0) + sum over the 8 slabs of part[z][slab][e], in slab order
__global__ void __launch_bounds__(256) rsp_dotsum_kernel(const double* __restrict__ part, long nnz, double* __restrict__ nzacc,
                                                         int accumulate) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= nnz) return;
  const long z = blockIdx.z;
  double s = accumulate ? nzacc[z * nnz + e] : 0.0;
#pragma unroll
  for (int x = 0; x < 8; ++x) s += part[(z * 8 + x) * nnz + e];
  nzacc[z * nnz + e] = s;
}

// part: batch * 8 * nnz doubles of scratch for the XCD-slab form (null: the one-wave-per-row kernel)
template <class T>
hipError_t rsp_mm_dot_launch(T* out, const T* V, const T* Add, double ca, const T* Xm, long stride, int d, const long* rowptr,
                             const int* col, const int* kid, const double* val, const double* u, int K, int batch, double dt,
                             long nnz, double* nzacc, int accumulate, double* part, hipStream_t st) {
  if (d <= 0 || batch <= 0) return hipSuccess;
  if (part && nnz > 0 && rsp_slab_ok<T>(out, V, Add, Xm, stride, d)) {
    constexpr int VEC = 16 / sizeof(T);
#ifndef DQ_RSPD_NQG
#define DQ_RSPD_NQG 8
#endif
    hipLaunchKernelGGL((rsp_mm_dot_slab_kernel<T, DQ_RSPD_NQG>), dim3((unsigned)(8 * ((d + 3) / 4)), 1, (unsigned)batch), dim3(256), 0,
                       st, out, V, Add, ca, Xm, stride, d, rowptr, col, kid, val, u, K, -dt, nnz, part, d / (8 * 64 * VEC));
    hipLaunchKernelGGL(rsp_dotsum_kernel, dim3((unsigned)((nnz + 255) / 256), 1, (unsigned)batch), dim3(256), 0, st, part, nnz, nzacc,
                       accumulate);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(rsp_mm_dot_kernel<T>, dim3((unsigned)((d + RSPD_WAVES - 1) / RSPD_WAVES), 1, (unsigned)batch),
                     dim3(RSPD_WAVES * 64), 0, st, out, V, Add, ca, Xm, stride, d, rowptr, col, kid, val, u, K, -dt, nnz, nzacc,
                     accumulate);
  return hipGetLastError();
}
template hipError_t rsp_mm_dot_launch<double>(double*, const double*, const double*, double, const double*, long, int,
                                              const long*, const int*, const int*, const double*, const double*, int, int,
                                              double, long, double*, int, double*, hipStream_t);
template hipError_t rsp_mm_dot_launch<float>(float*, const float*, const float*, double, const float*, long, int, const long*,
                                             const int*, const int*, const double*, const double*, int, int, double, long,
                                             double*, int, double*, hipStream_t);

// nzval[z][e] = val_e (nzacc[z][e] + dc D[z][j][i] + sum_{p in row j} w_p G2[z][col_p][i])  for the control entries e = (i, j)
// (0 for drift entries): the level sums of rsp_mm_dot_kernel, which sit at the TRANSPOSED position (j, i) of the generator's
// cotangent, completed by Xbar_1 = dc Sbar + X Xbar_2 at that same position -- the total is symmetric, its pieces are not.
template <class T>
__global__ void __launch_bounds__(256) rsp_finish_kernel(const double* __restrict__ nzacc, const T* __restrict__ D,
                                                         const T* __restrict__ G2, double dc, long stride, int d,
                                                         const long* __restrict__ rowptr, const int* __restrict__ col,
                                                         const int* __restrict__ kid, const double* __restrict__ val,
                                                         const double* __restrict__ u, int K, double mdt, long nnz,
                                                         double* __restrict__ nzval) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= d) return;
  const long z = blockIdx.z;
  const long zo = z * stride;
  const double* uz = u + z * K;
  for (long e = rowptr[i]; e < rowptr[i + 1]; ++e) {
    if (kid[e] < 0) { nzval[z * nnz + e] = 0.0; continue; }
    const int j = col[e];
    double v = nzacc[z * nnz + e] + dc * (double)D[zo + (long)j * d + i];
    for (long p = rowptr[j]; p < rowptr[j + 1]; ++p) {
      const int k = kid[p];
      v += mdt * (k < 0 ? 1.0 : uz[k]) * val[p] * (double)G2[zo + (long)col[p] * d + i];
    }
    nzval[z * nnz + e] = v * val[e];
  }
}

template <class T>
hipError_t rsp_finish_launch(const double* nzacc, const T* D, const T* G2, double dc, long stride, int d, const long* rowptr,
                             const int* col, const int* kid, const double* val, const double* u, int K, int batch, double dt,
                             long nnz, double* nzval, hipStream_t st) {
  if (d <= 0 || batch <= 0 || nnz <= 0) return hipSuccess;
  hipLaunchKernelGGL(rsp_finish_kernel<T>, dim3((unsigned)((d + 255) / 256), 1, (unsigned)batch), dim3(256), 0, st, nzacc, D, G2,
                     dc, stride, d, rowptr, col, kid, val, u, K, -dt, nnz, nzval);
  return hipGetLastError();
}
template hipError_t rsp_finish_launch<double>(const double*, const double*, const double*, double, long, int, const long*,
                                              const int*, const int*, const double*, const double*, int, int, double, long,
                                              double*, hipStream_t);
template hipError_t rsp_finish_launch<float>(const double*, const float*, const float*, double, long, int, const long*,
                                             const int*, const int*, const double*, const double*, int, int, double, long, double*,
                                             hipStream_t);

// nzval[z][e] (+)= val_e * ( direct_e + sum_{l < nl} <G_l[z][i,:], Xp_l[z][j,:]> )   for the CONTROL entries e = (i, j)
// with j >= i (doubled for j > i: the total cotangent Xi is symmetric and so are the H_k; drift entries, kid < 0, and the
// entries below the diagonal get 0), with  direct_e = dc D[z][i][j] + sum_{e' in row i} w_e' G2[z][col_e'][j]  when D is
// given (the cotangent of the dense copy X_1: Xbar_1 = dc Sbar + X Xbar_2, needed at the controls' nonzeros only).
constexpr int RSPL_MAX = 6, RSPL_U = 8, RSPL_NZ = 8;

template <class T>
struct RspLevels {
  const T* G[RSPL_MAX];   // cotangents Xbar_{m+1}
  const T* Xp[RSPL_MAX];  // powers X_m
  int nl;
  const T* D;             // direct term (may be null)
  const T* G2;
  double dc;
  int accumulate;
};

// One workgroup per row, one WAVE PER LEVEL: wave l keeps a 512-column strip of G_l[i,:] in registers while the partner rows
// Xp_l[j,:] of up to RSPL_NZ of the row's gradient-carrying entries stream past (the loop of the sparse x dense kernel, 8
// waves per SIMD); the per-level partial sums meet in LDS, wave 0 adds the direct term and writes one value per entry.
template <class T>
__global__ void __launch_bounds__(RSPL_MAX * 64) rsp_levels_kernel(RspLevels<T> a, long stride, int d,
                                                                   const long* __restrict__ rowptr,
                                                                   const int* __restrict__ col, const int* __restrict__ kid,
                                                                   const double* __restrict__ val,
                                                                   const double* __restrict__ u, int K, double mdt, long nnz,
                                                                   double* __restrict__ nzval) {
  // the row's entries that carry gradient (a control's, on or above the diagonal), compacted once per 64 CSR entries
  __shared__ int s_col[64];
  __shared__ int s_ent[64];
  __shared__ int s_nv;
  __shared__ double s_red[RSPL_MAX][RSPL_NZ];
  const int lane = threadIdx.x & 63, lvl = threadIdx.x >> 6;
  const int i = blockIdx.x;
  const long z = blockIdx.z;
  const long zo = z * stride;
  const double* uz = u + z * K;
  const long e0 = rowptr[i], e1 = rowptr[i + 1];
  const bool active = lvl < a.nl;
  const T* grow = active ? a.G[lvl] + zo + (long)i * d : nullptr;
  const T* xbase = active ? a.Xp[lvl] + zo : nullptr;
  for (long eb = e0; eb < e1; eb += 64) {
    if (lvl == 0) {
      const long e = eb + lane;
      const bool in = e < e1;
      const int cj = in ? col[e] : 0;
      const bool valid = in && kid[e] >= 0 && cj >= i;
      const unsigned long long mask = __ballot(valid);
      if (valid) {
        const int pos = __popcll(mask & ((1ull << lane) - 1ull));
        s_col[pos] = cj;
        s_ent[pos] = (int)(e - e0);
      } else if (in && !a.accumulate) {
        nzval[z * nnz + e] = 0.0;
      }
      if (lane == 0) s_nv = __popcll(mask);
    }
    __syncthreads();
    const int nv = s_nv;
    for (int q0 = 0; q0 < nv; q0 += RSPL_NZ) {
      const int cnt = min((int)RSPL_NZ, nv - q0);
      double acc[RSPL_NZ];
#pragma unroll
      for (int q = 0; q < RSPL_NZ; ++q) acc[q] = 0.0;
      if (active) {
        for (int m0 = 0; m0 < d; m0 += 64 * RSPL_U) {
          T g[RSPL_U];
#pragma unroll
          for (int r = 0; r < RSPL_U; ++r) {
            const int m = m0 + r * 64 + lane;
            g[r] = m < d ? grow[m] : (T)0;
          }
#pragma unroll
          for (int q = 0; q < RSPL_NZ; ++q) {
            if (q < cnt) {
              const int j = __builtin_amdgcn_readfirstlane(s_col[q0 + q]);
              const T* x = xbase + (long)j * d;
              T t[RSPL_U];  // independent chains
#pragma unroll
              for (int r = 0; r < RSPL_U; ++r) {
                const int m = m0 + r * 64 + lane;
                t[r] = m < d ? g[r] * x[m] : (T)0;
              }
              T ts = 0;
#pragma unroll
              for (int r = 0; r < RSPL_U; ++r) ts += t[r];
              acc[q] += (double)ts;
            }
          }
        }
      }
#pragma unroll
      for (int q = 0; q < RSPL_NZ; ++q) {
        if (q < cnt) {
          double v = acc[q];
          for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
          if (lane == 0) s_red[lvl][q] = v;
        }
      }
      __syncthreads();
      if (lvl == 0) {
        for (int q = 0; q < cnt; ++q) {
          const int j = __builtin_amdgcn_readfirstlane(s_col[q0 + q]);
          const long eq = e0 + __builtin_amdgcn_readfirstlane(s_ent[q0 + q]);
          double v = 0.0;
          if (a.D != nullptr) {  // Xbar_1[i][j] = dc D[i][j] + sum over the nonzeros of row i of X (one per lane)
            for (long p = e0 + lane; p < e1; p += 64) {
              const int k = kid[p];
              v += mdt * (k < 0 ? 1.0 : uz[k]) * val[p] * (double)a.G2[zo + (long)col[p] * d + j];
            }
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
          }
          if (lane == 0) {
            if (a.D != nullptr) v += a.dc * (double)a.D[zo + (long)i * d + j];
            for (int l = 0; l < RSPL_MAX; ++l) v += s_red[l][q];  // (inactive levels left zeros)
            v *= val[eq] * (j > i ? 2.0 : 1.0);
            if (a.accumulate) nzval[z * nnz + eq] += v;
            else nzval[z * nnz + eq] = v;
          }
        }
      }
      __syncthreads();
    }
    __syncthreads();  // (rows with more than 64 entries: s_nv / s_col are rewritten by wave 0 in the next round)
  }
}

template <class T>
hipError_t rsp_levels_launch(const T* const* G, const T* const* Xp, int nl, const T* D, const T* G2, double dc, long stride,
                             int d, const long* rowptr, const int* col, const int* kid, const double* val, const double* u,
                             int K, int batch, double dt, long nnz, double* nzval, hipStream_t st) {
  if (d <= 0 || batch <= 0 || nnz <= 0) return hipSuccess;
  int l0 = 0;
  do {
    RspLevels<T> a;
    memset(&a, 0, sizeof(a));
    a.nl = std::max(0, std::min((int)RSPL_MAX, nl - l0));
    for (int l = 0; l < a.nl; ++l) { a.G[l] = G[l0 + l]; a.Xp[l] = Xp[l0 + l]; }
    a.accumulate = l0 > 0 ? 1 : 0;
    if (l0 == 0) { a.D = D; a.G2 = G2; a.dc = dc; }
    hipLaunchKernelGGL(rsp_levels_kernel<T>, dim3((unsigned)d, 1, (unsigned)batch), dim3(RSPL_MAX * 64), 0, st, a, stride, d,
                       rowptr, col, kid, val, u, K, -dt, nnz, nzval);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    l0 += RSPL_MAX;
  } while (l0 < nl);
  return hipSuccess;
}
template hipError_t rsp_levels_launch<double>(const double* const*, const double* const*, int, const double*, const double*,
                                              double, long, int, const long*, const int*, const int*, const double*,
                                              const double*, int, int, double, long, double*, hipStream_t);
template hipError_t rsp_levels_launch<float>(const float* const*, const float* const*, int, const float*, const float*, double,
                                             long, int, const long*, const int*, const int*, const double*, const double*, int,
                                             int, double, long, double*, hipStream_t);

// grad[z][k] = scale * sum of nzval[z][e] over the entries of control k (the reduction of rsp_contract_launch, on its own)
hipError_t rsp_reduce_launch(const double* nzval, long nnz, const long* kptr, const long* kent, int K, int batch, double scale,
                             double* grad, hipStream_t st) {
  if (K <= 0 || batch <= 0) return hipSuccess;
  hipLaunchKernelGGL(rsp_reduce_kernel, dim3((unsigned)K, 1, (unsigned)batch), dim3(256), 0, st, nzval, nnz, kptr, kent, K,
                     scale, grad);
  return hipGetLastError();
}

}  // namespace dq
