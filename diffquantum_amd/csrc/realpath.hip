// Streaming kernels of the real-symmetric fast path (see rgemm.hip for when it applies): everything here is
// HBM-bound, one real scalar (8 B fp64 / 4 B fp32) per lane and access, consecutive lanes on consecutive
// elements; the transposing kernels go through a 32x33 LDS tile so both the direct and the transposed access
// are coalesced. Reductions accumulate in fp64 in a fixed order (bitwise reproducible gradients).
//
//   step generator   A = -i dt (H0 + sum_k u_k H_k) = i X,  X = -dt (H0 + sum_k u_k H_k)   (diffqc.cc:192-197)
//   propagator       U = exp(A) = cos X + i sin X = C + i S                                 (diffqc.cc:198)
#include <hip/hip_runtime.h>
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
    a.out[0][base + e] = v0;
    if (a.n_out > 1) a.out[1][base + e] = v1;
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

}  // namespace dq
