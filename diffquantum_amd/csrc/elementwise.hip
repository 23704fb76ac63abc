// HBM-bound streaming kernels of the propagator path: Hamiltonian assembly, the linear
// combinations between the polynomial's products, the control-gradient contractions and the
// small reductions (energies, norms). All operate on interleaved complex128, 16 B per lane
// per access, consecutive lanes on consecutive elements (fully coalesced 1 KiB per wave).
#include <hip/hip_runtime.h>
#include "dq_internal.h"

namespace dq {

// ---- A_t = -i dt (H0 + sum_k u_k(t) H_k) -------------------------------------------------
// Reference: /root/reference/diffqc.cc:192-197 (dH = -i*dt*g_H0; dH += -i*dt*coeff*g_Hs[h]),
//            /root/reference/sim_plain.py:137-143.  One launch assembles `batch` time steps.
__global__ void __launch_bounds__(256) assemble_kernel(double2* __restrict__ A, const double2* __restrict__ H0,
                                                       const double2* __restrict__ Hs, const double* __restrict__ u,
                                                       int K, long n_elem, double dt) {
  const long z = blockIdx.z;
  const double* uz = u + z * K;
  double2* Az = A + z * n_elem;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_elem; e += (long)gridDim.x * blockDim.x) {
    double2 h = H0[e];
    for (int k = 0; k < K; ++k) {
      const double c = uz[k];
      const double2 x = Hs[(long)k * n_elem + e];
      h.x = fma(c, x.x, h.x);
      h.y = fma(c, x.y, h.y);
    }
    Az[e] = make_double2(dt * h.y, -dt * h.x);  // (-i) * (x + i y) = y - i x
  }
}

hipError_t assemble_launch(double2* A, const double2* H0, const double2* Hs, const double* u, int K, long n_elem,
                           int batch, double dt, hipStream_t st) {
  if (n_elem <= 0 || batch <= 0) return hipSuccess;
  long nb = (n_elem + 255) / 256;
  if (nb > 65536) nb = 65536;
  dim3 grid((unsigned)nb, 1, (unsigned)batch);
  hipLaunchKernelGGL(assemble_kernel, grid, dim3(256), 0, st, A, H0, Hs, u, K, n_elem, dt);
  return hipGetLastError();
}

// ---- out = [out +] cI*I + sum_j c_j X_j ---------------------------------------------------
__global__ void __launch_bounds__(256) lincomb_kernel(LincombArgs a) {
  const long z = blockIdx.z;
  const long n_elem = (long)a.d * a.d;
  const long base = z * a.stride;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_elem; e += (long)gridDim.x * blockDim.x) {
    double2 v = a.accumulate ? a.out[base + e] : make_double2(0.0, 0.0);
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (j < a.n_in) {
        const double2 x = a.x[j][base + e];
        v.x = fma(a.c[j], x.x, v.x);
        v.y = fma(a.c[j], x.y, v.y);
      }
    if (a.cI != 0.0 && (e / a.d) == (e % a.d)) v.x += a.cI;
    a.out[base + e] = v;
  }
}

hipError_t lincomb_launch(const LincombArgs& a, hipStream_t st) {
  const long n_elem = (long)a.d * a.d;
  if (n_elem <= 0 || a.batch <= 0) return hipSuccess;
  long nb = (n_elem + 255) / 256;
  if (nb > 65536) nb = 65536;
  dim3 grid((unsigned)nb, 1, (unsigned)a.batch);
  hipLaunchKernelGGL(lincomb_kernel, grid, dim3(256), 0, st, a);
  return hipGetLastError();
}

// ---- dL/du_k[t] = Re<Abar_t, -i dt H_k>_F = dt * sum Im(conj(Abar) .* H_k) ---------------
// Stage 1: per-block partial sums for up to KC controls at a time; stage 2: fixed-order sum over
// blocks (no atomics -> bitwise reproducible gradients).
constexpr int KC = 8;
constexpr int GC_THREADS = 256;

__global__ void __launch_bounds__(GC_THREADS) grad_contract_kernel(const double2* __restrict__ Abar, long strideA,
                                                                   const double2* __restrict__ Hs, int K, int k0,
                                                                   long n_elem, double* __restrict__ part) {
  const long z = blockIdx.z;
  const double2* Az = Abar + z * strideA;
  double acc[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) acc[k] = 0.0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_elem; e += (long)gridDim.x * blockDim.x) {
    const double2 a = Az[e];
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (k0 + k < K) {
        const double2 h = Hs[(long)(k0 + k) * n_elem + e];
        acc[k] += a.x * h.y - a.y * h.x;  // Im(conj(a) * h)
      }
  }
  __shared__ double red[KC][GC_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    double v = acc[k];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[k][wave] = v;
  }
  __syncthreads();
  if (threadIdx.x < KC && k0 + (int)threadIdx.x < K) {
    double v = 0.0;
    for (int w = 0; w < GC_THREADS / 64; ++w) v += red[threadIdx.x][w];
    part[(z * gridDim.x + blockIdx.x) * K + k0 + threadIdx.x] = v;
  }
}

__global__ void grad_finish_kernel(const double* __restrict__ part, int nblk, int K, double scale,
                                   double* __restrict__ grad) {
  const long z = blockIdx.x;
  const int k = threadIdx.x;
  if (k >= K) return;
  double v = 0.0;
  for (int b = 0; b < nblk; ++b) v += part[(z * nblk + b) * K + k];
  grad[z * K + k] = scale * v;
}

int grad_contract_blocks(long n_elem) {
  long nb = (n_elem + GC_THREADS * 8 - 1) / (GC_THREADS * 8);
  if (nb < 1) nb = 1;
  if (nb > 1024) nb = 1024;
  return (int)nb;
}

hipError_t grad_contract_launch(const double2* Abar, long strideA, const double2* Hs, int K, long n_elem, int batch,
                                double scale, double* part, double* grad, hipStream_t st) {
  if (K <= 0 || batch <= 0) return hipSuccess;
  const int nb = grad_contract_blocks(n_elem);
  dim3 grid((unsigned)nb, 1, (unsigned)batch);
  for (int k0 = 0; k0 < K; k0 += KC)
    hipLaunchKernelGGL(grad_contract_kernel, grid, dim3(GC_THREADS), 0, st, Abar, strideA, Hs, K, k0, n_elem, part);
  const int threads = ((K + 63) / 64) * 64;
  hipLaunchKernelGGL(grad_finish_kernel, dim3((unsigned)batch), dim3(threads), 0, st, part, nb, K, scale, grad);
  return hipGetLastError();
}

// ---- e[b] = scale * sum_i Re(conj(X[i][b]) Y[i][b]) ---------------------------------------
// Energy <psi_b| M |psi_b> with Y = M Psi (sim_plain.py:281 `M.matrix_element(final, final).real`).
__global__ void __launch_bounds__(256) coldot_kernel(const double2* __restrict__ X, long ldx,
                                                     const double2* __restrict__ Y, long ldy, int d, int B,
                                                     double scale, double* __restrict__ e) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;
  for (int i = 0; i < d; ++i) {
    const double2 x = X[(long)i * ldx + b];
    const double2 y = Y[(long)i * ldy + b];
    s += x.x * y.x + x.y * y.y;
  }
  e[b] = scale * s;
}

hipError_t coldot_launch(const double2* X, long ldx, const double2* Y, long ldy, int d, int B, double scale, double* e,
                         hipStream_t st) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(coldot_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, X, ldx, Y, ldy, d, B, scale,
                     e);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) sum_kernel(const double* __restrict__ e, int n, double* __restrict__ out) {
  __shared__ double red[256];
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) v += e[i];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

hipError_t sum_launch(const double* e, int n, double* out, hipStream_t st) {
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(256), 0, st, e, n, out);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) colabs_kernel(const double2* __restrict__ X, int d, double* __restrict__ cs) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d) return;
  double s = 0.0;
  for (int i = 0; i < d; ++i) {
    const double2 x = X[(long)i * d + j];
    s += hypot(x.x, x.y);
  }
  cs[j] = s;
}

hipError_t colabs_launch(const double2* X, int d, double* colsum, hipStream_t st) {
  hipLaunchKernelGGL(colabs_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, X, d, colsum);
  return hipGetLastError();
}

__global__ void __launch_bounds__(256) herm_defect_kernel(const double2* __restrict__ X, int d,
                                                          unsigned long long* __restrict__ out) {
  const long n = (long)d * d;
  double mx = 0.0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const long i = e / d, j = e % d;
    const double2 a = X[e];
    const double2 b = X[j * d + i];
    mx = fmax(mx, fmax(fabs(a.x - b.x), fabs(a.y + b.y)));
  }
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_down(mx, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(mx));  // mx >= 0: order-preserving
}

hipError_t herm_defect_launch(const double2* X, int d, double* out, hipStream_t st) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(double), st);
  if (e != hipSuccess) return e;
  long nb = ((long)d * d + 255) / 256;
  if (nb > 4096) nb = 4096;
  hipLaunchKernelGGL(herm_defect_kernel, dim3((unsigned)nb), dim3(256), 0, st, X, d, (unsigned long long*)out);
  return hipGetLastError();
}

}  // namespace dq
