// HBM-bound streaming kernels of the propagator path: Hamiltonian assembly, the linear
// combinations between the polynomial's products, the control-gradient contractions and the
// small reductions (energies, norms). All operate on interleaved complex (fp64: 16 B, fp32: 8 B per lane
// per access), consecutive lanes on consecutive elements (fully coalesced). Reductions accumulate in fp64
// for both precisions. T = real scalar (double / float).
#include <hip/hip_runtime.h>
#include "dq_internal.h"

namespace dq {

// ---- A_t = -i dt (H0 + sum_k u_k(t) H_k) -------------------------------------------------
// Reference: /root/reference/diffqc.cc:192-197 (dH = -i*dt*g_H0; dH += -i*dt*coeff*g_Hs[h]),
//            /root/reference/sim_plain.py:137-143.  One launch assembles `batch` time steps.
template <class T>
__global__ void __launch_bounds__(256) assemble_kernel(typename Cx<T>::type* __restrict__ A,
                                                       const typename Cx<T>::type* __restrict__ H0,
                                                       const typename Cx<T>::type* __restrict__ Hs,
                                                       const double* __restrict__ u, int K, long n_elem, double dt_) {
  typedef typename Cx<T>::type T2;
  const long z = blockIdx.z;
  const double* uz = u + z * K;
  T2* Az = A + z * n_elem;
  const T dt = (T)dt_;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_elem; e += (long)gridDim.x * blockDim.x) {
    T2 h = H0[e];
    for (int k = 0; k < K; ++k) {
      const T c = (T)uz[k];
      const T2 x = Hs[(long)k * n_elem + e];
      h.x = fma(c, x.x, h.x);
      h.y = fma(c, x.y, h.y);
    }
    T2 o;  // (-i) * (x + i y) = y - i x
    o.x = dt * h.y;
    o.y = -dt * h.x;
    Az[e] = o;
  }
}

template <class T>
hipError_t assemble_launch(typename Cx<T>::type* A, const typename Cx<T>::type* H0, const typename Cx<T>::type* Hs,
                           const double* u, int K, long n_elem, int batch, double dt, hipStream_t st) {
  if (n_elem <= 0 || batch <= 0) return hipSuccess;
  long nb = (n_elem + 255) / 256;
  if (nb > 65536) nb = 65536;
  dim3 grid((unsigned)nb, 1, (unsigned)batch);
  hipLaunchKernelGGL(assemble_kernel<T>, grid, dim3(256), 0, st, A, H0, Hs, u, K, n_elem, dt);
  return hipGetLastError();
}
template hipError_t assemble_launch<double>(double2*, const double2*, const double2*, const double*, int, long, int,
                                            double, hipStream_t);
template hipError_t assemble_launch<float>(float2*, const float2*, const float2*, const double*, int, long, int, double,
                                           hipStream_t);

// ---- out = [out +] cI*I + sum_j c_j X_j ---------------------------------------------------
template <class T>
__global__ void __launch_bounds__(256) lincomb_kernel(LincombArgs<T> a) {
  typedef typename Cx<T>::type T2;
  const long z = blockIdx.z;
  const long n_elem = (long)a.d * a.d;
  const long base = z * a.stride;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_elem; e += (long)gridDim.x * blockDim.x) {
    T2 v;
    v.x = 0;
    v.y = 0;
    if (a.accumulate) v = a.out[base + e];
#pragma unroll
    for (int j = 0; j < 6; ++j)
      if (j < a.n_in) {
        const T2 x = a.x[j][base + e];
        v.x = fma((T)a.c[j], x.x, v.x);
        v.y = fma((T)a.c[j], x.y, v.y);
      }
    if (a.cI != 0.0 && (e / a.d) == (e % a.d)) v.x += (T)a.cI;
    a.out[base + e] = v;
  }
}

template <class T>
hipError_t lincomb_launch(const LincombArgs<T>& a, hipStream_t st) {
  const long n_elem = (long)a.d * a.d;
  if (n_elem <= 0 || a.batch <= 0) return hipSuccess;
  long nb = (n_elem + 255) / 256;
  if (nb > 65536) nb = 65536;
  dim3 grid((unsigned)nb, 1, (unsigned)a.batch);
  hipLaunchKernelGGL(lincomb_kernel<T>, grid, dim3(256), 0, st, a);
  return hipGetLastError();
}
template hipError_t lincomb_launch<double>(const LincombArgs<double>&, hipStream_t);
template hipError_t lincomb_launch<float>(const LincombArgs<float>&, hipStream_t);

// ---- out = [out +] a * X + b * X^H   (d x d, batched; LDS-tiled transpose so both accesses are coalesced) ------
// Used by the reverse mode of v1 = A A for anti-Hermitian A: S = herm(v1bar) (a = b = 1/2) and
// Abar += X^H - X (a = -1, b = 1) -- only the skew part of Abar reaches Re<Abar, -i dt H_k>.
template <class T>
__global__ void __launch_bounds__(256) tcombine_kernel(typename Cx<T>::type* __restrict__ out,
                                                       const typename Cx<T>::type* __restrict__ X, int d, double a_,
                                                       double b_, int accumulate, long stride) {
  typedef typename Cx<T>::type T2;
  __shared__ T2 tl[32][33];
  const long base = (long)blockIdx.z * stride;
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const T a = (T)a_, b = (T)b_;
  // transposed operand: tile X[j0 + r][i0 + c]
  for (int r = ty; r < 32; r += 8) {
    T2 v; v.x = 0; v.y = 0;
    if (j0 + r < d && i0 + tx < d) v = X[base + (long)(j0 + r) * d + i0 + tx];
    tl[r][tx] = v;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int i = i0 + r, j = j0 + tx;
    if (i < d && j < d) {
      const long off = base + (long)i * d + j;
      const T2 x = X[off];
      const T2 xt = tl[tx][r];  // X[j][i]
      T2 o;
      o.x = a * x.x + b * xt.x;
      o.y = a * x.y - b * xt.y;  // conj
      if (accumulate) { const T2 q = out[off]; o.x += q.x; o.y += q.y; }
      out[off] = o;
    }
  }
}

template <class T>
hipError_t tcombine_launch(typename Cx<T>::type* out, const typename Cx<T>::type* X, int d, double a, double b,
                           int accumulate, long stride, int batch, hipStream_t st) {
  if (d <= 0 || batch <= 0) return hipSuccess;
  dim3 grid((unsigned)((d + 31) / 32), (unsigned)((d + 31) / 32), (unsigned)batch);
  hipLaunchKernelGGL(tcombine_kernel<T>, grid, dim3(256), 0, st, out, X, d, a, b, accumulate, stride);
  return hipGetLastError();
}
template hipError_t tcombine_launch<double>(double2*, const double2*, int, double, double, int, long, int, hipStream_t);
template hipError_t tcombine_launch<float>(float2*, const float2*, int, double, double, int, long, int, hipStream_t);

// ---- out[z][j][i] = in[z][i][j] (conj: its complex conjugate), rows x cols -> cols x rows, explicit leading dimensions ----
// The pre-pass that gives a complex GEMM operand a k-major form (zgemm_lin.hip) when it has none: LDS-tiled, both sides coalesced.
template <class T>
__global__ void __launch_bounds__(256) ztranspose_kernel(typename Cx<T>::type* __restrict__ out, long ldo, long so,
                                                         const typename Cx<T>::type* __restrict__ in, long ldi, long si,
                                                         int rows, int cols, int conj) {
  typedef typename Cx<T>::type T2;
  __shared__ T2 tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const T2* I = in + (long)blockIdx.z * si;
  T2* O = out + (long)blockIdx.z * so;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    if (r < rows && c < cols) tile[ty + 8 * k][tx] = I[(long)r * ldi + c];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;  // output row c, output column r
    if (r < rows && c < cols) {
      T2 v = tile[tx][ty + 8 * k];
      if (conj) v.y = -v.y;
      O[(long)c * ldo + r] = v;
    }
  }
}

template <class T>
hipError_t ztranspose_launch(typename Cx<T>::type* out, long ldo, long so, const typename Cx<T>::type* in, long ldi, long si,
                             int rows, int cols, int conj, int batch, hipStream_t st) {
  if (rows <= 0 || cols <= 0 || batch <= 0) return hipSuccess;
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)batch);
  hipLaunchKernelGGL(ztranspose_kernel<T>, grid, dim3(256), 0, st, out, ldo, so, in, ldi, si, rows, cols, conj);
  return hipGetLastError();
}
template hipError_t ztranspose_launch<double>(double2*, long, long, const double2*, long, long, int, int, int, int, hipStream_t);
template hipError_t ztranspose_launch<float>(float2*, long, long, const float2*, long, long, int, int, int, int, hipStream_t);

// ---- dL/du_k[t] = Re<Abar_t, -i dt H_k>_F = dt * sum Im(conj(Abar) .* H_k) ---------------
// Stage 1: per-block partial sums for up to KC controls at a time; stage 2: fixed-order sum over
// blocks (no atomics -> bitwise reproducible gradients).
constexpr int KC = 8;
constexpr int GC_THREADS = 256;

template <class T>
__global__ void __launch_bounds__(GC_THREADS) grad_contract_kernel(const typename Cx<T>::type* __restrict__ Abar,
                                                                   long strideA,
                                                                   const typename Cx<T>::type* __restrict__ Hs, int K,
                                                                   int k0, long n_elem, double* __restrict__ part) {
  typedef typename Cx<T>::type T2;
  const long z = blockIdx.z;
  const T2* Az = Abar + z * strideA;
  double acc[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) acc[k] = 0.0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_elem; e += (long)gridDim.x * blockDim.x) {
    const T2 a = Az[e];
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (k0 + k < K) {
        const T2 h = Hs[(long)(k0 + k) * n_elem + e];
        acc[k] += (double)a.x * (double)h.y - (double)a.y * (double)h.x;  // Im(conj(a) * h)
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

// grad[z][k] = scale * sum_b part[z][b][k]: one 256-thread block per (z, k); thread t adds blocks t, t + 256, ... in
// order, then a fixed-shape tree over the 256 partial sums (same order on every run -> reproducible bits)
__global__ void __launch_bounds__(256) grad_finish_kernel(const double* __restrict__ part, int nblk, int K, double scale,
                                                          double* __restrict__ grad) {
  __shared__ double red[256];
  const long z = blockIdx.x;
  const int k = blockIdx.y;
  double v = 0.0;
  for (int b = threadIdx.x; b < nblk; b += 256) v += part[(z * nblk + b) * K + k];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) grad[z * K + k] = scale * red[0];
}

hipError_t grad_finish_launch(const double* part, int nblk, int K, int batch, double scale, double* grad, hipStream_t st) {
  if (K <= 0 || batch <= 0) return hipSuccess;
  hipLaunchKernelGGL(grad_finish_kernel, dim3((unsigned)batch, (unsigned)K), dim3(256), 0, st, part, nblk, K, scale, grad);
  return hipGetLastError();
}

int grad_contract_blocks(long n_elem) {
  long nb = (n_elem + GC_THREADS * 8 - 1) / (GC_THREADS * 8);
  if (nb < 1) nb = 1;
  if (nb > 1024) nb = 1024;
  return (int)nb;
}

template <class T>
hipError_t grad_contract_launch(const typename Cx<T>::type* Abar, long strideA, const typename Cx<T>::type* Hs, int K,
                                long n_elem, int batch, double scale, double* part, double* grad, hipStream_t st) {
  if (K <= 0 || batch <= 0) return hipSuccess;
  const int nb = grad_contract_blocks(n_elem);
  dim3 grid((unsigned)nb, 1, (unsigned)batch);
  for (int k0 = 0; k0 < K; k0 += KC)
    hipLaunchKernelGGL(grad_contract_kernel<T>, grid, dim3(GC_THREADS), 0, st, Abar, strideA, Hs, K, k0, n_elem, part);
  return grad_finish_launch(part, nb, K, batch, scale, grad, st);
}
template hipError_t grad_contract_launch<double>(const double2*, long, const double2*, int, long, int, double, double*,
                                                 double*, hipStream_t);
template hipError_t grad_contract_launch<float>(const float2*, long, const float2*, int, long, int, double, double*,
                                                double*, hipStream_t);

// ---- e[b] = scale * sum_i Re(conj(X[i][b]) Y[i][b]) ---------------------------------------
// Energy <psi_b| M |psi_b> with Y = M Psi (sim_plain.py:281 `M.matrix_element(final, final).real`).
// Two-stage so that tall-skinny batches (d = 2^18 rows, 64 columns) use the whole chip: stage 1 splits the rows
// into chunks (blockIdx.y), 64 columns x 4 row-lanes per block; stage 2 adds the chunks in a fixed order.
template <class T>
__global__ void __launch_bounds__(256) coldot_kernel(const typename Cx<T>::type* __restrict__ X, long ldx,
                                                     const typename Cx<T>::type* __restrict__ Y, long ldy, int d,
                                                     int B, int rows_per_chunk, double* __restrict__ part) {
  typedef typename Cx<T>::type T2;
  __shared__ double red[4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int b = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(d, r0 + rows_per_chunk);
  double s = 0.0;
  if (b < B)
    for (int i = r0 + rg; i < r1; i += 4) {
      const T2 x = X[(long)i * ldx + b];
      const T2 y = Y[(long)i * ldy + b];
      s += (double)x.x * (double)y.x + (double)x.y * (double)y.y;
    }
  red[rg][cl] = s;
  __syncthreads();
  if (rg == 0 && b < B) part[(long)blockIdx.y * B + b] = red[0][cl] + red[1][cl] + red[2][cl] + red[3][cl];
}

__global__ void __launch_bounds__(256) coldot_finish_kernel(const double* __restrict__ part, int nchunks, int B,
                                                            double scale, double* __restrict__ e) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double s = 0.0;
  for (int c = 0; c < nchunks; ++c) s += part[(long)c * B + b];
  e[b] = scale * s;
}

int coldot_chunks(int d) { return max(1, min(256, d / 64)); }

// `part` must hold coldot_chunks(d) * B doubles
template <class T>
hipError_t coldot_launch(const typename Cx<T>::type* X, long ldx, const typename Cx<T>::type* Y, long ldy, int d, int B,
                         double scale, double* e, double* part, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  const int nch = coldot_chunks(d);
  const int rpc = (d + nch - 1) / nch;
  hipLaunchKernelGGL(coldot_kernel<T>, dim3((unsigned)((B + 63) / 64), (unsigned)nch), dim3(256), 0, st, X, ldx, Y, ldy,
                     d, B, rpc, part);
  hipLaunchKernelGGL(coldot_finish_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, part, nch, B, scale, e);
  return hipGetLastError();
}
template hipError_t coldot_launch<double>(const double2*, long, const double2*, long, int, int, double, double*,
                                          double*, hipStream_t);
template hipError_t coldot_launch<float>(const float2*, long, const float2*, long, int, int, double, double*, double*,
                                         hipStream_t);

// ---- column-wise complex overlaps c[b] = sum_i conj(X[i][b]) Y[i][b] and the fidelity costate -----------------
// Fidelity objective (the batched form of train_fidelity, sim_plain.py:414-475: M = |psi1><psi1|):
//   F_b = |<phi_b|psi_b>|^2,  L = sum_b (1 - F_b),  lambda_b = dL/dpsi_b^* * 2 = -2 phi_b <phi_b|psi_b>.
template <class T>
__global__ void __launch_bounds__(256) colcdot_kernel(const typename Cx<T>::type* __restrict__ X, long ldx,
                                                      const typename Cx<T>::type* __restrict__ Y, long ldy, int d,
                                                      int B, int rows_per_chunk, double* __restrict__ part) {
  typedef typename Cx<T>::type T2;
  __shared__ double red[2][4][64];
  const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int b = blockIdx.x * 64 + cl;
  const int r0 = blockIdx.y * rows_per_chunk;
  const int r1 = min(d, r0 + rows_per_chunk);
  double sx = 0.0, sy = 0.0;
  if (b < B)
    for (int i = r0 + rg; i < r1; i += 4) {
      const T2 x = X[(long)i * ldx + b];
      const T2 y = Y[(long)i * ldy + b];
      sx += (double)x.x * (double)y.x + (double)x.y * (double)y.y;
      sy += (double)x.x * (double)y.y - (double)x.y * (double)y.x;
    }
  red[0][rg][cl] = sx;
  red[1][rg][cl] = sy;
  __syncthreads();
  if (rg == 0 && b < B) {
    part[((long)blockIdx.y * B + b) * 2 + 0] = red[0][0][cl] + red[0][1][cl] + red[0][2][cl] + red[0][3][cl];
    part[((long)blockIdx.y * B + b) * 2 + 1] = red[1][0][cl] + red[1][1][cl] + red[1][2][cl] + red[1][3][cl];
  }
}

// c[b] = sum of chunks; fid[b] = |c[b]|^2
__global__ void __launch_bounds__(256) colcdot_finish_kernel(const double* __restrict__ part, int nchunks, int B,
                                                             double* __restrict__ cvals, double* __restrict__ fid) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double sx = 0.0, sy = 0.0;
  for (int c = 0; c < nchunks; ++c) { sx += part[((long)c * B + b) * 2]; sy += part[((long)c * B + b) * 2 + 1]; }
  cvals[2 * b] = sx;
  cvals[2 * b + 1] = sy;
  fid[b] = sx * sx + sy * sy;
}

// lam[i][b] = scale * phi[i][b] * c[b]
template <class T>
__global__ void __launch_bounds__(256) scale_columns_kernel(const typename Cx<T>::type* __restrict__ phi,
                                                            const double* __restrict__ cvals, double scale,
                                                            typename Cx<T>::type* __restrict__ lam, long n_elem, int B) {
  typedef typename Cx<T>::type T2;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n_elem; e += (long)gridDim.x * blockDim.x) {
    const int b = (int)(e % B);
    const T cx = (T)(scale * cvals[2 * b]), cy = (T)(scale * cvals[2 * b + 1]);
    const T2 p = phi[e];
    T2 o;
    o.x = p.x * cx - p.y * cy;
    o.y = p.x * cy + p.y * cx;
    lam[e] = o;
  }
}

// part: coldot_chunks(d) * B * 2 doubles; cvals: 2 B doubles; fid: B doubles; lam = scale * phi .* c (may be null)
template <class T>
hipError_t fidelity_costate_launch(const typename Cx<T>::type* phi, const typename Cx<T>::type* psi, int d, int B,
                                   double scale, double* part, double* cvals, double* fid,
                                   typename Cx<T>::type* lam, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  const int nch = coldot_chunks(d);
  const int rpc = (d + nch - 1) / nch;
  hipLaunchKernelGGL(colcdot_kernel<T>, dim3((unsigned)((B + 63) / 64), (unsigned)nch), dim3(256), 0, st, phi, (long)B,
                     psi, (long)B, d, B, rpc, part);
  hipLaunchKernelGGL(colcdot_finish_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, st, part, nch, B, cvals, fid);
  if (lam) {
    const long n_elem = (long)d * B;
    long nb = (n_elem + 255) / 256;
    if (nb > 65536) nb = 65536;
    hipLaunchKernelGGL(scale_columns_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, phi, cvals, scale, lam, n_elem, B);
  }
  return hipGetLastError();
}
template hipError_t fidelity_costate_launch<double>(const double2*, const double2*, int, int, double, double*, double*,
                                                    double*, double2*, hipStream_t);
template hipError_t fidelity_costate_launch<float>(const float2*, const float2*, int, int, double, double*, double*,
                                                   double*, float2*, hipStream_t);

// ---- shifted states of the reference's Monte-Carlo estimator -------------------------------------------------
// sim_plain.py:194-199: gate_+- = (I +- i r H_k) / sqrt(1 + r^2) applied to phi. HP[k][i][b] = (H_k phi_b)_i;
// out[i][(b K + k) 2 + s] = (phi[i][b] + sign_s i r HP[k][i][b]) / sqrt(1 + r^2), s = 0: +, s = 1: -.
template <class T>
__global__ void __launch_bounds__(256) mc_shift_kernel(const typename Cx<T>::type* __restrict__ phi,
                                                       const typename Cx<T>::type* __restrict__ HP, int d, int B, int K,
                                                       double r_, typename Cx<T>::type* __restrict__ out) {
  typedef typename Cx<T>::type T2;
  const long n = (long)d * B * K;
  const T r = (T)r_, inv = (T)(1.0 / sqrt(1.0 + r_ * r_));
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const int k = (int)(e % K);
    const long ib = e / K;          // i * B + b
    const int b = (int)(ib % B);
    const long i = ib / B;
    const T2 p = phi[ib];
    const T2 h = HP[((long)k * d + i) * B + b];
    T2 plus, minus;                  // i r h = r (-h.y, h.x)
    plus.x = (p.x - r * h.y) * inv;
    plus.y = (p.y + r * h.x) * inv;
    minus.x = (p.x + r * h.y) * inv;
    minus.y = (p.y - r * h.x) * inv;
    T2* o = out + (i * (long)B * K + (long)b * K + k) * 2;
    o[0] = plus;
    o[1] = minus;
  }
}

template <class T>
hipError_t mc_shift_launch(const typename Cx<T>::type* phi, const typename Cx<T>::type* HP, int d, int B, int K, double r,
                           typename Cx<T>::type* out, hipStream_t st) {
  const long n = (long)d * B * K;
  if (n <= 0) return hipSuccess;
  long nb = (n + 255) / 256;
  if (nb > 65536) nb = 65536;
  hipLaunchKernelGGL(mc_shift_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, phi, HP, d, B, K, r, out);
  return hipGetLastError();
}
template hipError_t mc_shift_launch<double>(const double2*, const double2*, int, int, int, double, double2*, hipStream_t);
template hipError_t mc_shift_launch<float>(const float2*, const float2*, int, int, int, double, float2*, hipStream_t);

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

// two stages, fixed order: part[c][j] = sum of |X[i][j]| over the rows of chunk c; colsum[j] = sum_c part[c][j]
// (one thread per column over ALL rows was 16 workgroups on 256 CUs: 2.1 ms per 268 MB matrix at d = 4096)
constexpr int COLABS_CHUNKS = 64;

template <class T>
__global__ void __launch_bounds__(256) colabs_kernel(const typename Cx<T>::type* __restrict__ X, int d, int rows_per_chunk,
                                                     double* __restrict__ part) {
  typedef typename Cx<T>::type T2;
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d) return;
  const int i0 = blockIdx.y * rows_per_chunk, i1 = min(d, i0 + rows_per_chunk);
  double s = 0.0;
  for (int i = i0; i < i1; ++i) {
    const T2 x = X[(long)i * d + j];
    s += hypot((double)x.x, (double)x.y);
  }
  part[(long)blockIdx.y * d + j] = s;
}

__global__ void __launch_bounds__(256) colabs_finish_kernel(const double* __restrict__ part, int d, int nchunks,
                                                            double* __restrict__ cs) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= d) return;
  double s = 0.0;
  for (int c = 0; c < nchunks; ++c) s += part[(long)c * d + j];
  cs[j] = s;
}

int colabs_scratch_doubles(int d) { return COLABS_CHUNKS * d; }

template <class T>
hipError_t colabs_launch(const typename Cx<T>::type* X, int d, double* colsum, double* part, hipStream_t st) {
  const int rpc = (d + COLABS_CHUNKS - 1) / COLABS_CHUNKS;
  const int nchunks = (d + rpc - 1) / rpc;
  hipLaunchKernelGGL(colabs_kernel<T>, dim3((unsigned)((d + 255) / 256), (unsigned)nchunks), dim3(256), 0, st, X, d, rpc, part);
  hipLaunchKernelGGL(colabs_finish_kernel, dim3((unsigned)((d + 255) / 256)), dim3(256), 0, st, part, d, nchunks, colsum);
  return hipGetLastError();
}
template hipError_t colabs_launch<double>(const double2*, int, double*, double*, hipStream_t);
template hipError_t colabs_launch<float>(const float2*, int, double*, double*, hipStream_t);

template <class T>
__global__ void __launch_bounds__(256) herm_defect_kernel(const typename Cx<T>::type* __restrict__ X, int d,
                                                          unsigned long long* __restrict__ out) {
  typedef typename Cx<T>::type T2;
  const long n = (long)d * d;
  double mx = 0.0;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long)gridDim.x * blockDim.x) {
    const long i = e / d, j = e % d;
    const T2 a = X[e];
    const T2 b = X[j * d + i];
    mx = fmax(mx, fmax(fabs((double)a.x - (double)b.x), fabs((double)a.y + (double)b.y)));
  }
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_down(mx, off, 64));
  if ((threadIdx.x & 63) == 0) atomicMax(out, (unsigned long long)__double_as_longlong(mx));  // mx >= 0: order-preserving
}

template <class T>
hipError_t herm_defect_launch(const typename Cx<T>::type* X, int d, double* out, hipStream_t st) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(double), st);
  if (e != hipSuccess) return e;
  long nb = ((long)d * d + 255) / 256;
  if (nb > 1024) nb = 1024;
  hipLaunchKernelGGL(herm_defect_kernel<T>, dim3((unsigned)nb), dim3(256), 0, st, X, d, (unsigned long long*)out);
  return hipGetLastError();
}
template hipError_t herm_defect_launch<double>(const double2*, int, double*, hipStream_t);
template hipError_t herm_defect_launch<float>(const float2*, int, double*, hipStream_t);

}  // namespace dq
