// Internal declarations shared by the HIP translation units of libdiffqc_hip.so.
// Not part of the C-ABI (that is include/diffqc_hip.h).
#pragma once
#include <hip/hip_runtime.h>

namespace dq {

struct ZgemmArgs {
  const double2* A;
  const double2* B;
  double2* C;
  int M, N, K;          // op(A) is M x K, op(B) is K x N, C is M x N
  long lda, ldb, ldc;   // leading dimensions of the STORED matrices, in complex elements
  long sA, sB, sC;      // batch strides, in complex elements (0 = shared operand)
  int batch;
  double alpha, beta;
  int opA, opB;         // 0 = as stored, 1 = conjugate transpose
};

hipError_t zgemm_launch(const ZgemmArgs& g, hipStream_t st, int force_tile = 0, int alg = 0);
int zgemm_pick_tile(int M, int N, int batch);

// ---- elementwise / reduction kernels (elementwise.hip) ------------------------------
// A[z] = -i*dt*(H0 + sum_k u[z*K + k] * Hs[k])   (diffqc.cc:192-197, sim_plain.py:137-143)
hipError_t assemble_launch(double2* A, const double2* H0, const double2* Hs, const double* u,
                           int K, long n_elem, int batch, double dt, hipStream_t st);

// out[z] = (accumulate ? out[z] : 0) + cI * I + sum_{j<n_in} c[j] * X_j[z]      (d x d, batched)
struct LincombArgs {
  double2* out;
  const double2* x[6];
  double c[6];
  int n_in;
  double cI;
  int d;
  long stride;  // batch stride in complex elements (same for all operands)
  int batch;
  int accumulate;
};
hipError_t lincomb_launch(const LincombArgs& a, hipStream_t st);

// part[z][blk][k] = sum over the block's elements of Im(conj(Abar[z][e]) * Hs[k][e]);
// grad[z][k] = scale * sum_blk part   (deterministic two-stage reduction)
hipError_t grad_contract_launch(const double2* Abar, long strideA, const double2* Hs, int K, long n_elem,
                                int batch, double scale, double* part, double* grad, hipStream_t st);
int grad_contract_blocks(long n_elem);

// e[b] = scale * sum_i Re(conj(X[i][b]) * Y[i][b]);  X, Y are d x B (ld in complex elements)
hipError_t coldot_launch(const double2* X, long ldx, const double2* Y, long ldy, int d, int B, double scale,
                         double* e, hipStream_t st);
// out[0] = sum_b e[b]  (single block, deterministic)
hipError_t sum_launch(const double* e, int n, double* out, hipStream_t st);
// colsum[j] = sum_i |X[i][j]|  (for the 1-norm), d x d
hipError_t colabs_launch(const double2* X, int d, double* colsum, hipStream_t st);
// max |X - X^H| over the matrix -> out[0] (Hermiticity check)
hipError_t herm_defect_launch(const double2* X, int d, double* out, hipStream_t st);
// dst[i][0:B] = src[i][0:B] for row pitch conversion / plain copies handled by hipMemcpy2DAsync.

}  // namespace dq
