/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product; never linked into libdiffqc_hip.so.
 *
 * Plain-C (C99, single thread) restatement of the reference's native module
 * /root/reference/diffqc.cc: pulse envelope f_u / bspline / my_expit (diffqc.cc:75-135) and the
 * time stepper trotter() (diffqc.cc:173-205).
 *
 * The reference's exp() is Eigen's `unsupported/Eigen/MatrixFunctions` (diffqc.cc:8,198), a third-party
 * dependency that is NOT in /root/reference (Eigen3 >= 3.3.3, unpinned, CMakeLists.txt:26) and not in this
 * image, so oracle/_ref cannot be built ("unbuildable here").  Eigen implements Higham's 2005 scaling and
 * squaring method: Pade approximant of degree 3/5/7/9/13 chosen by the 1-norm, solved with partial-pivot LU,
 * then repeated squaring; dqo_expm below restates that published algorithm (N. J. Higham, "The scaling and
 * squaring method for the matrix exponential revisited", SIAM J. Matrix Anal. Appl. 26(4), 2005) with the
 * Pade coefficients generated from their closed form b_j = (2m-j)! m! / ((2m)! j! (m-j)!).
 * Parity status: pinned indirectly -- tests check it against scipy.linalg.expm and against the numpy oracle
 * whose sim_plain path is pinned on goldens from the reference; the f_u carrier path has no reference vectors
 * ("parity unpinned" for that part, see oracle/trotter_oracle.py).
 *
 * Layout: complex numbers are (re, im) double pairs; matrices are row-major d x d.
 */
#include <complex.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef double complex cplx;

/* ---- pulse model --------------------------------------------------------------------------------- */

double dqo_expit(double x) { /* diffqc.cc:75-80 */
  const double cutoff = 32.;
  if (x > cutoff) return 1.;
  if (x < -cutoff) return 0.;
  return 1 / (1 + exp(-x));
}

double dqo_bspline(int b, int n_basis, double t) { /* diffqc.cc:82-93 */
  double tau = 1. / (n_basis - 2.);
  double tau_b = tau * (b - 1.5);
  double l = tau_b - 1.5 * tau;
  double r = tau_b + 1.5 * tau;
  double ans = 0.;
  if (t < r && t > l) {
    double norm_factor = -(1.5 * tau) * (1.5 * tau);
    ans = (t - l) * (t - r) / norm_factor;
  }
  return ans;
}

double dqo_legendre(int n, double x) { /* std::legendre, diffqc.cc:117 */
  if (n == 0) return 1.0;
  double p0 = 1.0, p1 = x;
  for (int k = 1; k < n; ++k) {
    double p2 = ((2 * k + 1) * x * p1 - k * p0) / (k + 1);
    p0 = p1;
    p1 = p2;
  }
  return p1;
}

/* channels: rows of 4 doubles (unused, omega, w, idx); chan_off[h]..chan_off[h+1] are the rows of control h.
 * vv: [2][n_param][n_basis].                                                       diffqc.cc:95-135 */
double dqo_f_u(int h, double t, const double* vv, int n_param, int n_basis, const int* chan_off,
               const double* channels, double duration, int func_type) {
  double ans = 0;
  for (int ic = chan_off[h]; ic < chan_off[h + 1]; ++ic) {
    const double* chan = channels + 4 * ic;
    double A = 0, B = 0, N = 0;
    double omega = chan[1], w = chan[2];
    int idx = (int)round(chan[3]);
    for (int j = 0; j < n_basis; ++j) {
      double fv = (func_type == 0) ? dqo_legendre(j, 2 * t / duration - 1) : dqo_bspline(j, n_basis, t / duration);
      A += vv[(0 * n_param + idx) * n_basis + j] * fv;
      B += vv[(1 * n_param + idx) * n_basis + j] * fv;
    }
    N = sqrt(A * A + B * B);
    if (fabs(N - 0.0) < 0.000001) {
      ans += 0.0;
    } else {
      ans += omega * (2 * dqo_expit(N) - 1) / N * (cos(w * t) * A + sin(w * t) * B);
    }
  }
  return ans;
}

/* ---- dense helpers ------------------------------------------------------------------------------- */

static void matmul(int d, const cplx* A, const cplx* B, cplx* C) {
  for (int i = 0; i < d; ++i) {
    cplx* c = C + (size_t)i * d;
    for (int j = 0; j < d; ++j) c[j] = 0;
    for (int k = 0; k < d; ++k) {
      const cplx a = A[(size_t)i * d + k];
      const cplx* b = B + (size_t)k * d;
      for (int j = 0; j < d; ++j) c[j] += a * b[j];
    }
  }
}

static double norm1(int d, const cplx* A) {
  double mx = 0;
  for (int j = 0; j < d; ++j) {
    double s = 0;
    for (int i = 0; i < d; ++i) s += cabs(A[(size_t)i * d + j]);
    if (s > mx) mx = s;
  }
  return mx;
}

/* solve P X = Q in place (X overwrites Q), partial-pivot LU (Eigen partialPivLu().solve) */
static int lu_solve(int d, cplx* P, cplx* Q) {
  for (int k = 0; k < d; ++k) {
    int piv = k;
    double best = cabs(P[(size_t)k * d + k]);
    for (int i = k + 1; i < d; ++i) {
      double v = cabs(P[(size_t)i * d + k]);
      if (v > best) { best = v; piv = i; }
    }
    if (best == 0.0) return 1;
    if (piv != k) {
      for (int j = 0; j < d; ++j) {
        cplx t = P[(size_t)k * d + j]; P[(size_t)k * d + j] = P[(size_t)piv * d + j]; P[(size_t)piv * d + j] = t;
        t = Q[(size_t)k * d + j]; Q[(size_t)k * d + j] = Q[(size_t)piv * d + j]; Q[(size_t)piv * d + j] = t;
      }
    }
    for (int i = k + 1; i < d; ++i) {
      cplx f = P[(size_t)i * d + k] / P[(size_t)k * d + k];
      if (f == 0) continue;
      for (int j = k + 1; j < d; ++j) P[(size_t)i * d + j] -= f * P[(size_t)k * d + j];
      for (int j = 0; j < d; ++j) Q[(size_t)i * d + j] -= f * Q[(size_t)k * d + j];
    }
  }
  for (int i = d - 1; i >= 0; --i) {
    for (int j = 0; j < d; ++j) {
      cplx s = Q[(size_t)i * d + j];
      for (int k = i + 1; k < d; ++k) s -= P[(size_t)i * d + k] * Q[(size_t)k * d + j];
      Q[(size_t)i * d + j] = s / P[(size_t)i * d + i];
    }
  }
  return 0;
}

/* U = exp(A): Higham 2005 Algorithm 2.3 (what Eigen's MatrixBase::exp() runs for double). */
int dqo_expm(int d, const double* A_, double* U_) {
  const cplx* A0 = (const cplx*)A_;
  cplx* Uout = (cplx*)U_;
  const size_t nn = (size_t)d * d;
  static const int degs[5] = {3, 5, 7, 9, 13};
  static const double theta[5] = {1.495585217958292e-2, 2.539398330063230e-1, 9.504178996162932e-1,
                                  2.097847961257068e0, 5.371920351148152e0};
  double nrm = norm1(d, A0);
  int m = 13, s = 0;
  for (int i = 0; i < 4; ++i)
    if (nrm <= theta[i]) { m = degs[i]; break; }
  if (m == 13 && nrm > theta[4]) {
    s = (int)ceil(log2(nrm / theta[4]));
    if (s < 0) s = 0;
  }
  cplx* A = (cplx*)malloc(nn * sizeof(cplx));
  cplx* A2 = (cplx*)malloc(nn * sizeof(cplx));
  cplx* X = (cplx*)malloc(nn * sizeof(cplx));  /* running even power */
  cplx* T = (cplx*)malloc(nn * sizeof(cplx));
  cplx* Uo = (cplx*)calloc(nn, sizeof(cplx));  /* sum of odd terms before the final A* */
  cplx* V = (cplx*)calloc(nn, sizeof(cplx));
  if (!A || !A2 || !X || !T || !Uo || !V) return 2;
  const double sc = ldexp(1.0, -s);
  for (size_t i = 0; i < nn; ++i) A[i] = A0[i] * sc;
  /* b_j */
  double b[14];
  b[0] = 1.0;
  for (int j = 1; j <= m; ++j) b[j] = b[j - 1] * (double)(m - j + 1) / ((double)j * (double)(2 * m - j + 1));
  matmul(d, A, A, A2);
  /* even powers: V = sum b_{2k} A^{2k}, W = sum b_{2k+1} A^{2k}; U = A W  (straightforward evaluation;
     Higham groups the degree-13 case differently to save products, the value is the same polynomial) */
  for (size_t i = 0; i < nn; ++i) X[i] = 0;
  for (int i = 0; i < d; ++i) X[(size_t)i * d + i] = 1.0;
  for (int k = 0; 2 * k <= m; ++k) {
    for (size_t i = 0; i < nn; ++i) {
      V[i] += b[2 * k] * X[i];
      if (2 * k + 1 <= m) Uo[i] += b[2 * k + 1] * X[i];
    }
    if (2 * (k + 1) <= m) {
      matmul(d, X, A2, T);
      memcpy(X, T, nn * sizeof(cplx));
    }
  }
  matmul(d, A, Uo, T); /* T = U (odd part) */
  /* (V - U) R = (V + U) */
  for (size_t i = 0; i < nn; ++i) {
    X[i] = V[i] - T[i];
    Uo[i] = V[i] + T[i];
  }
  int rc = lu_solve(d, X, Uo);
  for (int j = 0; j < s && rc == 0; ++j) {
    matmul(d, Uo, Uo, T);
    memcpy(Uo, T, nn * sizeof(cplx));
  }
  memcpy(Uout, Uo, nn * sizeof(cplx));
  free(A); free(A2); free(X); free(T); free(Uo); free(V);
  return rc;
}

/* psi <- exp(dH_t) psi for S steps; u is [S][K]; psi is [d][B] row-major.        diffqc.cc:190-200 */
int dqo_propagate(int d, int K, const double* H0_, const double* Hs_, const double* u, int S, double dt,
                  double* psi_, int B) {
  const cplx* H0 = (const cplx*)H0_;
  const cplx* Hs = (const cplx*)Hs_;
  cplx* psi = (cplx*)psi_;
  const size_t nn = (size_t)d * d;
  cplx* dH = (cplx*)malloc(nn * sizeof(cplx));
  cplx* U = (cplx*)malloc(nn * sizeof(cplx));
  cplx* tmp = (cplx*)malloc((size_t)d * B * sizeof(cplx));
  if (!dH || !U || !tmp) return 2;
  const cplx i_unit = I;
  int rc = 0;
  for (int step = 0; step < S && rc == 0; ++step) {
    for (size_t i = 0; i < nn; ++i) dH[i] = -i_unit * dt * H0[i];
    for (int h = 0; h < K; ++h) {
      const double coeff = u[(size_t)step * K + h];
      const cplx* Hh = Hs + (size_t)h * nn;
      for (size_t i = 0; i < nn; ++i) dH[i] += -i_unit * dt * coeff * Hh[i];
    }
    rc = dqo_expm(d, (const double*)dH, (double*)U);
    for (int i = 0; i < d; ++i)
      for (int b = 0; b < B; ++b) {
        cplx s = 0;
        for (int k = 0; k < d; ++k) s += U[(size_t)i * d + k] * psi[(size_t)k * B + b];
        tmp[(size_t)i * B + b] = s;
      }
    memcpy(psi, tmp, (size_t)d * B * sizeof(cplx));
  }
  free(dH); free(U); free(tmp);
  return rc;
}

/* diffqc.trotter after diffqc.set_H (diffqc.cc:43-73,173-205), single state. psi is in/out. */
int dqo_trotter(int d, int K, const double* H0, const double* Hs, const int* chan_off, const double* channels,
                double duration, int func_type, const double* vv, int n_param, int n_basis, double* psi, double T0,
                double T, int per_step) {
  int n_steps = (int)(per_step * (fabs(T - T0) + 1));
  if (n_steps <= 0) return 3;
  double dt = (T - T0) / n_steps;
  double t = T0;
  double* u = (double*)malloc((size_t)n_steps * (K > 0 ? K : 1) * sizeof(double));
  if (!u) return 2;
  for (int step = 0; step < n_steps; ++step) {
    for (int h = 0; h < K; ++h)
      u[(size_t)step * K + h] = dqo_f_u(h, t, vv, n_param, n_basis, chan_off, channels, duration, func_type);
    t += dt;
  }
  int rc = dqo_propagate(d, K, H0, Hs, u, n_steps, dt, psi, 1);
  free(u);
  return rc;
}
