// host_linalg.hpp -- small dense FP64 routines for the one-time model preparation done at handle creation
// (GMMMapParam / GaussianMixtureModel / TrajectoryGMMMap constructors of the reference).  Row-major here.
#pragma once
#include <cmath>
#include <cstddef>
#include <vector>

namespace vcmi {
namespace la {

// inverse of a general n x n row-major matrix by Gauss-Jordan elimination with partial pivoting.
// Returns false on an exactly singular pivot.
inline bool inverse(const double *a_in, int n, double *inv) {
  std::vector<double> a(a_in, a_in + (size_t)n * n);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) inv[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int c = 0; c < n; ++c) {
    int p = c;
    double best = std::fabs(a[(size_t)c * n + c]);
    for (int r = c + 1; r < n; ++r)
      if (std::fabs(a[(size_t)r * n + c]) > best) { best = std::fabs(a[(size_t)r * n + c]); p = r; }
    if (best == 0.0) return false;
    if (p != c)
      for (int j = 0; j < n; ++j) {
        std::swap(a[(size_t)c * n + j], a[(size_t)p * n + j]);
        std::swap(inv[(size_t)c * n + j], inv[(size_t)p * n + j]);
      }
    double piv = a[(size_t)c * n + c];
    for (int j = 0; j < n; ++j) { a[(size_t)c * n + j] /= piv; inv[(size_t)c * n + j] /= piv; }
    for (int r = 0; r < n; ++r) {
      if (r == c) continue;
      double f = a[(size_t)r * n + c];
      if (f == 0.0) continue;
      for (int j = 0; j < n; ++j) {
        a[(size_t)r * n + j] -= f * a[(size_t)c * n + j];
        inv[(size_t)r * n + j] -= f * inv[(size_t)c * n + j];
      }
    }
  }
  return true;
}

// C = A * B, all n x n row-major
inline void matmul(const double *A, const double *B, int n, double *C) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) {
      double s = 0.0;
      for (int k = 0; k < n; ++k) s += A[(size_t)i * n + k] * B[(size_t)k * n + j];
      C[(size_t)i * n + j] = s;
    }
}

// Lower Cholesky factor (row-major, upper part zeroed) of the symmetric matrix whose UPPER triangle is
// given in S (row-major); the lower triangle of S is ignored -- this is Hermitian(S) of src/gmm.jl:16.
inline bool cholesky_from_upper(const double *S, int n, double *L) {
  for (size_t k = 0; k < (size_t)n * n; ++k) L[k] = 0.0;
  for (int j = 0; j < n; ++j) {
    double d = S[(size_t)j * n + j];
    for (int k = 0; k < j; ++k) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    L[(size_t)j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = S[(size_t)j * n + i];  // S[j][i], j < i: upper triangle
      for (int k = 0; k < j; ++k) s -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
      L[(size_t)i * n + j] = s / d;
    }
  }
  return true;
}

// U = inv(L) for lower-triangular L (row-major); U is lower-triangular.
inline void lower_inverse(const double *L, int n, double *U) {
  for (size_t k = 0; k < (size_t)n * n; ++k) U[k] = 0.0;
  for (int c = 0; c < n; ++c) {
    U[(size_t)c * n + c] = 1.0 / L[(size_t)c * n + c];
    for (int i = c + 1; i < n; ++i) {
      double s = 0.0;
      for (int k = c; k < i; ++k) s += L[(size_t)i * n + k] * U[(size_t)k * n + c];
      U[(size_t)i * n + c] = -s / L[(size_t)i * n + i];
    }
  }
}

// Eigen-decomposition of a symmetric n x n matrix by cyclic Jacobi rotations (n is a few tens, once per mixture at handle
// creation).  A (row-major) is destroyed: its diagonal ends as the eigenvalues; column j of V (row-major n x n) is the unit
// eigenvector of eigenvalue A[j][j].
inline void sym_eigen_jacobi(double *A, int n, double *V) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < n; ++p) {
      diag += A[(size_t)p * n + p] * A[(size_t)p * n + p];
      for (int q = p + 1; q < n; ++q) off += A[(size_t)p * n + q] * A[(size_t)p * n + q];
    }
    if (!(off > 1e-30 * diag)) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double theta = (A[(size_t)q * n + q] - A[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < n; ++k) {                 // columns p, q
          const double akp = A[(size_t)k * n + p], akq = A[(size_t)k * n + q];
          A[(size_t)k * n + p] = c * akp - sn * akq;
          A[(size_t)k * n + q] = sn * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {                 // rows p, q
          const double apk = A[(size_t)p * n + k], aqk = A[(size_t)q * n + k];
          A[(size_t)p * n + k] = c * apk - sn * aqk;
          A[(size_t)q * n + k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[(size_t)k * n + p], vkq = V[(size_t)k * n + q];
          V[(size_t)k * n + p] = c * vkp - sn * vkq;
          V[(size_t)k * n + q] = sn * vkp + c * vkq;
        }
      }
  }
}

}  // namespace la
}  // namespace vcmi
