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

}  // namespace la
}  // namespace vcmi
