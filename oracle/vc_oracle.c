/*
 * vc_oracle.c -- CPU oracle (plain C, FP64, single thread) for the VoiceConversion.jl hot path.
 *
 * TEST INFRASTRUCTURE ONLY: see vc_oracle.h.  Build with -O2 -ffp-contract=off (oracle/Makefile) so
 * that no multiply-add is fused -- the DTW parity contract is bit-exact on that arithmetic.
 *
 * Every function cites the reference lines it restates (paths relative to the reference root).
 * Third-party arithmetic the reference calls (Distributions.jl MvNormal logpdf via PDMats Cholesky,
 * StatsFuns.logsumexp, Base LinAlg inv, SuiteSparse `\`, sklearn.mixture) is restated from its
 * published mathematical definition; those package sources are not available offline.
 *
 * Pinning: DTW paths by test/dtw.jl:7-31 and W by test/trajectory_gmmmap.jl:1-34 (exact).  Everything
 * floating-point (fvconvert, predict_proba, trajectory solve, E-steps) is PARITY UNPINNED BY THE REFERENCE
 * -- its tests assert isfinite only and no Julia exists here -- and is pinned instead against third-party
 * code (sklearn, scipy, LAPACK, 50-digit mpmath) through the golden vectors: oracle/crosscheck.py,
 * tests/test_oracle_thirdparty.py.  Since round 3 also the GV ascent (dense numpy / scipy.sparse evaluation,
 * gvgrad and one step in 50-digit mpmath) and mc2e (frequency-domain evaluation with numpy.fft, no SPTK recursion).
 */
#include "vc_oracle.h"
#include "vc_oracle_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LOG2PI 1.8378770664093454835606594728112

/* ------------------------------------------------------------------------------------------------
 * small dense helpers (column-major, leading dimension = n)
 * ---------------------------------------------------------------------------------------------- */

/* inv(A) for a general n x n matrix: LU with partial pivoting then n solves (Julia `A^-1` on a
 * matrix = inv(A) = LAPACK getrf+getri; src/gmmmap.jl:35, src/trajectory_gmmmap.jl:27). */
static int lu_inverse(const double *Ain, int n, double *inv) {
  double *a = (double *)malloc(sizeof(double) * n * n);
  int *piv = (int *)malloc(sizeof(int) * n);
  memcpy(a, Ain, sizeof(double) * n * n);
  for (int k = 0; k < n; ++k) {
    int p = k;
    double best = fabs(a[k + n * k]);
    for (int i = k + 1; i < n; ++i)
      if (fabs(a[i + n * k]) > best) { best = fabs(a[i + n * k]); p = i; }
    piv[k] = p;
    if (best == 0.0) { free(a); free(piv); return 1; }
    if (p != k)
      for (int j = 0; j < n; ++j) { double t = a[k + n * j]; a[k + n * j] = a[p + n * j]; a[p + n * j] = t; }
    for (int i = k + 1; i < n; ++i) a[i + n * k] /= a[k + n * k];
    for (int j = k + 1; j < n; ++j) {
      double akj = a[k + n * j];
      for (int i = k + 1; i < n; ++i) a[i + n * j] -= a[i + n * k] * akj;
    }
  }
  for (int c = 0; c < n; ++c) {
    double *x = inv + (size_t)n * c;
    for (int i = 0; i < n; ++i) x[i] = (i == c) ? 1.0 : 0.0;
    for (int k = 0; k < n; ++k) { int p = piv[k]; if (p != k) { double t = x[k]; x[k] = x[p]; x[p] = t; } }
    for (int k = 0; k < n; ++k) { double xk = x[k]; for (int i = k + 1; i < n; ++i) x[i] -= a[i + n * k] * xk; }
    for (int k = n - 1; k >= 0; --k) {
      x[k] /= a[k + n * k];
      double xk = x[k];
      for (int i = 0; i < k; ++i) x[i] -= a[i + n * k] * xk;
    }
  }
  free(a); free(piv);
  return 0;
}

/* lower Cholesky factor of a symmetric matrix given in full storage; returns 1 if not PD
 * (PDMats / MvNormal constructor, src/gmm.jl:17). */
static int cholesky_lower(const double *S, int n, double *L) {
  memset(L, 0, sizeof(double) * n * n);
  for (int j = 0; j < n; ++j) {
    double d = S[j + n * j];
    for (int k = 0; k < j; ++k) d -= L[j + n * k] * L[j + n * k];
    if (!(d > 0.0)) return 1;
    d = sqrt(d);
    L[j + n * j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = S[i + n * j];
      for (int k = 0; k < j; ++k) s -= L[i + n * k] * L[j + n * k];
      L[i + n * j] = s / d;
    }
  }
  return 0;
}

/* C (n x n) = A (n x n) * B (n x n) */
static void matmul(const double *A, const double *B, int n, double *C) {
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      double s = 0.0;
      for (int k = 0; k < n; ++k) s += A[i + n * k] * B[k + n * j];
      C[i + n * j] = s;
    }
}

/* ------------------------------------------------------------------------------------------------
 * GMMMap -- src/gmmmap.jl, src/gmm.jl
 * ---------------------------------------------------------------------------------------------- */
/* struct vco_gmmmap: vc_oracle_internal.h */

vco_gmmmap *vco_gmmmap_new(const double *w, const double *mu, const double *sigma, int Dj, int M, int swap) {
  int D = Dj >> 1;                                          /* src/gmmmap.jl:70 */
  vco_gmmmap *g = (vco_gmmmap *)calloc(1, sizeof(*g));
  size_t dd = (size_t)D * D;
  g->D = D; g->M = M;
  g->w = (double *)malloc(sizeof(double) * M);
  g->mux = (double *)malloc(sizeof(double) * D * M);
  g->muy = (double *)malloc(sizeof(double) * D * M);
  g->Sxx = (double *)malloc(sizeof(double) * dd * M);
  g->Sxy = (double *)malloc(sizeof(double) * dd * M);
  g->Syx = (double *)malloc(sizeof(double) * dd * M);
  g->Syy = (double *)malloc(sizeof(double) * dd * M);
  g->A = (double *)malloc(sizeof(double) * dd * M);
  g->L = (double *)malloc(sizeof(double) * dd * M);
  g->logdet = (double *)malloc(sizeof(double) * M);
  memcpy(g->w, w, sizeof(double) * M);
  /* split_joint_gmm, src/gmmmap.jl:41-52; swap :74-78 */
  int xo = swap ? D : 0, yo = swap ? 0 : D;
  for (int m = 0; m < M; ++m) {
    for (int d = 0; d < D; ++d) {
      g->mux[d + D * m] = mu[xo + d + (size_t)Dj * m];
      g->muy[d + D * m] = mu[yo + d + (size_t)Dj * m];
    }
    const double *S = sigma + (size_t)Dj * Dj * m;
    for (int c = 0; c < D; ++c)
      for (int r = 0; r < D; ++r) {
        g->Sxx[r + D * c + dd * m] = S[(xo + r) + (size_t)Dj * (xo + c)];
        g->Sxy[r + D * c + dd * m] = S[(xo + r) + (size_t)Dj * (yo + c)];
        g->Syx[r + D * c + dd * m] = S[(yo + r) + (size_t)Dj * (xo + c)];
        g->Syy[r + D * c + dd * m] = S[(yo + r) + (size_t)Dj * (yo + c)];
      }
  }
  double *inv = (double *)malloc(sizeof(double) * dd);
  double *sym = (double *)malloc(sizeof(double) * dd);
  int bad = 0;
  for (int m = 0; m < M && !bad; ++m) {
    /* A_m = Syx_m * inv(Sxx_m) on the RAW (unsymmetrised) block, src/gmmmap.jl:35 */
    if (lu_inverse(g->Sxx + dd * m, D, inv)) { bad = 1; break; }
    matmul(g->Syx + dd * m, inv, D, g->A + dd * m);
    /* Array(Hermitian(covars[:,:,m])) mirrors the upper triangle, src/gmm.jl:16; MvNormal factorises it */
    const double *Sx = g->Sxx + dd * m;
    for (int c = 0; c < D; ++c)
      for (int r = 0; r < D; ++r) sym[r + D * c] = (r <= c) ? Sx[r + D * c] : Sx[c + D * r];
    if (cholesky_lower(sym, D, g->L + dd * m)) { bad = 1; break; }
    double ld = 0.0;
    for (int d = 0; d < D; ++d) ld += log(g->L[d + D * d + dd * m]);
    g->logdet[m] = 2.0 * ld;
  }
  free(inv); free(sym);
  if (bad) { vco_gmmmap_free(g); return NULL; }
  return g;
}

void vco_gmmmap_free(vco_gmmmap *g) {
  if (!g) return;
  free(g->w); free(g->mux); free(g->muy); free(g->Sxx); free(g->Sxy); free(g->Syx); free(g->Syy);
  free(g->A); free(g->L); free(g->logdet); free(g);
}
int vco_gmmmap_dim(const vco_gmmmap *g) { return g->D; }
int vco_gmmmap_ncomponents(const vco_gmmmap *g) { return g->M; }
void vco_gmmmap_get_A(const vco_gmmmap *g, double *A) { memcpy(A, g->A, sizeof(double) * g->D * g->D * g->M); }

/* lpr_m = logpdf(MvNormal(mux_m, Sxx_m), x) + log(w_m)  for w_m > 0 -- src/gmm.jl:25-27.
 * logpdf = -(D log 2pi + logdet)/2 - |L^-1 (x - mu)|^2 / 2 (Cholesky forward substitution).
 * Zero-weight components: the reference drops them from the vector; here they get
 * lpr = -inf so that their posterior is exactly 0 and indices do not shift. */
static void log_weighted_densities(const vco_gmmmap *g, const double *x, double *lpr, double *z) {
  int D = g->D;
  size_t dd = (size_t)D * D;
  for (int m = 0; m < g->M; ++m) {
    if (!(g->w[m] > 0.0)) { lpr[m] = -INFINITY; continue; }
    const double *L = g->L + dd * m, *mu = g->mux + (size_t)D * m;
    double q = 0.0;
    for (int i = 0; i < D; ++i) {
      double s = x[i] - mu[i];
      for (int k = 0; k < i; ++k) s -= L[i + D * k] * z[k];
      z[i] = s / L[i + D * i];
      q += z[i] * z[i];
    }
    lpr[m] = (-(D * LOG2PI + g->logdet[m]) / 2.0 - q / 2.0) + log(g->w[m]);
  }
}

/* StatsFuns.logsumexp: u = maximum(x); u + log(sum(exp(x - u)))  (src/gmm.jl:28) */
static double logsumexp(const double *x, int n) {
  double u = x[0];
  for (int i = 1; i < n; ++i) if (x[i] > u) u = x[i];
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += exp(x[i] - u);
  return u + log(s);
}

static void posterior(const vco_gmmmap *g, const double *x, double *p, double *z) {
  log_weighted_densities(g, x, p, z);
  double lse = logsumexp(p, g->M);
  for (int m = 0; m < g->M; ++m) p[m] = exp(p[m] - lse);   /* src/gmm.jl:29 */
}

/* one frame with caller-provided scratch: E (D*M), p (M), z (D), dx (D) */
static void fvconvert_core(const vco_gmmmap *g, const double *x, double *y, double *post, double *E, double *p, double *z,
                           double *dx) {
  int D = g->D, M = g->M;
  size_t dd = (size_t)D * D;
  /* Eq.(11): E[:,m] = muy[:,m] + A[:,:,m] * (x - mux[:,m])  -- src/gmmmap.jl:109-111 */
  for (int m = 0; m < M; ++m) {
    const double *A = g->A + dd * m;
    for (int i = 0; i < D; ++i) dx[i] = x[i] - g->mux[i + (size_t)D * m];
    for (int i = 0; i < D; ++i) {
      double s = 0.0;
      for (int k = 0; k < D; ++k) s += A[i + D * k] * dx[k];
      E[i + (size_t)D * m] = g->muy[i + (size_t)D * m] + s;
    }
  }
  posterior(g, x, p, z);                                     /* src/gmmmap.jl:114 */
  for (int i = 0; i < D; ++i) {                              /* E * posterior, src/gmmmap.jl:117 */
    double s = 0.0;
    for (int m = 0; m < M; ++m) s += E[i + (size_t)D * m] * p[m];
    y[i] = s;
  }
  if (post) memcpy(post, p, sizeof(double) * M);
}

/* the reference allocates its temporaries per frame (src/gmmmap.jl:105-117); so does this entry */
void vco_fvconvert(const vco_gmmmap *g, const double *x, double *y, double *post) {
  int D = g->D, M = g->M;
  double *E = (double *)malloc(sizeof(double) * D * M);
  double *p = (double *)malloc(sizeof(double) * M);
  double *z = (double *)malloc(sizeof(double) * D);
  double *dx = (double *)malloc(sizeof(double) * D);
  fvconvert_core(g, x, y, post, E, p, z, dx);
  free(E); free(p); free(z); free(dx);
}

void vco_fvconvert_batch(const vco_gmmmap *g, const double *X, int64_t T, double *Y) {
  for (int64_t t = 0; t < T; ++t) vco_fvconvert(g, X + (size_t)g->D * t, Y + (size_t)g->D * t, NULL);
}

/* The strong CPU baseline of SURVEY 8d(ii): the same per-frame arithmetic on every host core (OpenMP, frames are
 * independent); returns the number of threads used.  Built with -fopenmp (oracle/Makefile); without it: 1 thread. */
int vco_fvconvert_batch_mt(const vco_gmmmap *g, const double *X, int64_t T, double *Y) {
  int nthreads = 1;
#ifdef _OPENMP
#pragma omp parallel
  {
    int D = g->D, M = g->M;                        /* per-thread scratch, allocated once */
    double *E = (double *)malloc(sizeof(double) * D * M), *p = (double *)malloc(sizeof(double) * M);
    double *z = (double *)malloc(sizeof(double) * D), *dx = (double *)malloc(sizeof(double) * D);
#pragma omp single
    nthreads = omp_get_num_threads();
#pragma omp for schedule(static)
    for (int64_t t = 0; t < T; ++t) fvconvert_core(g, X + (size_t)D * t, Y + (size_t)D * t, NULL, E, p, z, dx);
    free(E); free(p); free(z); free(dx);
  }
#else
  vco_fvconvert_batch(g, X, T, Y);
#endif
  return nthreads;
}

void vco_predict_proba(const vco_gmmmap *g, const double *X, int64_t T, double *P) {
  double *z = (double *)malloc(sizeof(double) * g->D);
  for (int64_t t = 0; t < T; ++t) posterior(g, X + (size_t)g->D * t, P + (size_t)g->M * t, z);
  free(z);
}

/* the log-weighted densities themselves, lpr of src/gmm.jl:25-27 before the logsumexp: L is (M,T).  What the adversarial
 * test generator (oracle/adversarial.py) bisects on to place frames on decision boundaries. */
void vco_logdens(const vco_gmmmap *g, const double *X, int64_t T, double *L) {
  /* (frames are independent: OpenMP over them only shortens the generator's bisections; the arithmetic per frame is unchanged) */
  /* at most 16 threads, one per 64 frames: the generator calls this thousands of times on a few thousand frames, and a
   * parallel region over every core of a big (and shared) host costs more in barriers than the frames do */
#ifdef _OPENMP
  int nt = omp_get_max_threads();
  if (nt > 16) nt = 16;
  if ((int64_t)nt > (T + 63) / 64) nt = (int)((T + 63) / 64);
  if (nt < 1) nt = 1;
#pragma omp parallel num_threads(nt)
#endif
  {
    double *z = (double *)malloc(sizeof(double) * g->D);
#ifdef _OPENMP
#pragma omp for schedule(static)
#endif
    for (int64_t t = 0; t < T; ++t) log_weighted_densities(g, X + (size_t)g->D * t, L + (size_t)g->M * t, z);
    free(z);
  }
}

void vco_predict(const vco_gmmmap *g, const double *X, int64_t T, int64_t *idx) {
  double *z = (double *)malloc(sizeof(double) * g->D);
  double *p = (double *)malloc(sizeof(double) * g->M);
  for (int64_t t = 0; t < T; ++t) {
    posterior(g, X + (size_t)g->D * t, p, z);               /* src/gmm.jl:45 */
    int best = 0;                                            /* indmax: first maximum, src/gmm.jl:46 */
    for (int m = 1; m < g->M; ++m) if (p[m] > p[best]) best = m;
    idx[t] = best + 1;
  }
  free(z); free(p);
}

void vco_vc_frames(const vco_gmmmap *g, const double *fm, int64_t T, double *out) {
  int D = g->D;
  for (int64_t t = 0; t < T; ++t) {
    const double *col = fm + (size_t)(D + 1) * t;
    double *o = out + (size_t)(D + 1) * t;
    vco_fvconvert(g, col + 1, o + 1, NULL);                  /* src/common.jl:17-19 */
    o[0] = col[0];                                           /* power row kept, src/common.jl:23 */
  }
}

/* ------------------------------------------------------------------------------------------------
 * DTW -- src/dtw.jl
 * ---------------------------------------------------------------------------------------------- */

/* transition(d, i, j) -- src/dtw.jl:23-31 */
static double dtw_transition(int64_t i, int64_t j) {
  if (j == i + 1) return 0.0;
  if (i == j) return 1.0;
  return 2.0;
}

/* observation(d, v, i) = sumabs2(v - template[:,i]) -- src/dtw.jl:33-35; summed sequentially in d,
 * each square and each addition rounded separately (the oracle's fixed association). */
static double dtw_observation(const double *v, const double *tcol, int D) {
  double s = 0.0;
  for (int d = 0; d < D; ++d) {
    double df = v[d] - tcol[d];
    double sq = df * df;
    s = s + sq;
  }
  return s;
}

void vco_dtw_fit(const double *tmpl, int64_t S, const double *seq, int64_t T, int D, int fstep, int bstep,
                 double *cost, int64_t *bp, int64_t *path) {
  /* lazy_init!(d, S, T) -- src/dtw.jl:44-51 */
  int own_c = (cost == NULL), own_b = (bp == NULL);
  if (own_c) cost = (double *)malloc(sizeof(double) * S * (T + 1));
  if (own_b) bp = (int64_t *)malloc(sizeof(int64_t) * S * (T + 1));
  for (int64_t k = 0; k < S * (T + 1); ++k) { cost[k] = 0.0; bp[k] = 1; }
  for (int64_t i = 1; i <= S; ++i) { cost[i - 1] = (double)i; bp[i - 1] = i; }
  /* fit! main loop -- src/dtw.jl:104-125 (1-based i, j, t as in the reference) */
  for (int64_t t = 1; t <= T; ++t) {
    const double *v = seq + (size_t)D * (t - 1);
    const double *cprev = cost + (size_t)S * (t - 1);
    for (int64_t i = 1; i <= S; ++i) {
      int64_t minindex = i;
      double ocost = dtw_observation(v, tmpl + (size_t)D * (i - 1), D);
      double tcost = dtw_transition(minindex, i);
      double mincost = cprev[minindex - 1] + ocost + tcost;
      for (int64_t j = i - bstep; j <= i + fstep; ++j) {
        if (j < 1 || j > S) continue;
        double c = cprev[j - 1] + ocost + dtw_transition(j, i);
        if (c < mincost) { mincost = c; minindex = j; }
      }
      cost[(size_t)S * t + (i - 1)] = mincost;
      bp[(size_t)S * t + (i - 1)] = minindex;
    }
  }
  /* backward(d) -- src/dtw.jl:133-145 */
  if (path && T > 0) {
    const double *last = cost + (size_t)S * T;
    int64_t best = 1;
    for (int64_t i = 2; i <= S; ++i) if (last[i - 1] < last[best - 1]) best = i;   /* indmin: first minimum */
    path[T - 1] = best;
    for (int64_t i = T; i >= 2; --i) path[i - 2] = bp[(size_t)S * i + (path[i - 1] - 1)];
  }
  if (own_c) free(cost);
  if (own_b) free(bp);
}

void vco_align(const double *src, int64_t S, const double *tgt, int64_t T, int D, double *newtgt, int64_t *path_out) {
  int64_t *path = (int64_t *)malloc(sizeof(int64_t) * (T > 0 ? T : 1));
  vco_dtw_fit(src, S, tgt, T, D, /*fstep=*/0, /*bstep=*/2, NULL, NULL, path);   /* src/align.jl:16-17 */
  memset(newtgt, 0, sizeof(double) * D * S);                                     /* src/align.jl:20 */
  char *hit = (char *)calloc(S + 1, 1);
  for (int64_t k = 0; k < T; ++k) {                                              /* src/align.jl:21, later writes win */
    memcpy(newtgt + (size_t)D * (path[k] - 1), tgt + (size_t)D * k, sizeof(double) * D);
    hit[path[k]] = 1;
  }
  if (T > 0)
    for (int64_t i = path[0]; i <= path[T - 1]; ++i) {                           /* src/align.jl:25-32 */
      if (hit[i]) continue;
      if (i > 1 && i < S)
        for (int j = 0; j < D; ++j)
          newtgt[j + (size_t)D * (i - 1)] = (newtgt[j + (size_t)D * (i - 2)] + newtgt[j + (size_t)D * i]) / 2.0;
    }
  if (path_out) memcpy(path_out, path, sizeof(int64_t) * T);
  free(path); free(hit);
}

/* ------------------------------------------------------------------------------------------------
 * Trajectory conversion -- src/trajectory_gmmmap.jl, src/datasets.jl:6-13
 * ---------------------------------------------------------------------------------------------- */

int64_t vco_constructW(int D, int64_t T, int64_t *rows, int64_t *cols, double *vals) {
  int64_t n = 0;
  for (int64_t t = 1; t <= T; ++t) {                          /* compute_wt, src/trajectory_gmmmap.jl:39-53 */
    for (int d = 1; d <= D; ++d) {
      if (rows) { rows[n] = 2 * D * (t - 1) + d; cols[n] = (t - 1) * D + d; vals[n] = 1.0; }
      ++n;
    }
    for (int d = 1; d <= D; ++d) {
      if (t >= 2) { if (rows) { rows[n] = 2 * D * (t - 1) + D + d; cols[n] = (t - 2) * D + d; vals[n] = -0.5; } ++n; }
      if (t < T)  { if (rows) { rows[n] = 2 * D * (t - 1) + D + d; cols[n] = t * D + d;       vals[n] = 0.5; } ++n; }
    }
  }
  return n;
}

void vco_push_delta(const double *src, int D, int64_t T, double *out) {
  for (int64_t t = 0; t < T; ++t)                             /* repmat(src, 2), src/datasets.jl:8 */
    for (int d = 0; d < D; ++d) { out[d + (size_t)2 * D * t] = src[d + (size_t)D * t]; out[D + d + (size_t)2 * D * t] = src[d + (size_t)D * t]; }
  for (int64_t t = 1; t + 1 < T; ++t)                         /* t = 2:T-1, src/datasets.jl:9-11 */
    for (int d = 0; d < D; ++d)
      out[D + d + (size_t)2 * D * t] = -0.5 * src[d + (size_t)D * (t - 1)] + 0.5 * src[d + (size_t)D * (t + 1)];
}

struct vco_traj {
  const vco_gmmmap *g;
  int D2;        /* dim(g) = 2D (static + delta) */
  double *Dy;    /* (2D,2D,M): inv(Syy - A*Sxy), src/trajectory_gmmmap.jl:24-28 */
};

vco_traj *vco_traj_new(const vco_gmmmap *g) {
  vco_traj *t = (vco_traj *)calloc(1, sizeof(*t));
  int n = g->D;
  size_t nn = (size_t)n * n;
  t->g = g; t->D2 = n;
  t->Dy = (double *)malloc(sizeof(double) * nn * g->M);
  double *tmp = (double *)malloc(sizeof(double) * nn);
  for (int m = 0; m < g->M; ++m) {
    matmul(g->A + nn * m, g->Sxy + nn * m, n, tmp);
    for (size_t k = 0; k < nn; ++k) tmp[k] = g->Syy[nn * m + k] - tmp[k];
    if (lu_inverse(tmp, n, t->Dy + nn * m)) { free(tmp); vco_traj_free(t); return NULL; }
  }
  free(tmp);
  return t;
}
void vco_traj_free(vco_traj *t) { if (t) { free(t->Dy); free(t); } }

/* Solve P y = r with P SPD, stored as a lower band (bandwidth bw), n unknowns: band[(i-j) + (bw+1)*j] = P[i,j]. */
static int band_cholesky_solve(double *band, int64_t n, int bw, double *r) {
  int ld = bw + 1;
  for (int64_t j = 0; j < n; ++j) {
    double d = band[(size_t)ld * j];
    int64_t k0 = j - bw < 0 ? 0 : j - bw;
    for (int64_t k = k0; k < j; ++k) { double l = band[(j - k) + (size_t)ld * k]; d -= l * l; }
    if (!(d > 0.0)) return 1;
    d = sqrt(d);
    band[(size_t)ld * j] = d;
    int64_t imax = j + bw < n - 1 ? j + bw : n - 1;
    for (int64_t i = j + 1; i <= imax; ++i) {
      double s = band[(i - j) + (size_t)ld * j];
      int64_t kk = i - bw < 0 ? 0 : i - bw;
      if (kk < k0) kk = k0;
      for (int64_t k = kk; k < j; ++k) s -= band[(i - k) + (size_t)ld * k] * band[(j - k) + (size_t)ld * k];
      band[(i - j) + (size_t)ld * j] = s / d;
    }
  }
  for (int64_t i = 0; i < n; ++i) {                           /* L z = r */
    double s = r[i];
    int64_t k0 = i - bw < 0 ? 0 : i - bw;
    for (int64_t k = k0; k < i; ++k) s -= band[(i - k) + (size_t)ld * k] * r[k];
    r[i] = s / band[(size_t)ld * i];
  }
  for (int64_t i = n - 1; i >= 0; --i) {                      /* L' y = z */
    double s = r[i];
    int64_t kmax = i + bw < n - 1 ? i + bw : n - 1;
    for (int64_t k = i + 1; k <= kmax; ++k) s -= band[(k - i) + (size_t)ld * i] * r[k];
    r[i] = s / band[(size_t)ld * i];
  }
  return 0;
}

int vco_traj_fvconvert(const vco_traj *tj, const double *X, int64_t T, double *Y, int64_t *mhat_out, double *Ey_out) {
  const vco_gmmmap *g = tj->g;
  int D2 = tj->D2, D = D2 >> 1;                              /* src/trajectory_gmmmap.jl:67 */
  size_t nn = (size_t)D2 * D2;
  int64_t *mhat = (int64_t *)malloc(sizeof(int64_t) * T);
  double *Ey = (double *)malloc(sizeof(double) * D2 * T);
  vco_predict(g, X, T, mhat);                                /* eq.(37), src/trajectory_gmmmap.jl:82 */
  for (int64_t t = 0; t < T; ++t) {                          /* eq.(40), src/trajectory_gmmmap.jl:85-89 */
    int m = (int)mhat[t] - 1;
    const double *A = g->A + nn * m;
    for (int i = 0; i < D2; ++i) {
      double s = 0.0;
      for (int k = 0; k < D2; ++k) s += A[i + D2 * k] * (X[k + (size_t)D2 * t] - g->mux[k + (size_t)D2 * m]);
      Ey[i + (size_t)D2 * t] = g->muy[i + (size_t)D2 * m] + s;
    }
  }
  /* y = (W' Dy^-1 W) \ (W' Dy^-1 Ey), src/trajectory_gmmmap.jl:95-105.  W is the stencil
   * [y_t ; (y_{t+1} - y_{t-1})/2] with missing neighbours dropped (:39-53); P and r are accumulated
   * frame by frame without materialising W.  P is block-pentadiagonal, SPD. */
  int64_t n = (int64_t)D * T;
  int bw = 3 * D - 1;
  if (bw > n - 1) bw = (int)(n - 1);
  int ld = bw + 1;
  double *band = (double *)calloc((size_t)ld * n, sizeof(double));
  double *r = (double *)calloc(n, sizeof(double));
  double *gvec = (double *)malloc(sizeof(double) * D2);
#define PADD(I, J, V) do { int64_t i_ = (I), j_ = (J); if (i_ >= j_) band[(i_ - j_) + (size_t)ld * j_] += (V); } while (0)
  for (int64_t t = 0; t < T; ++t) {
    const double *Q = tj->Dy + nn * (mhat[t] - 1);           /* block of blkdiag, src/trajectory_gmmmap.jl:95 */
    for (int i = 0; i < D2; ++i) {
      double s = 0.0;
      for (int k = 0; k < D2; ++k) s += Q[i + D2 * k] * Ey[k + (size_t)D2 * t];
      gvec[i] = s;
    }
    int64_t nb[2]; double cf[2]; int nnb = 0;
    if (t >= 1) { nb[nnb] = t - 1; cf[nnb] = -0.5; ++nnb; }
    if (t + 1 < T) { nb[nnb] = t + 1; cf[nnb] = 0.5; ++nnb; }
    for (int a = 0; a < D; ++a) {
      r[t * D + a] += gvec[a];
      for (int b = 0; b < D; ++b) PADD(t * D + a, t * D + b, Q[a + D2 * b]);                  /* Qss */
    }
    for (int u = 0; u < nnb; ++u) {
      for (int a = 0; a < D; ++a) {
        r[nb[u] * D + a] += cf[u] * gvec[D + a];
        for (int b = 0; b < D; ++b) {
          PADD(t * D + a, nb[u] * D + b, cf[u] * Q[a + D2 * (D + b)]);                         /* Qsd */
          PADD(nb[u] * D + a, t * D + b, cf[u] * Q[(D + a) + D2 * b]);                         /* Qds */
        }
      }
      for (int v = 0; v < nnb; ++v)
        for (int a = 0; a < D; ++a)
          for (int b = 0; b < D; ++b)
            PADD(nb[u] * D + a, nb[v] * D + b, cf[u] * cf[v] * Q[(D + a) + D2 * (D + b)]);     /* Qdd */
    }
  }
#undef PADD
  int rc = band_cholesky_solve(band, n, bw, r);
  if (!rc) memcpy(Y, r, sizeof(double) * n);                 /* reshape(y, D, T), src/trajectory_gmmmap.jl:109 */
  if (mhat_out) memcpy(mhat_out, mhat, sizeof(int64_t) * T);
  if (Ey_out) memcpy(Ey_out, Ey, sizeof(double) * D2 * T);
  free(band); free(r); free(gvec); free(mhat); free(Ey);
  return rc;
}

int vco_vc_traj(const vco_traj *tj, const double *fm, int64_t T, int64_t L, double *out) {
  int D2 = tj->D2, D = D2 >> 1;
  int rows_in = D2 + 1, rows_out = D + 1;                    /* src/common.jl:38 */
  int rc = 0;
  for (int64_t b = 0; b < T && !rc; b += L) {                /* src/common.jl:42-57 */
    int64_t e = b + L < T ? b + L : T, len = e - b;
    double *ph = (double *)malloc(sizeof(double) * D2 * len);
    double *yy = (double *)malloc(sizeof(double) * D * len);
    for (int64_t t = 0; t < len; ++t) memcpy(ph + (size_t)D2 * t, fm + (size_t)rows_in * (b + t) + 1, sizeof(double) * D2);
    rc = vco_traj_fvconvert(tj, ph, len, yy, NULL, NULL);
    for (int64_t t = 0; t < len && !rc; ++t) memcpy(out + (size_t)rows_out * (b + t) + 1, yy + (size_t)D * t, sizeof(double) * D);
    free(ph); free(yy);
  }
  for (int64_t t = 0; t < T; ++t) out[(size_t)rows_out * t] = fm[(size_t)rows_in * t];   /* src/common.jl:60 */
  return rc;
}

/* fvpostf!(vs::VarianceScaling, src), src/gv.jl:10-15: per dimension sqrt(sigma2 / var) * (x - mean) + mean with the
 * corrected variance (Julia's var: 1/(T-1)).  src, out (D,T); in-place allowed. */
void vco_variance_scaling(const double *src, int D, int64_t T, const double *sigma2, double *out) {
  for (int d = 0; d < D; ++d) {
    double mu = 0.0, var = 0.0;
    for (int64_t t = 0; t < T; ++t) mu += src[d + (size_t)D * t];
    mu /= (double)T;
    for (int64_t t = 0; t < T; ++t) { double e = src[d + (size_t)D * t] - mu; var += e * e; }
    var /= (double)(T - 1);
    double sc = sqrt(sigma2[d] / var);
    for (int64_t t = 0; t < T; ++t) out[d + (size_t)D * t] = sc * (src[d + (size_t)D * t] - mu) + mu;
  }
}

/* fvconvert(tgv::TrajectoryGVGMMMap, X; epochs, alpha), src/trajectory_gmmmap.jl:139-168 (+ gvgrad :171-189).
 * muv (D), Sigvv (D,D).  W'D^-1 W y and W'D^-1 E are applied through the stencil, never materialised. */
int vco_trajgv_fvconvert(const vco_traj *tj, const double *X, int64_t T, const double *muv, const double *Sigvv,
                         int epochs, double alpha, double *Y) {
  int D2 = tj->D2, D = D2 >> 1;
  size_t nn = (size_t)D2 * D2;
  int64_t *mhat = (int64_t *)malloc(sizeof(int64_t) * T);
  double *Ey = (double *)malloc(sizeof(double) * D2 * T);
  double *y = (double *)malloc(sizeof(double) * D * T);
  int rc = vco_traj_fvconvert(tj, X, T, y, mhat, Ey);          /* y0, :146 */
  if (rc) { free(mhat); free(Ey); free(y); return rc; }
  double *pv = (double *)malloc(sizeof(double) * D * D);
  if (lu_inverse(Sigvv, D, pv)) { free(mhat); free(Ey); free(y); free(pv); return 2; }   /* :127 */
  vco_variance_scaling(y, D, T, muv, y);                       /* eq. (58), :152 */
  double omega = 1.0 / (2.0 * (double)T);
  double *v = (double *)malloc(sizeof(double) * D2 * T);       /* D^-1 (W y) per frame */
  double *g = (double *)malloc(sizeof(double) * D2 * T);       /* D^-1 E per frame */
  double *u = (double *)malloc(sizeof(double) * D2);
  double *mu_y = (double *)malloc(sizeof(double) * D), *gv = (double *)malloc(sizeof(double) * D);
  double *coef = (double *)malloc(sizeof(double) * D), *dy = (double *)malloc(sizeof(double) * D * T);
  for (int64_t t = 0; t < T; ++t) {
    const double *Q = tj->Dy + nn * (mhat[t] - 1);
    for (int i = 0; i < D2; ++i) {
      double s = 0.0;
      for (int k = 0; k < D2; ++k) s += Q[i + D2 * k] * Ey[k + (size_t)D2 * t];
      g[i + (size_t)D2 * t] = s;
    }
  }
  for (int ep = 0; ep < epochs; ++ep) {
    for (int64_t t = 0; t < T; ++t) {                          /* v_t = Q_t (W y)_t */
      for (int d = 0; d < D; ++d) {
        u[d] = y[d + (size_t)D * t];
        double dl = 0.0;
        if (t >= 1) dl -= 0.5 * y[d + (size_t)D * (t - 1)];
        if (t + 1 < T) dl += 0.5 * y[d + (size_t)D * (t + 1)];
        u[D + d] = dl;
      }
      const double *Q = tj->Dy + nn * (mhat[t] - 1);
      for (int i = 0; i < D2; ++i) {
        double s = 0.0;
        for (int k = 0; k < D2; ++k) s += Q[i + D2 * k] * u[k];
        v[i + (size_t)D2 * t] = s;
      }
    }
    for (int d = 0; d < D; ++d) {                              /* gvgrad, :171-189 */
      double m = 0.0, var = 0.0;
      for (int64_t t = 0; t < T; ++t) m += y[d + (size_t)D * t];
      m /= (double)T;
      for (int64_t t = 0; t < T; ++t) { double e = y[d + (size_t)D * t] - m; var += e * e; }
      mu_y[d] = m; gv[d] = var / (double)(T - 1);
    }
    for (int d = 0; d < D; ++d) {
      double s = 0.0;
      for (int j = 0; j < D; ++j) s += pv[j + D * d] * (gv[j] - muv[j]);   /* (pv' (gv - muv))_d */
      coef[d] = -2.0 / (double)T * s;
    }
    for (int64_t t = 0; t < T; ++t)
      for (int d = 0; d < D; ++d) {                            /* W' applied to v and g; :163 */
        double py = v[d + (size_t)D2 * t], r = g[d + (size_t)D2 * t];
        if (t >= 1) { py += 0.5 * v[D + d + (size_t)D2 * (t - 1)]; r += 0.5 * g[D + d + (size_t)D2 * (t - 1)]; }
        if (t + 1 < T) { py -= 0.5 * v[D + d + (size_t)D2 * (t + 1)]; r -= 0.5 * g[D + d + (size_t)D2 * (t + 1)]; }
        dy[d + (size_t)D * t] = omega * (-py + r) + coef[d] * (y[d + (size_t)D * t] - mu_y[d]);
      }
    for (size_t e = 0; e < (size_t)D * T; ++e) y[e] += alpha * dy[e];   /* eq. (52), :166 */
  }
  memcpy(Y, y, sizeof(double) * D * T);
  free(mhat); free(Ey); free(y); free(pv); free(v); free(g); free(u); free(mu_y); free(gv); free(coef); free(dy);
  return 0;
}

/* diffgmm(params), src/diffgmm.jl:9-25, on the joint parameters mu (2D,M), sigma (2D,2D,M). */
void vco_diffgmm(const double *mu, const double *sigma, int Dj, int M, double *mu_out, double *sigma_out) {
  int D = Dj >> 1;
  for (int m = 0; m < M; ++m) {
    const double *S = sigma + (size_t)Dj * Dj * m;
    double *O = sigma_out + (size_t)Dj * Dj * m;
    for (int d = 0; d < D; ++d) {
      mu_out[d + (size_t)Dj * m] = mu[d + (size_t)Dj * m];
      mu_out[D + d + (size_t)Dj * m] = mu[D + d + (size_t)Dj * m] - mu[d + (size_t)Dj * m];
    }
    for (int c = 0; c < D; ++c)
      for (int r = 0; r < D; ++r) {
        double xx = S[r + (size_t)Dj * c], xy = S[r + (size_t)Dj * (D + c)], yx = S[(D + r) + (size_t)Dj * c],
               yy = S[(D + r) + (size_t)Dj * (D + c)];
        O[r + (size_t)Dj * c] = xx;
        O[r + (size_t)Dj * (D + c)] = xy - xx;                 /* Sxy' = Sxy - Sxx */
        O[(D + c) + (size_t)Dj * r] = xy - xx;                 /* Syx' = (Sxy - Sxx)' */
        O[(D + r) + (size_t)Dj * (D + c)] = xx + yy - xy - yx;
      }
  }
}

/* mc2e(mc, alpha, len) -- call site src/align.jl:48; MelGeneralizedCepstrums (third party, not under /root/reference):
 * e = sum_n h[n]^2, h = c2ir(freqt(mc, len-1, -alpha), len); freqt / c2ir restated from the published SPTK recursions.
 * mc (D,T) -> e (T). */
void vco_mc2e(const double *mc, int D, int64_t T, double alpha, int fftlen, double *e) {
  int m2 = fftlen - 1;
  double a = -alpha, b = 1.0 - a * a;
  double *g = (double *)malloc(sizeof(double) * (m2 + 1)), *d = (double *)malloc(sizeof(double) * (m2 + 1));
  double *h = (double *)malloc(sizeof(double) * fftlen);
  for (int64_t t = 0; t < T; ++t) {
    const double *c = mc + (size_t)D * t;
    memset(g, 0, sizeof(double) * (m2 + 1));
    for (int i = D - 1; i >= 0; --i) {                        /* freqt */
      memcpy(d, g, sizeof(double) * (m2 + 1));
      g[0] = c[i] + a * d[0];
      if (m2 >= 1) g[1] = b * d[0] + a * d[1];
      for (int j = 2; j <= m2; ++j) g[j] = d[j - 1] + a * (d[j] - g[j - 1]);
    }
    h[0] = exp(g[0]);                                         /* c2ir */
    for (int n = 1; n < fftlen; ++n) {
      double s = 0.0;
      int up = n < m2 ? n : m2;
      for (int k = 1; k <= up; ++k) s += k * g[k] * h[n - k];
      h[n] = s / n;
    }
    double en = 0.0;
    for (int n = 0; n < fftlen; ++n) en += h[n] * h[n];
    e[t] = en;
  }
  free(g); free(d); free(h);
}

/* align_mcep(src, tgt, alpha, fftlen; threshold, remove_silence) -- src/align.jl:38-55.  src (D,S), tgt (D,T);
 * outputs (D, <=S) compacted; returns the number of kept columns. */
int64_t vco_align_mcep(const double *src, int64_t S, const double *tgt, int64_t T, int D, double alpha, int fftlen,
                       double threshold, int remove_silence, double *src_out, double *newtgt_out) {
  double *nt = (double *)malloc(sizeof(double) * D * S);
  vco_align(src, S, tgt, T, D, nt, NULL);
  int64_t k = 0;
  if (remove_silence) {
    double *e = (double *)malloc(sizeof(double) * S);
    vco_mc2e(src, D, S, alpha, fftlen, e);
    for (int64_t i = 0; i < S; ++i)
      if (log(e[i]) > threshold) {
        memcpy(src_out + (size_t)D * k, src + (size_t)D * i, sizeof(double) * D);
        memcpy(newtgt_out + (size_t)D * k, nt + (size_t)D * i, sizeof(double) * D);
        ++k;
      }
    free(e);
  } else {
    memcpy(src_out, src, sizeof(double) * D * S);
    memcpy(newtgt_out, nt, sizeof(double) * D * S);
    k = S;
  }
  free(nt);
  return k;
}

/* One utterance of ParallelDataset(...; joint=true), src/datasets.jl:60-84: drop row 1 (ignore0th), push_delta,
 * tgt - src (diff), vcat.  src, tgt (D,n) -> out (Dj,n), Dj = 2 (D - ignore0th) (1 + add_delta). */
void vco_joint_features(const double *src, const double *tgt, int D, int64_t n, int ignore0th, int add_delta, int diff,
                        double *out) {
  int r0 = ignore0th ? 1 : 0, Ds = D - r0, Dh = Ds * (add_delta ? 2 : 1), Dj = 2 * Dh;
  for (int half = 0; half < 2; ++half) {
    const double *x = half ? tgt : src;
    for (int64_t t = 0; t < n; ++t)
      for (int d = 0; d < Ds; ++d) {
        double v = x[(r0 + d) + (size_t)D * t];
        out[half * Dh + d + (size_t)Dj * t] = v;
        if (add_delta) {                                      /* push_delta, src/datasets.jl:6-13 */
          double dl = v;
          if (t >= 1 && t + 1 < n) dl = -0.5 * x[(r0 + d) + (size_t)D * (t - 1)] + 0.5 * x[(r0 + d) + (size_t)D * (t + 1)];
          out[half * Dh + Ds + d + (size_t)Dj * t] = dl;
        }
      }
  }
  if (diff)
    for (int64_t t = 0; t < n; ++t)
      for (int d = 0; d < Dh; ++d) out[Dh + d + (size_t)Dj * t] -= out[d + (size_t)Dj * t];
}

/* ------------------------------------------------------------------------------------------------
 * Diagonal E-step
 * ---------------------------------------------------------------------------------------------- */
void vco_estep_diag(const double *X, int64_t N, int Dj, int M, const double *w, const double *mu, const double *var,
                    double *S0, double *S1, double *S2, double *loglik) {
  double *lpr = (double *)malloc(sizeof(double) * M);
  double *cst = (double *)malloc(sizeof(double) * M);
  memset(S0, 0, sizeof(double) * M);
  memset(S1, 0, sizeof(double) * Dj * M);
  memset(S2, 0, sizeof(double) * Dj * M);
  for (int m = 0; m < M; ++m) {
    double s = 0.0;
    for (int d = 0; d < Dj; ++d) s += log(var[d + (size_t)Dj * m]);
    cst[m] = log(w[m]) - 0.5 * (Dj * LOG2PI + s);
  }
  double ll = 0.0;
  for (int64_t n = 0; n < N; ++n) {
    const double *x = X + (size_t)Dj * n;
    for (int m = 0; m < M; ++m) {
      double q = 0.0;
      for (int d = 0; d < Dj; ++d) {
        double df = x[d] - mu[d + (size_t)Dj * m];
        q += df * df / var[d + (size_t)Dj * m];
      }
      lpr[m] = cst[m] - 0.5 * q;
    }
    double lse = logsumexp(lpr, M);
    ll += lse;
    for (int m = 0; m < M; ++m) {
      double gam = exp(lpr[m] - lse);
      S0[m] += gam;
      for (int d = 0; d < Dj; ++d) {
        S1[d + (size_t)Dj * m] += gam * x[d];
        S2[d + (size_t)Dj * m] += gam * x[d] * x[d];
      }
    }
  }
  *loglik = ll;
  free(lpr); free(cst);
}

/* ------------------------------------------------------------------------------------------------
 * Full-covariance E-step: what sklearn.mixture.GMM(covariance_type="full").fit computes per EM
 * iteration (call site bin/train_gmm.jl:84-103; third-party arithmetic restated from its definition).
 * X (Dj,N); w (M); mu (Dj,M); sigma (Dj,Dj,M) column-major.  S0 (M), S1 (Dj,M), S2 (Dj,Dj,M), loglik.
 * Returns nonzero if a covariance is not positive definite.
 * ---------------------------------------------------------------------------------------------- */
int vco_estep_full(const double *X, int64_t N, int Dj, int M, const double *w, const double *mu, const double *sigma,
                   double *S0, double *S1, double *S2, double *loglik) {
  size_t dd = (size_t)Dj * Dj;
  double *L = (double *)malloc(sizeof(double) * dd * M);
  double *cst = (double *)malloc(sizeof(double) * M);
  double *lpr = (double *)malloc(sizeof(double) * M);
  double *z = (double *)malloc(sizeof(double) * Dj);
  double *sym = (double *)malloc(sizeof(double) * dd);
  int bad = 0;
  for (int m = 0; m < M && !bad; ++m) {
    const double *S = sigma + dd * m;
    for (int c = 0; c < Dj; ++c)
      for (int r = 0; r < Dj; ++r) sym[r + (size_t)Dj * c] = (r <= c) ? S[r + (size_t)Dj * c] : S[c + (size_t)Dj * r];
    if (cholesky_lower(sym, Dj, L + dd * m)) { bad = 1; break; }
    double ld = 0.0;
    for (int d = 0; d < Dj; ++d) ld += log(L[d + (size_t)Dj * d + dd * m]);
    cst[m] = (w[m] > 0.0 ? log(w[m]) : -INFINITY) - 0.5 * (Dj * LOG2PI + 2.0 * ld);
  }
  if (!bad) {
    memset(S0, 0, sizeof(double) * M);
    memset(S1, 0, sizeof(double) * Dj * M);
    memset(S2, 0, sizeof(double) * dd * M);
    double ll = 0.0;
    for (int64_t n = 0; n < N; ++n) {
      const double *x = X + (size_t)Dj * n;
      for (int m = 0; m < M; ++m) {
        const double *Lm = L + dd * m, *mm = mu + (size_t)Dj * m;
        double q = 0.0;
        for (int i = 0; i < Dj; ++i) {
          double s = x[i] - mm[i];
          for (int k = 0; k < i; ++k) s -= Lm[i + (size_t)Dj * k] * z[k];
          z[i] = s / Lm[i + (size_t)Dj * i];
          q += z[i] * z[i];
        }
        lpr[m] = cst[m] - 0.5 * q;
      }
      double lse = logsumexp(lpr, M);
      ll += lse;
      for (int m = 0; m < M; ++m) {
        double gam = exp(lpr[m] - lse);
        if (gam == 0.0) continue;
        S0[m] += gam;
        double *s2 = S2 + dd * m;
        for (int j = 0; j < Dj; ++j) {
          double gx = gam * x[j];
          S1[j + (size_t)Dj * m] += gx;
          for (int i = 0; i < Dj; ++i) s2[i + (size_t)Dj * j] += gx * x[i];
        }
      }
    }
    *loglik = ll;
  }
  free(L); free(cst); free(lpr); free(z); free(sym);
  return bad;
}
