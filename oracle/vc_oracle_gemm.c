/*
 * vc_oracle_gemm.c -- the strong CPU baseline of SURVEY 8d(ii): GMMMap fvconvert (src/gmmmap.jl:101-118 over the frame
 * loop of src/common.jl:17-19) with "the same math restructured as batched GEMMs with OpenMP over all host cores".
 *
 * TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg and tests/): never linked into libvcmi.so.  It is NOT a parity
 * reference -- it is itself checked against the per-frame oracle (vco_fvconvert_batch, <= 1e-12 relative per frame,
 * tests/test_oracle_golden.py) -- and exists so that the GPU/CPU ratio in the bench line has an honest denominator.
 *
 * Structure (every mixture evaluated for every frame: 3 D^2 M flop per frame, the SURVEY 8(d) count with the whitening taken
 * as the triangular product it is):
 *   per block of NB = 32 frames (OpenMP over blocks; the block's operands live in L1):
 *     Xt (D x NB)             the frames transposed, so that a SIMD vector holds one feature of eight frames
 *     pass 1, per mixture:    Z = inv(L_m) Xt - inv(L_m) mu^x_m   (lower-triangular GEMM, 4 x 32 register tile, FMA)
 *                             l_m = c_m - |z|^2 / 2                (src/gmm.jl:25-27)
 *     softmax over m          (src/gmm.jl:28-29; exp skipped where it underflows to exactly 0)
 *     pass 2, per mixture:    Y += p_m * (A_m Xt + b_m),  b_m = mu^y_m - A_m mu^x_m     (src/gmmmap.jl:109-117)
 * inv(L_m), A_m row-major and padded to a multiple of four rows, prepared once per call (M D^3 / 3 flop: negligible).
 * Built -O3 -ffp-contract=fast (oracle/Makefile: this file only -- vc_oracle.c stays unfused for the DTW contract) with
 * function clones for AVX-512 / AVX2+FMA / baseline x86-64 picked at load time, so the library built in the container
 * runs on whatever the GPU box's host is.
 */
#include "vc_oracle_internal.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define LOG2PI 1.8378770664093454835606594728112
#define NB 32                     /* frames per block */
#define NV (NB / 8)               /* 8-double vectors per feature row of a block */
#define MR 4                      /* rows per register tile */

/* (gcc 11 cannot see that the two-level initialisation loops of the register tiles cover every element) */
#pragma GCC diagnostic ignored "-Wmaybe-uninitialized"

typedef double v8d __attribute__((vector_size(64), aligned(64)));

typedef struct {
  int D, D4, M;
  double *Li;                     /* (M, D4, D) row-major inv(L_m), zero above the diagonal and in the padding rows */
  double *A;                      /* (M, D4, D) row-major A_m */
  double *cz;                     /* (M, D4) inv(L_m) mu^x_m */
  double *b;                      /* (M, D4) mu^y_m - A_m mu^x_m */
  double *c;                      /* (M) log w_m - (D log 2pi + logdet_m) / 2; -inf for w_m = 0 */
} gemm_plan;

static void plan_free(gemm_plan *p) {
  free(p->Li); free(p->A); free(p->cz); free(p->b); free(p->c);
}

static int plan_make(const vco_gmmmap *g, gemm_plan *p) {
  const int D = g->D, M = g->M, D4 = (D + MR - 1) / MR * MR;
  const size_t dd = (size_t)D * D;
  p->D = D; p->D4 = D4; p->M = M;
  p->Li = (double *)calloc((size_t)M * D4 * D, sizeof(double));
  p->A = (double *)calloc((size_t)M * D4 * D, sizeof(double));
  p->cz = (double *)calloc((size_t)M * D4, sizeof(double));
  p->b = (double *)calloc((size_t)M * D4, sizeof(double));
  p->c = (double *)calloc((size_t)M, sizeof(double));
  if (!p->Li || !p->A || !p->cz || !p->b || !p->c) { plan_free(p); return 1; }
  for (int m = 0; m < M; ++m) {
    const double *L = g->L + dd * m;                 /* column-major lower factor */
    double *Li = p->Li + (size_t)m * D4 * D;
    /* column j of inv(L): forward substitution on e_j */
    for (int j = 0; j < D; ++j) {
      for (int i = j; i < D; ++i) {
        double s = (i == j) ? 1.0 : 0.0;
        for (int k = j; k < i; ++k) s -= L[i + (size_t)D * k] * Li[(size_t)k * D + j];
        Li[(size_t)i * D + j] = s / L[i + (size_t)D * i];
      }
    }
    const double *Am = g->A + dd * m, *mux = g->mux + (size_t)D * m, *muy = g->muy + (size_t)D * m;
    double *A = p->A + (size_t)m * D4 * D;
    for (int i = 0; i < D; ++i) {
      double sz = 0.0, sa = 0.0;
      for (int k = 0; k < D; ++k) {
        A[(size_t)i * D + k] = Am[i + (size_t)D * k];
        sz += Li[(size_t)i * D + k] * mux[k];
        sa += Am[i + (size_t)D * k] * mux[k];
      }
      p->cz[(size_t)m * D4 + i] = sz;
      p->b[(size_t)m * D4 + i] = muy[i] - sa;
    }
    p->c[m] = (g->w[m] > 0.0) ? log(g->w[m]) - (D * LOG2PI + g->logdet[m]) / 2.0 : -INFINITY;
  }
  return 0;
}

/* One block of NB frames (nf <= NB valid).  Xt, Yt: (D4, NB); lw: (M, NB).  Cloned per ISA; everything is in this one function
 * so that the clones carry their own vector code. */
__attribute__((target_clones("avx512f", "arch=haswell", "default")))
static void convert_block(const gemm_plan *p, const double *X, int nf, double *Y, double *Xt_, double *Yt_, double *lw_) {
  const int D = p->D, D4 = p->D4, M = p->M;
  v8d *Xt = (v8d *)Xt_, *Yt = (v8d *)Yt_, *lw = (v8d *)lw_;
  for (int k = 0; k < D; ++k)
    for (int b = 0; b < NB; ++b) Xt_[(size_t)k * NB + b] = (b < nf) ? X[(size_t)b * D + k] : 0.0;
  /* pass 1: log-weighted densities */
  for (int m = 0; m < M; ++m) {
    const double *Li = p->Li + (size_t)m * D4 * D, *cz = p->cz + (size_t)m * D4;
    v8d q[NV];
    for (int v = 0; v < NV; ++v) q[v] = (v8d){0, 0, 0, 0, 0, 0, 0, 0};
    if (p->c[m] == -INFINITY) {
      for (int v = 0; v < NV; ++v) lw[(size_t)m * NV + v] = (v8d){-INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY, -INFINITY};
      continue;
    }
    for (int i0 = 0; i0 < D4; i0 += MR) {
      v8d acc[MR][NV];
      for (int r = 0; r < MR; ++r)
        for (int v = 0; v < NV; ++v) { const double z = -cz[i0 + r]; acc[r][v] = (v8d){z, z, z, z, z, z, z, z}; }
      const int kend = (i0 + MR < D) ? i0 + MR : D;          /* lower triangular: columns <= the tile's last row */
      const double *l0 = Li + (size_t)i0 * D;
      for (int k = 0; k < kend; ++k) {
        const v8d x0 = Xt[(size_t)k * NV + 0], x1 = Xt[(size_t)k * NV + 1], x2 = Xt[(size_t)k * NV + 2], x3 = Xt[(size_t)k * NV + 3];
        for (int r = 0; r < MR; ++r) {
          const double a = l0[(size_t)r * D + k];
          acc[r][0] += a * x0; acc[r][1] += a * x1; acc[r][2] += a * x2; acc[r][3] += a * x3;
        }
      }
      for (int r = 0; r < MR; ++r)
        for (int v = 0; v < NV; ++v) q[v] += acc[r][v] * acc[r][v];
    }
    const double c = p->c[m];
    for (int v = 0; v < NV; ++v) lw[(size_t)m * NV + v] = c - 0.5 * q[v];
  }
  /* softmax over the mixtures (StatsFuns.logsumexp: max-shifted), per frame */
  for (int b = 0; b < NB; ++b) {
    double u = -INFINITY, s = 0.0;
    for (int m = 0; m < M; ++m) { const double l = lw_[(size_t)m * NB + b]; if (l > u) u = l; }
    for (int m = 0; m < M; ++m) { const double d = lw_[(size_t)m * NB + b] - u; if (d > -746.0) s += exp(d); }
    const double lse = u + log(s);
    for (int m = 0; m < M; ++m) { const double d = lw_[(size_t)m * NB + b] - lse; lw_[(size_t)m * NB + b] = (d > -746.0) ? exp(d) : 0.0; }
  }
  /* pass 2: y = sum_m p_m (A_m x + b_m) */
  for (int i = 0; i < D4 * NV; ++i) Yt[i] = (v8d){0, 0, 0, 0, 0, 0, 0, 0};
  for (int m = 0; m < M; ++m) {
    const double *A = p->A + (size_t)m * D4 * D, *bm = p->b + (size_t)m * D4;
    const v8d p0 = lw[(size_t)m * NV + 0], p1 = lw[(size_t)m * NV + 1], p2 = lw[(size_t)m * NV + 2], p3 = lw[(size_t)m * NV + 3];
    for (int i0 = 0; i0 < D4; i0 += MR) {
      v8d acc[MR][NV];
      for (int r = 0; r < MR; ++r)
        for (int v = 0; v < NV; ++v) { const double z = bm[i0 + r]; acc[r][v] = (v8d){z, z, z, z, z, z, z, z}; }
      const double *a0 = A + (size_t)i0 * D;
      for (int k = 0; k < D; ++k) {
        const v8d x0 = Xt[(size_t)k * NV + 0], x1 = Xt[(size_t)k * NV + 1], x2 = Xt[(size_t)k * NV + 2], x3 = Xt[(size_t)k * NV + 3];
        for (int r = 0; r < MR; ++r) {
          const double a = a0[(size_t)r * D + k];
          acc[r][0] += a * x0; acc[r][1] += a * x1; acc[r][2] += a * x2; acc[r][3] += a * x3;
        }
      }
      for (int r = 0; r < MR; ++r) {
        v8d *y = Yt + (size_t)(i0 + r) * NV;
        y[0] += p0 * acc[r][0]; y[1] += p1 * acc[r][1]; y[2] += p2 * acc[r][2]; y[3] += p3 * acc[r][3];
      }
    }
  }
  for (int b = 0; b < nf; ++b)
    for (int k = 0; k < D; ++k) Y[(size_t)b * D + k] = Yt_[(size_t)k * NB + b];
}

int vco_fvconvert_batch_gemm(const vco_gmmmap *g, const double *X, int64_t T, double *Y) {
  gemm_plan p;
  if (plan_make(g, &p)) return 0;
  int nthreads = 1;
  const int64_t nblk = (T + NB - 1) / NB;
#ifdef _OPENMP
#pragma omp parallel
#endif
  {
    double *Xt = (double *)aligned_alloc(64, sizeof(double) * p.D4 * NB);
    double *Yt = (double *)aligned_alloc(64, sizeof(double) * p.D4 * NB);
    double *lw = (double *)aligned_alloc(64, sizeof(double) * p.M * NB);
#ifdef _OPENMP
#pragma omp single
    nthreads = omp_get_num_threads();
#pragma omp for schedule(dynamic, 16)
#endif
    for (int64_t k = 0; k < nblk; ++k) {
      const int64_t t0 = k * NB;
      const int nf = (T - t0 < NB) ? (int)(T - t0) : NB;
      convert_block(&p, X + (size_t)g->D * t0, nf, Y + (size_t)g->D * t0, Xt, Yt, lw);
    }
    free(Xt); free(Yt); free(lw);
  }
  plan_free(&p);
  return nthreads;
}
