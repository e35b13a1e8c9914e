/*
 * vc_oracle.h -- CPU oracle for the VoiceConversion.jl hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C, single-threaded, FP64 restatement of the
 * reference's algorithm (r9y9/VoiceConversion.jl, Julia 0.5).  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the product library (libvcmi.so) never links,
 * loads or calls anything in oracle/.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - DTW (fit!/backward): pinned by the reference's own known-answer tests test/dtw.jl:7-31.
 *   - constructW: pinned by the structure test test/trajectory_gmmmap.jl:1-34.
 *   - fvconvert / predict_proba / trajectory solve / diag E-step: PARITY UNPINNED -- the reference's
 *     tests only check isfinite (test/vc.jl:26,50,72); the reference itself (Julia) cannot run in
 *     this image.  These functions are cross-checked against an independent numpy/scipy restatement
 *     (oracle/np_oracle.py) instead.
 *
 * All matrices use the Julia memory image: column-major Float64, a (D,T) feature matrix is
 * T frames of D contiguous doubles.  Indices returned to callers are 1-based like Julia's.
 */
#ifndef VC_ORACLE_H
#define VC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vco_gmmmap vco_gmmmap;

/* GMMMap(weights, mu, Sigma; swap) -- src/gmmmap.jl:62-90 (+ GMMMapParam :23-38, split_joint_gmm :41-52,
 * GaussianMixtureModel src/gmm.jl:8-20).  Returns NULL if a Sigma^xx block is not positive definite. */
vco_gmmmap *vco_gmmmap_new(const double *w, const double *mu, const double *sigma, int Dj, int M, int swap);
void vco_gmmmap_free(vco_gmmmap *g);
int vco_gmmmap_dim(const vco_gmmmap *g);          /* src/gmmmap.jl:94 */
int vco_gmmmap_ncomponents(const vco_gmmmap *g);  /* src/gmmmap.jl:95 */
/* copy of the precomputed Sigma^yx Sigma^xx^-1 tensor (D,D,M), src/gmmmap.jl:33-36 */
void vco_gmmmap_get_A(const vco_gmmmap *g, double *A);

/* fvconvert(g::GMMMap, x) -- src/gmmmap.jl:101-118. post (M) may be NULL. */
void vco_fvconvert(const vco_gmmmap *g, const double *x, double *y, double *post);
/* batch: X (D,T) -> Y (D,T); the loop of src/common.jl:17-19 without the power row */
void vco_fvconvert_batch(const vco_gmmmap *g, const double *X, int64_t T, double *Y);
/* same, frames spread over all host cores with OpenMP (strong CPU baseline); returns the thread count */
int vco_fvconvert_batch_mt(const vco_gmmmap *g, const double *X, int64_t T, double *Y);
/* predict_proba(gmm, X) -- src/gmm.jl:24-41 ; P is (M,T) */
void vco_predict_proba(const vco_gmmmap *g, const double *X, int64_t T, double *P);
/* lpr of src/gmm.jl:25-27 (log w_m + logpdf_m(x), -inf for a zero weight) before the softmax; L is (M,T) */
void vco_logdens(const vco_gmmmap *g, const double *X, int64_t T, double *L);
/* SURVEY 8d(ii), the strong CPU baseline as specified: the same math restructured as blocked FP64 GEMMs over frame blocks
 * (whitening with inv(L_m), regression with A_m, register-blocked micro-kernel, FMA) with OpenMP over the blocks;
 * vc_oracle_gemm.c.  Returns the thread count.  Not a parity reference: it is checked AGAINST vco_fvconvert_batch. */
int vco_fvconvert_batch_gemm(const vco_gmmmap *g, const double *X, int64_t T, double *Y);
/* predict(gmm, X) -- src/gmm.jl:44-58 ; 1-based argmax, first maximum wins */
void vco_predict(const vco_gmmmap *g, const double *X, int64_t T, int64_t *idx);
/* vc(c::FrameByFrameConverter, fm) -- src/common.jl:7-26 ; fm and out are (D+1,T) */
void vco_vc_frames(const vco_gmmmap *g, const double *fm, int64_t T, double *out);

/* DTW -- src/dtw.jl:93-145.  tmpl (D,S), seq (D,T). cost (S,T+1) f64 and bp (S,T+1) int64 may be NULL.
 * path (T) 1-based template index per sequence frame. */
void vco_dtw_fit(const double *tmpl, int64_t S, const double *seq, int64_t T, int D, int fstep, int bstep,
                 double *cost, int64_t *bp, int64_t *path);
/* align(src, tgt) -- src/align.jl:8-35.  src (D,S), tgt (D,T) -> newtgt (D,S); path (T) optional */
void vco_align(const double *src, int64_t S, const double *tgt, int64_t T, int D, double *newtgt, int64_t *path);

/* constructW(D,T) -- src/trajectory_gmmmap.jl:39-61, as COO triplets (1-based rows/cols).
 * Returns nnz; pass NULLs to query the count. */
int64_t vco_constructW(int D, int64_t T, int64_t *rows, int64_t *cols, double *vals);
/* push_delta(src) -- src/datasets.jl:6-13 ; src (D,T) -> out (2D,T) */
void vco_push_delta(const double *src, int D, int64_t T, double *out);

typedef struct vco_traj vco_traj;
/* TrajectoryGMMMap(g, T) -- src/trajectory_gmmmap.jl:11-31 (g's dim is 2D: static+delta). */
vco_traj *vco_traj_new(const vco_gmmmap *g);
void vco_traj_free(vco_traj *t);
/* fvconvert(tgmm, X) -- src/trajectory_gmmmap.jl:65-110.  X (2D,T) -> Y (D,T).
 * Optional outputs: mhat (T, 1-based), Ey (2D,T). Returns 0, or nonzero if the normal matrix is not PD. */
int vco_traj_fvconvert(const vco_traj *t, const double *X, int64_t T, double *Y, int64_t *mhat, double *Ey);
/* vc(c::TrajectoryConverter, fm) -- src/common.jl:31-63.  fm (2D+1,T) -> out (D+1,T), chunks of L frames */
int vco_vc_traj(const vco_traj *t, const double *fm, int64_t T, int64_t L, double *out);

/* fvpostf(vs::VarianceScaling, src) -- src/gv.jl:10-15; src, out (D,T) */
void vco_variance_scaling(const double *src, int D, int64_t T, const double *sigma2, double *out);
/* fvconvert(tgv::TrajectoryGVGMMMap, X; epochs, alpha) -- src/trajectory_gmmmap.jl:139-189.  muv (D), Sigvv (D,D).
 * Returns nonzero if the trajectory solve fails or Sigvv is singular.  PARITY UNPINNED by the reference's tests. */
int vco_trajgv_fvconvert(const vco_traj *t, const double *X, int64_t T, const double *muv, const double *Sigvv,
                         int epochs, double alpha, double *Y);
/* diffgmm(params) -- src/diffgmm.jl:9-25, on joint parameters mu (2D,M), sigma (2D,2D,M) */
void vco_diffgmm(const double *mu, const double *sigma, int Dj, int M, double *mu_out, double *sigma_out);

/* mc2e(mc, alpha, len) (MelGeneralizedCepstrums, third party; call site src/align.jl:48).  mc (D,T) -> e (T) */
void vco_mc2e(const double *mc, int D, int64_t T, double alpha, int fftlen, double *e);
/* align_mcep -- src/align.jl:38-55; returns the number of kept columns; outputs hold up to S columns */
int64_t vco_align_mcep(const double *src, int64_t S, const double *tgt, int64_t T, int D, double alpha, int fftlen,
                       double threshold, int remove_silence, double *src_out, double *newtgt_out);
/* one utterance of ParallelDataset(joint=true) -- src/datasets.jl:60-84 */
void vco_joint_features(const double *src, const double *tgt, int D, int64_t n, int ignore0th, int add_delta, int diff,
                        double *out);

/* Diagonal-covariance E-step (call site bin/train_gmm.jl:103 -> sklearn.mixture).
 * X (Dj,N); w (M); mu, var (Dj,M).  Outputs S0 (M), S1,S2 (Dj,M), loglik = sum_n lse_n. */
void vco_estep_diag(const double *X, int64_t N, int Dj, int M, const double *w, const double *mu, const double *var,
                    double *S0, double *S1, double *S2, double *loglik);

/* Full-covariance E-step (what bin/train_gmm.jl:84-103 runs through sklearn.mixture.GMM(covariance_type="full")).
 * X (Dj,N); w (M); mu (Dj,M); sigma (Dj,Dj,M).  S0 (M), S1 (Dj,M), S2 (Dj,Dj,M) = sum gamma x x', loglik.
 * Returns nonzero if a covariance is not positive definite.  PARITY UNPINNED by the reference's tests. */
int vco_estep_full(const double *X, int64_t N, int Dj, int M, const double *w, const double *mu, const double *sigma,
                   double *S0, double *S1, double *S2, double *loglik);

#ifdef __cplusplus
}
#endif
#endif
