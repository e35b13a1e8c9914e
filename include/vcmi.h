/*
 * vcmi.h -- C-ABI of libvcmi.so: the MI355X (gfx950) hot path of r9y9/VoiceConversion.jl.
 *
 * The reference is pure Julia and has no FFI of its own; its boundary is the exported Julia API
 * (src/VoiceConversion.jl:12-38, src/dtw.jl:7).  Each entry point below replaces the body of one of
 * those Julia functions and is what the Julia wrapper (voiceconversion.jl_amd/julia/VoiceConversionMI.jl)
 * binds with `ccall((:sym, "libvcmi"), Cint, ...)`; the same symbols are bound from Python ctypes
 * (voiceconversion.jl_amd/_lib.py).  INTEGRATION.md shows both bindings.
 *
 * Conventions
 *   - every array is the Julia memory image: column-major Float64; a (D,T) feature matrix is T frames
 *     of D contiguous doubles (`ld` = distance in doubles between consecutive frames, >= D);
 *   - indices written to callers are Julia's: Int64, 1-based;
 *   - plain "host" entry points take host pointers, copy to the current HIP device, run, copy back and
 *     return when the result is in the caller's buffer; `_dev` entry points take DEVICE pointers and a
 *     hipStream_t (passed as void*; NULL = default stream) and only enqueue work;
 *   - every function returns a vcmi_status; vcmi_last_error() gives the thread-local message.  No C++
 *     exception or abort crosses this boundary.  The Julia wrapper maps VCMI_ERR_DIM ->
 *     DimensionMismatch (src/gmmmap.jl:102, src/trajectory_gmmmap.jl:68, src/align.jl:11-13),
 *     VCMI_ERR_NOT_PD -> PosDefException (raised by MvNormal in src/gmm.jl:17), others -> ErrorException;
 *   - handles are not re-entrant: one handle must not be used from two threads at once (the reference's
 *     GMMMap carries mutable scratch too, src/gmmmap.jl:59).
 */
#ifndef VCMI_H
#define VCMI_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  VCMI_OK = 0,
  VCMI_ERR_DIM = 1,       /* inconsistent dimensions              -> DimensionMismatch */
  VCMI_ERR_NOT_PD = 2,    /* covariance block not positive definite -> PosDefException */
  VCMI_ERR_HIP = 3,       /* HIP runtime / kernel failure */
  VCMI_ERR_OOM = 4,       /* host or device allocation failed */
  VCMI_ERR_ARG = 5,       /* NULL pointer, negative size, unsupported option */
  VCMI_ERR_NO_DEVICE = 6  /* no gfx950 device visible */
} vcmi_status;

const char *vcmi_last_error(void);
const char *vcmi_version(void);
int vcmi_device_count(int *count);
/* Select the HIP device used by this thread's subsequent calls (one process per GPU: LOCAL_RANK). */
int vcmi_set_device(int device);
/* ONE host process driving several GPUs -- what a Julia host is (the reference is a single process, src/common.jl:7-63,
 * bin/train_gmm.jl:103).  vcmi_set_devices(devs, n) makes a device group: one persistent worker thread per listed
 * device.  From then on the HOST-POINTER entry points shard their work over the group and join before returning:
 *   vcmi_gmmmap_convert / vcmi_vc_frames / _posterior / _predict   contiguous frame blocks, no collective;
 *   vcmi_dtw_fit_batch / vcmi_align_batch                          pairs split by cost S*T, no collective;
 *   vcmi_traj_convert_batch / vcmi_vc_traj / vcmi_trajgv_convert_batch   utterances (chunks) split by length, no collective;
 *   vcmi_estep_diag / vcmi_estep_full                              frame blocks, then ONE ncclAllReduce(sum) of the packed
 *                                                                  statistics over RCCL (librccl is loaded on first use).
 * Converters are re-created on the other devices on first use.  The `_dev` entry points are unaffected (they run on the
 * device that owns their pointers).  n = 0 removes the group.  The list may name a device twice (both workers share it;
 * useful to test the sharding on a 1-GPU box) except for the E-step, whose RCCL communicator needs distinct devices. */
int vcmi_set_devices(const int *devices, int n);
int vcmi_get_devices(int *devices, int capacity, int *n);

/* Caller-pinned arrays.  A Julia process keeps its feature matrices across calls (src/common.jl:7-26: `vc(c, fm)` reads fm
 * and returns `converted`; bin/vc.jl:82 calls it once per file on arrays it owns).  vcmi_host_register(ptr, bytes) page-locks
 * [ptr, ptr + bytes) ONCE (hipHostRegister); from then on every host-pointer entry point whose dense input and / or output
 * lies inside a registered range moves it by DMA straight between that array and HBM -- no staging slot, no host memcpy --
 * (vcmi_gmmmap_convert, vcmi_vc_frames' input, vcmi_gmmmap_posterior / _predict: each side independently; a side that is
 * not registered, or strided, keeps the staged path); the plain uploads / downloads of the other entry points (the frame
 * matrix of vcmi_estep_diag / _full, result buffers) go straight from / to a registered array as well.  Arrays pinned by
 * the caller's own runtime (hipHostMalloc /
 * hipHostRegister) are recognised as well.  Results are identical either way.  vcmi_host_unregister(ptr) takes the pointer
 * that was registered; an array must be unregistered before it is freed.  Ranges must not overlap (VCMI_ERR_ARG).
 * What is locked are the WHOLE PAGES inside the array -- its partial first and last page, which it shares with its neighbours on
 * the heap, travel as ordinary copies (unmapping such a shared page at unregistration broke later transfers of the HIP runtime
 * from pageable memory nearby: DESIGN 4b, round 6) -- and an array with less than 1 MB of whole pages is recorded, not locked:
 * its calls stage as before.
 * vcmi_host_is_registered: *flag = 1 when the whole range lies in an array registered here (locked or recorded) or in memory
 * another runtime pinned. */
int vcmi_host_register(void *ptr, size_t bytes);
int vcmi_host_unregister(void *ptr);
int vcmi_host_is_registered(const void *ptr, size_t bytes, int *flag);

/* ---------------------------------------------------------------------------------------------
 * GMMMap -- src/gmmmap.jl:57-118, posterior helpers src/gmm.jl:24-58
 * ------------------------------------------------------------------------------------------- */
typedef struct vcmi_gmmmap vcmi_gmmmap;

/* GMMMap(weights, mu, Sigma; swap=false), src/gmmmap.jl:62-90: weights (M), mu (Dj,M), Sigma (Dj,Dj,M).
 * Splits the joint GMM (:41-52), forms A_m = Sigma^yx_m inv(Sigma^xx_m) (:33-36) and the Cholesky factor
 * of Hermitian(Sigma^xx_m) (src/gmm.jl:16-17) on the host, and uploads the packed per-mixture blocks. */
int vcmi_gmmmap_create(const double *weights, const double *mu, const double *sigma, int Dj, int M, int swap,
                       vcmi_gmmmap **out);
int vcmi_gmmmap_destroy(vcmi_gmmmap *g);
int vcmi_gmmmap_dim(const vcmi_gmmmap *g);          /* dim(g),         src/gmmmap.jl:94 */
int vcmi_gmmmap_ncomponents(const vcmi_gmmmap *g);  /* ncomponents(g), src/gmmmap.jl:95 */
/* g.params.ΣʸˣΣˣˣ⁻¹ (D,D,M), src/gmmmap.jl:21 */
int vcmi_gmmmap_get_A(const vcmi_gmmmap *g, double *A);

/* fvconvert(g, x) for every column of X (D,T) -> Y (D,T); src/gmmmap.jl:101-118.  One launch replaces the
 * frame loop of src/common.jl:17-19.  T = 1 is the reference's per-frame call. */
int vcmi_gmmmap_convert(vcmi_gmmmap *g, const double *X, int64_t ldx, int64_t T, double *Y, int64_t ldy);
int vcmi_gmmmap_convert_dev(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dY, int64_t ldy,
                            void *stream);
/* vc(c::FrameByFrameConverter, fm), src/common.jl:7-26: fm, out are (D+1,T); row 1 (power) is copied. */
int vcmi_vc_frames(vcmi_gmmmap *g, const double *fm, int64_t T, double *out);
/* ... with the VarianceScaling post-filter (src/gv.jl:10-15) as a fused post step: out[2:end,:] = fvpostf(VarianceScaling(sigma2),
 * vc(g, fm)[2:end,:]); sigma2 (D) host vector, NULL = plain vcmi_vc_frames.  The converted matrix never leaves HBM between the
 * conversion and the filter: one upload, one download. */
int vcmi_vc_frames_postf(vcmi_gmmmap *g, const double *fm, int64_t T, const double *sigma2, double *out);
/* predict_proba(g.px, X) -> P (M,T), src/gmm.jl:24-41 */
int vcmi_gmmmap_posterior(vcmi_gmmmap *g, const double *X, int64_t ldx, int64_t T, double *P);
int vcmi_gmmmap_posterior_dev(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dP, void *stream);
/* predict(g.px, X) -> idx (T), 1-based, first maximum wins; src/gmm.jl:44-58.  The index is exact on every path: a
 * mixture's evaluation is cut short (or, on inputs of 8192 frames or more of a peaked model, skipped after a four-row
 * screen on grouped frames) only when an upper bound of its log-density lies below a log-density that was evaluated. */
int vcmi_gmmmap_predict(vcmi_gmmmap *g, const double *X, int64_t ldx, int64_t T, int64_t *idx);
int vcmi_gmmmap_predict_dev(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, int64_t *didx, void *stream);
/* Kernel selection: 0 = auto (MFMA tile kernel when dim(g), rounded up to a multiple of 4, is one of 16..80 in steps of
 * 4; other dimensions up to 160: the tiled MFMA log-density kernel, followed for fvconvert by a softmax / regression
 * kernel over the mixtures that matter; else the generic VALU kernel), 1 = force the generic VALU kernel, 2 = force the
 * MFMA tile kernel (VCMI_ERR_ARG if unavailable). */
int vcmi_gmmmap_set_kernel(vcmi_gmmmap *g, int which);
/* Posterior pruning of fvconvert (src/gmmmap.jl:109-117 sums over ALL mixtures; this changes which terms are evaluated,
 * not the result): the regression A_m x + b_m of mixture m is skipped for a tile of 16 consecutive frames when
 * l_m < max_k l_k - nats for every frame of the tile, i.e. when its posterior there is below e^-nats.  Default 46.0
 * (1e-20: below the rounding error of the remaining terms; y changes by < 1e-18 relative).  +infinity (or any value
 * >= 1e300) evaluates every mixture for every frame -- the dense loop the flop count of SURVEY 8(d) assumes.
 * VCMI_ERR_ARG for nats < 40 (would be visible in y at double precision) or NaN. */
int vcmi_gmmmap_set_prune(vcmi_gmmmap *g, double nats);
/* Diagnostic counter of the pruning: *evaluated (may be NULL) receives the number of (16-frame tile, mixture) regressions
 * the MFMA fvconvert kernel has evaluated on this handle's device since the counter was last enabled (the dense count is
 * ceil(T / 16) * M per converted matrix); then enable != 0 (re)starts the counter at zero, enable == 0 switches it off
 * (the default: the kernel then updates nothing).  Synchronises with the device. */
int vcmi_gmmmap_prune_stats(vcmi_gmmmap *g, int enable, int64_t *evaluated);
/* What fvconvert does with this handle, for measurement (bench.py prices the matrix pipe with it):
 *   *mfma_issued (may be NULL)  v_mfma_f64_16x16x4 instructions (2048 flop each) the MFMA fvconvert kernel has issued since
 *                               vcmi_gmmmap_prune_stats last enabled the counters (0 while they are off); synchronises;
 *   *shape (may be NULL)        the loop the library chose for this model: 0 dense (prune = +inf), 1 "broad" (every
 *                               whitening tile, one branch around the regression), 2 "peaked" (a wrong mixture is decided
 *                               out on its last whitening tile), 3 "screened" (calls of 8192 frames or more, D <= 48: the
 *                               frames are grouped, the group's own mixture is evaluated, every other one is ruled out -- four
 *                               at a time -- by a lower bound of its distance from the four largest eigenpairs of inv(Sxx),
 *                               survivors are evaluated in full; smaller calls of such a model run shape 2),
 *                               -1 no MFMA tile kernel for this dimension;
 *   *model_active_frac, *model_undecided_frac (may be NULL)  the model properties behind that choice, estimated once at
 *                               creation on 256 frames drawn from the model itself: the mean fraction of the mixtures within
 *                               e^-46 of the best one, and the fraction that the last 16-row whitening tile's share of |z|^2
 *                               alone does NOT put e^-46 under the best one ("peaked" is chosen below 0.35; "screened" when
 *                               the screen's own rows leave at most 0.05 undecided).  The shape selects code, never a result:
 *                               every shape gives the dense loop's y to rounding. */
int vcmi_gmmmap_convert_plan(vcmi_gmmmap *g, int64_t *mfma_issued, int *shape, double *model_active_frac,
                             double *model_undecided_frac);

/* ---------------------------------------------------------------------------------------------
 * DTW -- src/dtw.jl:93-145 (fit! + backward), align -- src/align.jl:8-35
 * ------------------------------------------------------------------------------------------- */
/* fit!(DTW(fstep,bstep), template (D,S), sequence (D,T)) -> path (T).  costtable (S,T+1) f64 and
 * backpointer (S,T+1) Int64 are the fields d.costtable / d.backpointer (src/dtw.jl:15-16); pass NULL
 * for both to skip materialising them ("path-only" mode). */
int vcmi_dtw_fit(const double *tmpl, int64_t S, const double *seq, int64_t T, int D, int fstep, int bstep,
                 int64_t *path, double *costtable, int64_t *backpointer);
/* n independent pairs in one launch (one workgroup per pair).  Host pointers. */
int vcmi_dtw_fit_batch(int64_t n, const double *const *tmpl, const int64_t *S, const double *const *seq,
                       const int64_t *T, int D, int fstep, int bstep, int64_t *const *path);
/* Device-resident batch: `feats` is one device buffer holding every (D,len) matrix; pair p has its template at
 * feats + tmpl_off[p] (S[p] frames) and its sequence at feats + seq_off[p] (T[p] frames); its path goes to
 * paths + path_off[p].  Offsets (in elements) and lengths are HOST arrays of n entries. */
/* Stream semantics: the call only enqueues work on `stream`.  The pair descriptors and the per-thread workspaces are
 * shared between calls: every call first makes `stream` wait for the previous call's last kernel (whatever stream that
 * ran on), and the descriptors are copied asynchronously on `stream` from pinned slots -- consecutive calls on any
 * streams, including non-blocking ones, are ordered correctly without a host synchronisation. */
int vcmi_dtw_fit_batch_dev(int64_t n, const double *feats, const int64_t *tmpl_off, const int64_t *S,
                           const int64_t *seq_off, const int64_t *T, int D, int fstep, int bstep, int64_t *paths,
                           const int64_t *path_off, void *stream);
/* align(src (D,S), tgt (D,T)) -> newtgt (D,S); path (T) optional (NULL).  src/align.jl:8-35 */
int vcmi_align(const double *src, int64_t S, const double *tgt, int64_t T, int D, double *newtgt, int64_t *path);
int vcmi_align_batch(int64_t n, const double *const *src, const int64_t *S, const double *const *tgt,
                     const int64_t *T, int D, double *const *newtgt);

/* ---------------------------------------------------------------------------------------------
 * Diagonal-covariance E-step (call site bin/train_gmm.jl:103; math SURVEY A.6)
 * ------------------------------------------------------------------------------------------- */
/* X (Dj,N); w (M); mu, var (Dj,M) -> S0 (M), S1, S2 (Dj,M), loglik (1). */
int vcmi_estep_diag(const double *X, int64_t N, int Dj, int M, const double *w, const double *mu, const double *var,
                    double *S0, double *S1, double *S2, double *loglik);
/* Device X; parameters are HOST arrays (tiny).  dstats is a DEVICE buffer of vcmi_estep_stats_len(Dj,M)
 * doubles laid out [S0 (M) | S1 (Dj,M) | S2 (Dj,M) | loglik] -- one contiguous buffer so that the
 * multi-GPU path is a single all-reduce(sum).  Every sum has a fixed order and nothing depends on earlier calls: identical inputs
 * give identical bits, run to run, thread to thread.  From 65536 frames on (M <= 128, Dj <= 80) there are two paths: frames that
 * ONE mixture owns (every other responsibility exactly 0 in double precision) can be settled by a certified low-precision screen
 * and summed per mixture in one pass over X, the rest goes through the FP64 kernel (csrc/estep_hard.hpp, estep_path.hpp;
 * statistics within 1e-12 of each other).  VCMI_ESTEP_AUTO (default) decides per call, on the device, from a sample of the call's
 * own frames (16 chunks of 1024: at most a quarter without an owner -> the hard-assignment path); vcmi_estep_set_path pins
 * VCMI_ESTEP_HARD or VCMI_ESTEP_SOFT for the calling thread (training loops that must take the same path on every rank).
 * Log-densities of competing mixtures are evaluated term by term, (x - mu)^2 / var, wherever the expanded form's
 * rounding-error bound exceeds 1e-10 (variances near min_covar). */
enum { VCMI_ESTEP_AUTO = 0, VCMI_ESTEP_HARD = 1, VCMI_ESTEP_SOFT = 2 };
int vcmi_estep_set_path(int path);
int vcmi_estep_get_path(int *path);
int64_t vcmi_estep_stats_len(int Dj, int M);
int vcmi_estep_diag_dev(const double *dX, int64_t N, int Dj, int M, const double *w, const double *mu,
                        const double *var, double *dstats, void *stream);

/* Full-covariance E-step -- what the reference's training actually runs: bin/train_gmm.jl:84-89 builds
 * sklearn.mixture.GMM(covariance_type="full") and :103 calls fit.  w (M); mu (Dj,M); sigma (Dj,Dj,M).
 * Statistics S0 (M), S1 (Dj,M), S2 (Dj,Dj,M) = sum_n gamma_nm x_n x_n', loglik; device layout
 * [S0 | S1 | S2 | loglik] of vcmi_estep_full_stats_len(Dj,M) doubles (one all-reduce).  VCMI_ERR_NOT_PD when a
 * covariance is not positive definite. */
int64_t vcmi_estep_full_stats_len(int Dj, int M);
int vcmi_estep_full(const double *X, int64_t N, int Dj, int M, const double *w, const double *mu, const double *sigma,
                    double *S0, double *S1, double *S2, double *loglik);
int vcmi_estep_full_dev(const double *dX, int64_t N, int Dj, int M, const double *w, const double *mu,
                        const double *sigma, double *dstats, void *stream);

/* EM state resident on the device -- the loop inside `gmm[:fit](dataset.X')`, bin/train_gmm.jl:84-103
 * (sklearn.mixture.GMM(covariance_type="full", min_covar)).  Parameters and whitening blocks stay in HBM between
 * iterations; one iteration is
 *   vcmi_gmm_em_estep_dev (local statistics of this GPU's frames, asynchronous on `stream`)
 *   -> the caller sums the statistics buffers of all ranks (one RCCL all-reduce; nothing to do on one GPU)
 *   -> vcmi_gmm_em_mstep (w = S0/sum S0, mu = S1/S0, Sigma = S2/S0 - mu mu' + min_covar I, then the Cholesky
 *      whitening of every mixture on the device; returns the log-likelihood the statistics carry).
 * dstats: caller-owned DEVICE buffer of vcmi_estep_full_stats_len(Dj,M) doubles, layout as vcmi_estep_full_dev.
 * vcmi_gmm_em_mstep synchronises the stream and reports VCMI_ERR_NOT_PD if a covariance (initial or updated) is not
 * positive definite. */
typedef struct vcmi_gmm_em vcmi_gmm_em;
int vcmi_gmm_em_create(int Dj, int M, const double *w, const double *mu, const double *sigma, double min_covar,
                       vcmi_gmm_em **out);
int vcmi_gmm_em_destroy(vcmi_gmm_em *h);
int vcmi_gmm_em_estep_dev(vcmi_gmm_em *h, const double *dX, int64_t N, double *dstats, void *stream);
int vcmi_gmm_em_mstep(vcmi_gmm_em *h, const double *dstats, void *stream, double *loglik);
int vcmi_gmm_em_get(vcmi_gmm_em *h, double *w, double *mu, double *sigma);

/* ---------------------------------------------------------------------------------------------
 * TrajectoryGMMMap -- src/trajectory_gmmmap.jl:3-110, vc src/common.jl:31-63, push_delta src/datasets.jl:6-13
 * ------------------------------------------------------------------------------------------- */
typedef struct vcmi_traj vcmi_traj;
/* TrajectoryGMMMap(g, T): g's dim is 2D (static+delta); precomputes Dy_m = inv(Sigma^yy_m - A_m Sigma^xy_m)
 * (:24-28).  T only sets length(t) (:34); W is a stencil and is never materialised.  g must outlive t. */
int vcmi_traj_create(vcmi_gmmmap *g, int64_t T, vcmi_traj **out);
int vcmi_traj_destroy(vcmi_traj *t);
/* length(t).  As in the reference, fvconvert rebuilds W when the sequence length differs (src/trajectory_gmmmap.jl:70-72),
 * so after vcmi_traj_convert(t, X, T, Y) the length is T, and after vcmi_vc_traj it is the length of the last chunk --
 * which is the chunk length the NEXT vc call uses (src/common.jl:41).  The batch entry points leave it unchanged. */
int64_t vcmi_traj_length(const vcmi_traj *t);
/* fvconvert(t, X (2D,T)) -> Y (D,T); src/trajectory_gmmmap.jl:65-110 */
int vcmi_traj_convert(vcmi_traj *t, const double *X, int64_t T, double *Y);
/* n utterances, one launch; host pointers */
int vcmi_traj_convert_batch(vcmi_traj *t, int64_t n, const double *const *X, const int64_t *T, double *const *Y);
/* device-resident batch: utterance u has X at dX + x_off[u] ((2D,T[u])) and Y at dY + y_off[u] ((D,T[u])) */
int vcmi_traj_convert_batch_dev(vcmi_traj *t, int64_t n, const double *dX, const int64_t *x_off, const int64_t *T,
                                double *dY, const int64_t *y_off, void *stream);
/* vc(c::TrajectoryConverter, fm (2D+1,T)) -> out (D+1,T) in chunks of length(t) frames; src/common.jl:31-63 */
int vcmi_vc_traj(vcmi_traj *t, const double *fm, int64_t T, double *out);
/* push_delta(src (D,T)) -> out (2D,T); src/datasets.jl:6-13.  Host matrices, host arithmetic (O(DT), no device needed). */
int vcmi_push_delta(const double *src, int D, int64_t T, double *out);
/* ... and on DEVICE-RESIDENT matrices with leading dimensions (lds >= D, ldo >= 2D), asynchronous on `stream`: the input of
 * the trajectory conversion is built where the features already are (bin/vc.jl:75-78 builds it in front of vc). */
int vcmi_push_delta_dev(const double *dsrc, int64_t lds, int D, int64_t T, double *dout, int64_t ldo, void *stream);
/* vc(c::TrajectoryConverter, fm) with the VarianceScaling post-filter (src/gv.jl:10-15) applied to the converted rows
 * 2..D+1 BEFORE the download: out[2:end,:] = fvpostf(VarianceScaling(sigma2), vc(t, fm)[2:end,:]); sigma2 (D) host vector,
 * NULL = plain vcmi_vc_traj.  One upload, one download; everything in between stays in HBM. */
int vcmi_vc_traj_postf(vcmi_traj *t, const double *fm, int64_t T, const double *sigma2, double *out);

/* ---------------------------------------------------------------------------------------------
 * TrajectoryGVGMMMap -- src/trajectory_gmmmap.jl:114-189
 * ------------------------------------------------------------------------------------------- */
typedef struct vcmi_trajgv vcmi_trajgv;
/* TrajectoryGVGMMMap(tgmm, mu^v (D), Sigma^vv (D,D)) :118-129; keeps (does not own) the trajectory handle.
 * VCMI_ERR_ARG if a GV mean is negative (the @assert of :124), VCMI_ERR_NOT_PD if Sigma^vv is singular. */
int vcmi_trajgv_create(vcmi_traj *t, const double *muv, const double *sigmavv, vcmi_trajgv **out);
int vcmi_trajgv_destroy(vcmi_trajgv *h);
/* fvconvert(tgv, X; epochs=100, alpha=1.0e-5) :139-168: trajectory solve, eq.(58) rescaling, then `epochs` steps of
 * y += alpha (omega (W'D^-1E - W'D^-1W y) + gvgrad(y)), all on the device.  X (2D,T) -> Y (D,T). */
int vcmi_trajgv_convert(vcmi_trajgv *h, const double *X, int64_t T, int epochs, double alpha, double *Y);
int vcmi_trajgv_convert_batch(vcmi_trajgv *h, int64_t n, const double *const *X, const int64_t *T, int epochs, double alpha,
                              double *const *Y);
int vcmi_trajgv_convert_batch_dev(vcmi_trajgv *h, int64_t n, const double *dX, const int64_t *x_off, const int64_t *T,
                                  int epochs, double alpha, double *dY, const int64_t *y_off, void *stream);

/* fvpostf(vs::VarianceScaling, src) -- src/gv.jl:10-21.  src, out (D,T), sigma2 (D); out may alias src. */
int vcmi_variance_scaling(const double *src, int D, int64_t T, const double *sigma2, double *out);
/* fvpostf! on a DEVICE-RESIDENT matrix (leading dimensions lds, ldo >= D; dout may be dsrc: in place), asynchronous on
 * `stream`; sigma2 (D) is a host vector.  Deterministic: every sum has a fixed order. */
int vcmi_variance_scaling_dev(const double *dsrc, int64_t lds, int D, int64_t T, const double *sigma2, double *dout,
                              int64_t ldo, void *stream);
/* diffgmm(params) -- src/diffgmm.jl:9-25 on joint parameters mu (2D,M), sigma (2D,2D,M); host arithmetic. */
int vcmi_diffgmm(const double *mu, const double *sigma, int Dj, int M, double *mu_out, double *sigma_out);

/* ---------------------------------------------------------------------------------------------
 * align_mcep and the joint training matrix -- src/align.jl:38-55, src/datasets.jl:52-98
 * ------------------------------------------------------------------------------------------- */
/* mc2e(mc, alpha, len) of MelGeneralizedCepstrums (third party; call site src/align.jl:48) for every column of
 * mc (D,T): energy of c2ir(freqt(mc, len-1, -alpha), len).  e (T). */
int vcmi_mc2e(const double *mc, int D, int64_t T, double alpha, int fftlen, double *e);
/* align_mcep(src, tgt, alpha, fftlen; threshold=-14.0, remove_silence=true): align, then keep the columns whose
 * log(mc2e(src)) exceeds the threshold.  src (D,S), tgt (D,T); src_out, newtgt_out (D, up to S); *ncols kept. */
int vcmi_align_mcep(const double *src, int64_t S, const double *tgt, int64_t T, int D, double alpha, int fftlen,
                    double threshold, int remove_silence, double *src_out, double *newtgt_out, int64_t *ncols);
/* align_mcep of every pair followed by ParallelDataset(joint=true; diff, ignore0th, add_delta).X, left ON THE DEVICE
 * for the E-step: per pair drop row 1 (ignore0th), push_delta on the kept frames (add_delta), tgt - src (diff), vcat;
 * pairs concatenated in order.  do_align = 0 takes pairs that are already aligned (S == T).  dXY: device buffer of
 * Dj * capacity_frames doubles with capacity_frames >= sum(S), Dj = 2 (D - ignore0th)(1 + add_delta); *nframes
 * receives the number of columns written; counts (optional, host, n entries) the frames kept per pair. */
int vcmi_parallel_dataset_dev(int64_t n, const double *const *src, const int64_t *S, const double *const *tgt,
                              const int64_t *T, int D, int do_align, double alpha, int fftlen, double threshold,
                              int remove_silence, int ignore0th, int add_delta, int diff, double *dXY,
                              int64_t capacity_frames, int64_t *nframes, int64_t *counts);

/* GVDataset(path; ignore0th, add_delta, nmax) from in-memory feature matrices -- src/datasets.jl:134-183 (the file loop
 * is the caller's): per utterance tgt = fm[i] (D,T[i]) without row 1 (ignore0th), with push_delta (add_delta);
 * gv = var(tgt, 2) (corrected, Julia's default); an utterance whose variance has a NaN (T = 1) is skipped as the
 * reference does.  out: (Dout, n) column-major, Dout = (D - ignore0th)(1 + add_delta), the kept utterances' columns in
 * order; *nkept = how many.  Host-side helper (O(D sum T), shared out over the library's host threads). */
int vcmi_gv_dataset(int64_t n, const double *const *fm, const int64_t *T, int D, int ignore0th, int add_delta, double *out,
                    int64_t *nkept);

#ifdef __cplusplus
}
#endif
#endif /* VCMI_H */
