// gmmmap_handle.hpp -- the opaque vcmi_gmmmap handle (shared by gmmmap.hip and traj.hip).
#pragma once
#include "vcmi_common.hpp"

struct vcmi_gmmmap {
  int D = 0;    // dim(g): source feature dimension, src/gmmmap.jl:94
  int DP = 0;   // D rounded up to a multiple of 4 (MFMA k-step)
  int M = 0;    // ncomponents(g)
  int device = 0;
  int kernel_choice = 0;   // 0 auto, 1 generic VALU, 2 MFMA
  // fvconvert skips the regression of mixture m for a 16-frame tile when l_m < max_l - prune (nats) for all its frames:
  // the posterior there is below e^-prune (1e-20 at 46: under the rounding error of the sum).  +inf: dense loop.
  double prune = 46.0;
  vcmi::DevBuf<unsigned long long> prune_count;   // optional diagnostic counters (vcmi_gmmmap_prune_stats): [0] (tile, mixture) regressions, [1] FP64 MFMAs issued, [2] BF16 MFMAs of the screen
  // Mean fraction of the mixtures that lie within e^-46 of the best one for a frame drawn from the model itself (256 frames
  // sampled on the host by prepare(), fixed seed).  Reported by vcmi_gmmmap_convert_plan (synthetic SURVEY 8d models: 1/M;
  // the reference's trained 32-mixture model: 0.37).
  double model_active_frac = 0.0;
  // ... and the mean fraction of the mixtures that the LAST 16-row whitening tile's share of |z|^2 alone does not put
  // e^-46 under the best one: what the "peaked" loop's first test leaves undecided.  Selects the loop shape (convert_shape).
  double model_undecided_frac = 0.0;
  // ... and the fractions that the 4 / 2 / 1 strongest screening rows (largest eigenpairs of inv(Sxx_m)) leave undecided: what the screen of shape 3
  // (gmmmap_screen.hpp) lets through with that many rows per mixture.  prepare() picks screen_rpm (the cheapest) and keeps its
  // fraction in model_undecided4_frac: small (<= kScreenModelFrac) -> grouped calls run the screening kernel.
  double model_undecided_rows[3] = {1.0, 1.0, 1.0};
  double model_undecided4_frac = 1.0;
  int screen_rpm = 4;
  // mean fraction of the mixtures whose four-row bound reaches a model-drawn frame's best log-density: what predict's screen
  // (gmmmap_screen_argmax_kernel) would have to evaluate in full besides the frame's own mixture
  double model_argmax_survivors_frac = 1.0;

  // host copies kept for accessors and for TrajectoryGMMMap's constructor (row-major (D,D) per mixture)
  std::vector<double> h_A_julia;   // Julia memory image (D,D,M) of ΣʸˣΣˣˣ⁻¹
  std::vector<double> h_A, h_Sxy, h_Syy, h_mux, h_muy;

  // device parameters, generic layout: [M][DP][DP] row-major / [M][DP] / [M]
  vcmi::DevBuf<double> U, A, cz, b, lc;
  vcmi::DevBuf<double> At;        // A transposed, [M][DP (k)][DP (row)]: only for dimensions without a tile-kernel instantiation
  // device parameters, MFMA fragment order: [M][Tiling::BLK]
  vcmi::DevBuf<double> packed;    // [U_m ; A_m] tiles (convert)
  vcmi::DevBuf<double> packedU;   // U_m tiles only (log-density / posterior / argmax)
  vcmi::DevBuf<double> packedQ;   // stages of the screen: 4 tiles x (screen_rpm rows of 16 / screen_rpm mixtures) per stage (convert, shape 3)
  vcmi::DevBuf<double> packedQ16;  // the four-row screen split into bf16 hi + lo (screen on the BF16 matrix pipe; DP <= 40)
  vcmi::DevBuf<double> packedQA;  // stages of predict's screen: four rows per mixture, every tile-kernel dimension (gmmmap_screen_argmax_kernel)
  vcmi::DevBuf<double> packedU2;  // U_m tiles only, tile by tile, last tile first (predict with early exit; host-prepared handles)
  // fvconvert's frame grouping (gmmmap_group_key_kernel): nearest-source-mean operand [-2 mu | |mu|^2] in MFMA fragment order,
  // and the call's scratch: key (T), perm (T), counts (M), cursors (M)
  vcmi::DevBuf<double> gfrag;
  vcmi::DevBuf<double> gfrag16;   // the same operand split into bf16 hi + lo for gmmmap_group_key16_kernel
  vcmi::DevBuf<int> grp;
  vcmi::StreamOrder grp_order;

  // issue-order table of the U-only tiling (slot -> tile << 16 | k-step) for the on-device packer (gmm_px_prepare_device)
  vcmi::DevBuf<int> px_table;
  int px_table_dp = 0;

  // grow-only device scratch (two-pass predict of the generic path)
  vcmi::DevBuf<double> scratch_lp;

  // constructor arguments, kept so that the converter can be re-created on the other devices of a device group
  // (vcmi_set_devices): replicas[i] lives on member i's device and is made lazily by that member's worker thread
  std::vector<double> in_w, in_mu, in_sigma;
  int in_Dj = 0, in_swap = 0;
  std::vector<vcmi_gmmmap *> replicas;
  uint64_t replicas_epoch = 0;
  ~vcmi_gmmmap() {
    for (vcmi_gmmmap *r : replicas) delete r;
  }
};

namespace vcmi {
bool gmmmap_has_mfma(int DP);
int gmm_px_create(const double *w, const double *mu, const double *sigma, int D, int M, vcmi_gmmmap **out);
// Same handle prepared ON THE DEVICE from device-resident parameters (w (M), mu (D,M), sigma (D,D,M)): one workgroup
// per mixture does the Cholesky, the triangular inverse and the MFMA operand packing.  Asynchronous on `st`;
// *d_flag (device int, zeroed by the caller) receives m+1 for a mixture whose covariance is not positive definite.
bool gmm_px_device_prepare_supported(int D);
int gmm_px_prepare_device(vcmi_gmmmap **inout, const double *d_w, const double *d_mu, const double *d_sigma, int D, int M,
                          int *d_flag, hipStream_t st);
int gmmmap_convert_device(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dY, int64_t ldy, hipStream_t st);
int gmmmap_logdens_device(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dLP, hipStream_t st);
int gmmmap_posterior_device(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, double *dP, hipStream_t st);
// allow_screen: long inputs may be grouped and run the screened arg-max (exact; pays for draws from a peaked p(x))
int gmmmap_predict_device(vcmi_gmmmap *g, const double *dX, int64_t ldx, int64_t T, int64_t *didx, hipStream_t st, bool allow_screen = true);
// Device group support (devgroup.hpp): gmmmap_sync_replicas is called by the host thread before group_run; inside the
// run member i obtains its converter (g itself on g's device, else a replica created on first use).
void gmmmap_sync_replicas(vcmi_gmmmap *g);
int gmmmap_member(vcmi_gmmmap *g, int member, vcmi_gmmmap **out);
}  // namespace vcmi
