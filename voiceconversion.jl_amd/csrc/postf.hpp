// postf.hpp -- the pre / post steps of vc on DEVICE-RESIDENT matrices: push_delta (src/datasets.jl:6-13)
// and the VarianceScaling post-filter (src/gv.jl:10-15), shared by postf.hip (C entries) and traj.hip (vcmi_vc_traj_postf).
#pragma once
#include "vcmi_common.hpp"

namespace vcmi {
// out (2D,T) with leading dimension ldo  <-  [src; delta(src)], src (D,T) with leading dimension lds; asynchronous on st
int push_delta_device(const double *dsrc, int64_t lds, int D, int64_t T, double *dout, int64_t ldo, hipStream_t st);
// fvpostf!: out[d,t] = sqrt(sigma2[d] / var_d) (src[d,t] - mean_d) + mean_d, mean / corrected variance per row over the T
// frames; sigma2 is a HOST vector (D).  dout may be dsrc (in place).  Asynchronous on st; deterministic (fixed-order sums).
int variance_scaling_device(const double *dsrc, int64_t lds, int D, int64_t T, const double *sigma2_host, double *dout,
                            int64_t ldo, hipStream_t st);
// rows r0 .. r0 + nrows - 1 of a (ldi, T) matrix -> rows q0 .. of a (ldo, T) matrix (power row / feature rows of vc's matrices)
int copy_rows_device(const double *din, int64_t ldi, int r0, int nrows, int64_t T, double *dout, int64_t ldo, int q0, hipStream_t st);
}  // namespace vcmi
