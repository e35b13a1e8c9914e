/* abi_smoke.c -- the C-ABI of libvcmi.so driven from plain C (what a Julia ccall or any other FFI host sees):
 * no Python, no torch.  Checks the reference's DTW known-answer tests (test/dtw.jl:7-31) and a GMMMap round trip
 * (accessors of test/gmmmap.jl:9-14, single-mixture model => fvconvert is the affine map y = muy + A (x - mux)).
 * Build: gcc -std=c11 -I include tests/c/abi_smoke.c -L voiceconversion.jl_amd -lvcmi -Wl,-rpath,... -lm */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "vcmi.h"

#define CHECK(x) do { int s_ = (x); if (s_ != VCMI_OK) { printf("FAIL %s -> %d: %s\n", #x, s_, vcmi_last_error()); return 1; } } while (0)

int main(void) {
  int ndev = 0;
  CHECK(vcmi_device_count(&ndev));
  if (ndev < 1) { printf("no device\n"); return 2; }
  /* ---- DTW KAT 1: template 4 frames, sequence 5 frames, 3-dim (column-major (D,frames)) */
  const double v1[] = {1, 2, 3, 1, 2, 4, 1, 8, 5, 10, 3, 6};
  const double v2[] = {1, 2, 3, 1, 2, 4, 1, 2, 5, 1, 8, 5, 10, 3, 6};
  int64_t path[8];
  double cost[4 * 6];
  int64_t bp[4 * 6];
  CHECK(vcmi_dtw_fit(v1, 4, v2, 5, 3, /*fstep*/ 0, /*bstep*/ 1, path, cost, bp));
  const int64_t exp1[] = {1, 2, 2, 3, 4};
  for (int i = 0; i < 5; ++i) if (path[i] != exp1[i]) { printf("FAIL dtw kat1 at %d: %lld\n", i, (long long)path[i]); return 1; }
  for (int i = 0; i < 4; ++i) if (cost[i] != (double)(i + 1) || bp[i] != i + 1) { printf("FAIL lazy_init column\n"); return 1; }
  /* ---- DTW KAT 2: 1-dim */
  const double a1[] = {0, 1, 2, 3, 4, 5}, a2[] = {0, 0, 1, 2, 3, 4, 4, 5};
  const int64_t exp2[] = {1, 1, 2, 3, 4, 5, 5, 6};
  CHECK(vcmi_dtw_fit(a1, 6, a2, 8, 1, 0, 1, path, NULL, NULL));
  for (int i = 0; i < 8; ++i) if (path[i] != exp2[i]) { printf("FAIL dtw kat2 at %d\n", i); return 1; }
  /* ---- GMMMap: one mixture, D = 4 (joint 8): Sigma = [[2I, 0.5I],[0.5I, I]] => A = 0.25 I */
  enum { D = 4, DJ = 8, T = 37 };
  double w[1] = {1.0}, mu[DJ], sig[DJ * DJ] = {0};
  for (int d = 0; d < DJ; ++d) mu[d] = 0.1 * (d + 1);
  for (int d = 0; d < D; ++d) {
    sig[d + DJ * d] = 2.0;
    sig[(D + d) + DJ * (D + d)] = 1.0;
    sig[d + DJ * (D + d)] = 0.5;
    sig[(D + d) + DJ * d] = 0.5;
  }
  vcmi_gmmmap *g = NULL;
  CHECK(vcmi_gmmmap_create(w, mu, sig, DJ, 1, 0, &g));
  if (vcmi_gmmmap_dim(g) != D || vcmi_gmmmap_ncomponents(g) != 1) { printf("FAIL accessors\n"); return 1; }
  double X[D * T], Y[D * T];
  for (int i = 0; i < D * T; ++i) X[i] = sin(0.37 * i);
  CHECK(vcmi_gmmmap_convert(g, X, D, T, Y, D));
  double worst = 0;
  for (int t = 0; t < T; ++t)
    for (int d = 0; d < D; ++d) {
      const double want = mu[D + d] + 0.25 * (X[d + D * t] - mu[d]);
      const double e = fabs(Y[d + D * t] - want);
      if (e > worst) worst = e;
    }
  if (!(worst < 1e-12)) { printf("FAIL fvconvert affine check, err %g\n", worst); return 1; }
  /* dimension error is reported, not crashed (src/gmmmap.jl:102) */
  if (vcmi_gmmmap_convert(g, X, D - 1, T, Y, D) != VCMI_ERR_DIM) { printf("FAIL expected VCMI_ERR_DIM\n"); return 1; }
  CHECK(vcmi_gmmmap_destroy(g));

  /* ---- VarianceScaling (src/gv.jl): every row ends with the requested corrected variance and keeps its mean */
  {
    enum { VD = 3, VT = 50 };
    double src[VD * VT], out[VD * VT], s2[VD] = {0.5, 1.0, 2.0};
    for (int i = 0; i < VD * VT; ++i) src[i] = sin(0.7 * i) + 0.1 * (i % VD);
    CHECK(vcmi_variance_scaling(src, VD, VT, s2, out));
    for (int d = 0; d < VD; ++d) {
      double m0 = 0, m1 = 0, v1 = 0;
      for (int t = 0; t < VT; ++t) { m0 += src[d + VD * t]; m1 += out[d + VD * t]; }
      m0 /= VT; m1 /= VT;
      for (int t = 0; t < VT; ++t) v1 += (out[d + VD * t] - m1) * (out[d + VD * t] - m1);
      v1 /= (VT - 1);
      if (fabs(m1 - m0) > 1e-12 || fabs(v1 - s2[d]) > 1e-12) { printf("FAIL variance scaling row %d\n", d); return 1; }
    }
  }
  /* ---- mc2e: a cepstrum with only c0 has the impulse response [exp(c0), 0, ...] => energy exp(2 c0) */
  {
    double c[5] = {0.7, 0, 0, 0, 0}, e = 0;
    CHECK(vcmi_mc2e(c, 5, 1, 0.35, 64, &e));
    if (fabs(e - exp(1.4)) > 1e-12) { printf("FAIL mc2e %.17g\n", e); return 1; }
  }
  /* ---- full-covariance E-step + device-resident EM on a single Gaussian: one iteration from any start lands on the
   *      sample mean / (biased) sample covariance + min_covar */
  {
    enum { ED = 3, EN = 400 };
    double Xe[ED * EN], w1[1] = {1.0}, mu0[ED] = {0, 0, 0}, sg0[ED * ED] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int n = 0; n < EN; ++n) {
      Xe[0 + ED * n] = sin(0.3 * n) + 1.0;
      Xe[1 + ED * n] = cos(0.11 * n) * 2.0 - 0.5;
      Xe[2 + ED * n] = 0.5 * Xe[0 + ED * n] + sin(1.7 * n);
    }
    double S0[1], S1[ED], S2[ED * ED], ll;
    CHECK(vcmi_estep_full(Xe, EN, ED, 1, w1, mu0, sg0, S0, S1, S2, &ll));
    if (fabs(S0[0] - EN) > 1e-9) { printf("FAIL estep_full S0 %.17g\n", S0[0]); return 1; }
    double mean[ED] = {0};
    for (int n = 0; n < EN; ++n) for (int d = 0; d < ED; ++d) mean[d] += Xe[d + ED * n] / EN;
    for (int d = 0; d < ED; ++d) if (fabs(S1[d] / EN - mean[d]) > 1e-12) { printf("FAIL estep_full S1\n"); return 1; }
    if (vcmi_estep_full_stats_len(ED, 1) != 1 + ED + ED * ED + 1) { printf("FAIL stats len\n"); return 1; }
    sg0[0] = -1.0;                                       /* not positive definite -> reported, not crashed */
    if (vcmi_estep_full(Xe, EN, ED, 1, w1, mu0, sg0, S0, S1, S2, &ll) != VCMI_ERR_NOT_PD) { printf("FAIL expected VCMI_ERR_NOT_PD\n"); return 1; }
  }
  /* ---- diffgmm (src/diffgmm.jl): mu^y - mu^x, Sxy - Sxx, Sxx + Syy - Sxy - Syx */
  {
    double mo[DJ], so[DJ * DJ];
    CHECK(vcmi_diffgmm(mu, sig, DJ, 1, mo, so));
    for (int d = 0; d < D; ++d)
      if (fabs(mo[D + d] - (mu[D + d] - mu[d])) > 1e-15 || fabs(so[d + DJ * (D + d)] - (0.5 - 2.0)) > 1e-15 ||
          fabs(so[(D + d) + DJ * (D + d)] - (2.0 + 1.0 - 0.5 - 0.5)) > 1e-15) { printf("FAIL diffgmm\n"); return 1; }
  }
  printf("abi_smoke ok (devices %d, affine err %.2e, %s)\n", ndev, worst, vcmi_version());
  return 0;
}
