/* vc_oracle_internal.h -- the oracle's model record, shared by vc_oracle.c and vc_oracle_gemm.c.  TEST INFRASTRUCTURE ONLY. */
#ifndef VC_ORACLE_INTERNAL_H
#define VC_ORACLE_INTERNAL_H
#include "vc_oracle.h"
struct vco_gmmmap {
  int D, M;
  double *w;                       /* (M) */
  double *mux, *muy;               /* (D,M) */
  double *Sxx, *Sxy, *Syx, *Syy;   /* (D,D,M) */
  double *A;                       /* Sigma^yx Sigma^xx^-1, (D,D,M)  src/gmmmap.jl:33-36 */
  double *L;                       /* Cholesky factor of Hermitian(Sigma^xx) (upper triangle mirrored), (D,D,M) */
  double *logdet;                  /* (M) */
};
#endif
