// dtw.hip -- placeholder translation unit (filled in below in this round)
#include "vcmi_common.hpp"
using namespace vcmi;
extern "C" int vcmi_dtw_fit(const double *, int64_t, const double *, int64_t, int, int, int, int64_t *, double *, int64_t *) { return fail(VCMI_ERR_ARG, "not implemented"); }
extern "C" int vcmi_dtw_fit_batch(int64_t, const double *const *, const int64_t *, const double *const *, const int64_t *, int, int, int, int64_t *const *) { return fail(VCMI_ERR_ARG, "not implemented"); }
extern "C" int vcmi_dtw_fit_batch_dev(int64_t, const double *, const int64_t *, const int64_t *, const int64_t *, const int64_t *, int, int, int, int64_t *, const int64_t *, void *) { return fail(VCMI_ERR_ARG, "not implemented"); }
extern "C" int vcmi_align(const double *, int64_t, const double *, int64_t, int, double *, int64_t *) { return fail(VCMI_ERR_ARG, "not implemented"); }
extern "C" int vcmi_align_batch(int64_t, const double *const *, const int64_t *, const double *const *, const int64_t *, int, double *const *) { return fail(VCMI_ERR_ARG, "not implemented"); }
