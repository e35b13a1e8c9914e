// estep.hip -- placeholder translation unit (filled in below in this round)
#include "vcmi_common.hpp"
using namespace vcmi;
extern "C" int vcmi_estep_diag(const double *, int64_t, int, int, const double *, const double *, const double *, double *, double *, double *, double *) { return fail(VCMI_ERR_ARG, "not implemented"); }
extern "C" int64_t vcmi_estep_stats_len(int Dj, int M) { return (int64_t)M * (1 + 2 * Dj) + 1; }
extern "C" int vcmi_estep_diag_dev(const double *, int64_t, int, int, const double *, const double *, const double *, double *, void *) { return fail(VCMI_ERR_ARG, "not implemented"); }
