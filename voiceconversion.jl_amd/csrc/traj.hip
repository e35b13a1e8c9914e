// traj.hip -- placeholder translation unit (filled in below in this round)
#include "vcmi_common.hpp"
using namespace vcmi;
extern "C" int vcmi_traj_create(vcmi_gmmmap *, int64_t, vcmi_traj **) { return fail(VCMI_ERR_ARG, "not implemented"); }
extern "C" int vcmi_traj_destroy(vcmi_traj *) { return VCMI_OK; }
extern "C" int64_t vcmi_traj_length(const vcmi_traj *) { return 0; }
extern "C" int vcmi_traj_convert(vcmi_traj *, const double *, int64_t, double *) { return fail(VCMI_ERR_ARG, "not implemented"); }
extern "C" int vcmi_traj_convert_batch(vcmi_traj *, int64_t, const double *const *, const int64_t *, double *const *) { return fail(VCMI_ERR_ARG, "not implemented"); }
extern "C" int vcmi_traj_convert_batch_dev(vcmi_traj *, int64_t, const double *, const int64_t *, const int64_t *, double *, const int64_t *, void *) { return fail(VCMI_ERR_ARG, "not implemented"); }
extern "C" int vcmi_vc_traj(vcmi_traj *, const double *, int64_t, double *) { return fail(VCMI_ERR_ARG, "not implemented"); }
extern "C" int vcmi_push_delta(const double *, int, int64_t, double *) { return fail(VCMI_ERR_ARG, "not implemented"); }
