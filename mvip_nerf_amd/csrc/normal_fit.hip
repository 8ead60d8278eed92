// placeholder: replaced by the real kernels in a later milestone
#include "common.h"
extern "C" int mvip_normal_fit_forward(const float *, int, int, float, float, float, float, int, float *, float *,
                                       float *, void *) { return MVIP_EUNSUP; }
extern "C" int mvip_normal_fit_backward(const float *, const float *, const float *, const float *, int, int,
                                        float, float, float, float, int, float *, float *, void *) { return MVIP_EUNSUP; }
