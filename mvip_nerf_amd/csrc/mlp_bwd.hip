// placeholder: replaced by the real backward in the next milestone
#include "common.h"
extern "C" int64_t mvip_mlp_backward_workspace_bytes(int64_t) { return 0; }
extern "C" int mvip_mlp_backward_rays(const float *, const float *, const float *, int64_t, int, const float *,
                                      float *, void *, int64_t, int, void *) { return MVIP_EUNSUP; }
extern "C" int mvip_mlp_backward_points(const float *, const float *, const float *, int64_t, const float *,
                                        float *, void *, int64_t, int, void *) { return MVIP_EUNSUP; }
