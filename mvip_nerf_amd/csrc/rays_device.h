// Device side of the stratified depth sampling (DS_NeRF/run.py:1759-1781), shared by csrc/rays.hip (stand-alone launch) and
// the fused coarse pass of csrc/mlp_fwd16.hip.
#pragma once
#include "common.h"

namespace mvip {

__device__ __forceinline__ float z_at(float near, float far, float t, int lindisp) {
    if (lindisp) return 1.f / ((1.f / near) * (1.f - t) + (1.f / far) * t);
    return near * (1.f - t) + far * t;
}

// depth of sample s of S along a ray with bounds (near, far): the bin centre, or -- tr != nullptr -- the point
// lower + (upper - lower) * *tr of its stratum (run.py:1771-1781)
__device__ __forceinline__ float stratified_point(float near, float far, const float *__restrict__ t_vals, int s, int S,
                                                  int lindisp, const float *__restrict__ tr) {
    const float zc = z_at(near, far, t_vals[s], lindisp);
    if (!tr) return zc;
    const float zl = s > 0 ? z_at(near, far, t_vals[s - 1], lindisp) : zc;
    const float zr = s < S - 1 ? z_at(near, far, t_vals[s + 1], lindisp) : zc;
    const float upper = s < S - 1 ? .5f * (zr + zc) : zc;
    const float lower = s > 0 ? .5f * (zc + zl) : zc;
    return lower + (upper - lower) * *tr;
}

}  // namespace mvip
