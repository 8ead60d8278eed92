// Shared helpers for the gfx950 kernels of libmvipnerf.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/mvip_nerf.h"

#define MVIP_WAVE 64

namespace mvip {

void set_last_error(hipError_t e);

static inline int check_launch() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_last_error(e); return MVIP_ELAUNCH; }
    return MVIP_OK;
}

static inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Zeroing of small scratch words as a KERNEL, not hipMemsetAsync: inside a captured hipGraph the 16-byte memset
// node in front of an atomicMax reduction was seen to leave stale values on replay (the conv data-gradient scale of
// the graphed SDS step drifted from replay to replay); a kernel node is ordered like every other launch.
static __global__ void mvip_zero_words_kernel(unsigned *p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
static inline void zero_words(void *p, int n_words, hipStream_t st) {
    hipLaunchKernelGGL(mvip_zero_words_kernel, dim3((n_words + 63) / 64), dim3(64), 0, st, (unsigned *)p, n_words);
}

// ---- wave-level primitives (64 lanes) -------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// inclusive prefix sum / product over the 64 lanes (Hillis-Steele on shuffles)
__device__ __forceinline__ float wave_incl_sum(float v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(v, o, 64); if (l >= o) v += t; }
    return v;
}
__device__ __forceinline__ float wave_incl_prod(float v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { float t = __shfl_up(v, o, 64); if (l >= o) v *= t; }
    return v;
}
// inclusive suffix sum (lane l gets sum over lanes >= l)
__device__ __forceinline__ float wave_incl_suffix_sum(float v) {
    const int l = lane_id();
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { float t = __shfl_down(v, o, 64); if (l + o < 64) v += t; }
    return v;
}

}  // namespace mvip
