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


// ---- the same primitives on DPP (data-parallel primitives: cross-lane operands of ordinary VALU instructions, ~8 cycles)
// instead of ds_bpermute shuffles (LDS crossbar, ~100 cycles of dependent latency each).  gfx9 controls: row_shr:n = 0x110+n,
// row_ror:n = 0x120+n, wave_shl:1 = 0x130, wave_shr:1 = 0x138, row_bcast:15 = 0x142, row_bcast:31 = 0x143, quad_perm = 0x00..0xff.
// A lane whose DPP source does not exist (or whose row / bank is masked off) keeps `old`.
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_f32(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL,
                                                                 ROW_MASK, BANK_MASK, false));
}
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ double dpp_f64(double old, double v) {
    const long long o = __builtin_bit_cast(long long, old), x = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp((int)o, (int)x, CTRL, ROW_MASK, BANK_MASK, false);
    const int hi = __builtin_amdgcn_update_dpp((int)(o >> 32), (int)(x >> 32), CTRL, ROW_MASK, BANK_MASK, false);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
// lane l <- lane l - 1 (lane 0 <- first) / lane l <- lane l + 1 (lane 63 <- last)
__device__ __forceinline__ float dpp_from_prev(float v, float first) { return dpp_f32<0x138>(first, v); }
__device__ __forceinline__ float dpp_from_next(float v, float last) { return dpp_f32<0x130>(last, v); }

// inclusive prefix sum over the 64 lanes: row_shr 1, 2, 4, 8 inside the rows of 16, then the row totals by row_bcast:15 (into
// rows 1 and 3) and row_bcast:31 (into rows 2 and 3) -- six dependent VALU steps
template <class T, class Dpp>
__device__ __forceinline__ T dpp_incl_scan_add(T v, Dpp) {
    v += Dpp::template z<0x111>(v);
    v += Dpp::template z<0x112>(v);
    v += Dpp::template z<0x114>(v);
    v += Dpp::template z<0x118>(v);
    v += Dpp::template f<0x142, 0xa>((T)0, v);
    v += Dpp::template f<0x143, 0xc>((T)0, v);
    return v;
}
// inclusive prefix PRODUCT over the 64 lanes, same six steps (a lane without a source multiplies by 1)
__device__ __forceinline__ float dpp_incl_prod(float v) {
    v *= dpp_f32<0x111>(1.f, v);
    v *= dpp_f32<0x112>(1.f, v);
    v *= dpp_f32<0x114>(1.f, v);
    v *= dpp_f32<0x118>(1.f, v);
    v *= dpp_f32<0x142, 0xa>(1.f, v);
    v *= dpp_f32<0x143, 0xc>(1.f, v);
    return v;
}
// z<C>: the unmasked steps, a lane without a source reads zero (bound_ctrl) -- for the 64-bit form that is ONE v_mov_b32_dpp per
// half with no zero-initialised destination (the masked form needs the "old" value in place: two more moves per step)
template <int CTRL>
__device__ __forceinline__ double dpp_f64_zero(double v) {
    const long long x = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)x, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(x >> 32), CTRL, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
struct DppF32 {
    template <int C, int R = 0xf, int B = 0xf> static __device__ __forceinline__ float f(float o, float v) { return dpp_f32<C, R, B>(o, v); }
    template <int C> static __device__ __forceinline__ float z(float v) { return dpp_f32<C>(0.f, v); }
};
struct DppF32Z {      // every step through mov_dpp with bound_ctrl: a lane without a source reads zero, no "old" value to put in place
    template <int C> static __device__ __forceinline__ float z(float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), C, 0xf, 0xf, true));
    }
};
struct DppF64 {
    template <int C, int R = 0xf, int B = 0xf> static __device__ __forceinline__ double f(double o, double v) { return dpp_f64<C, R, B>(o, v); }
    template <int C> static __device__ __forceinline__ double z(double v) { return dpp_f64_zero<C>(v); }
};
__device__ __forceinline__ float dpp_incl_sum(float v) { return dpp_incl_scan_add<float>(v, DppF32{}); }
__device__ __forceinline__ double dpp_incl_sum(double v) { return dpp_incl_scan_add<double>(v, DppF64{}); }
// total over the wave, in every lane (the scan's last lane, read back through a scalar register)
__device__ __forceinline__ float dpp_wave_sum(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dpp_incl_sum(v)), 63));
}
__device__ __forceinline__ double dpp_wave_sum(double v) {
    const long long t = __builtin_bit_cast(long long, dpp_incl_sum(v));
    const int lo = __builtin_amdgcn_readlane((int)t, 63), hi = __builtin_amdgcn_readlane((int)(t >> 32), 63);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
}
// value of lane (l ^ J) for J a power of two: quad permutes, two masked row shifts (4), a row rotation (8), ds_swizzle (16),
// the half-wave swap (32)
template <int J>
__device__ __forceinline__ float dpp_xor(float v) {
    if constexpr (J == 1) return dpp_f32<0xB1>(v, v);                       // quad_perm [1,0,3,2]
    else if constexpr (J == 2) return dpp_f32<0x4E>(v, v);                  // quad_perm [2,3,0,1]
    else if constexpr (J == 4) return dpp_f32<0x114, 0xf, 0xa>(dpp_f32<0x104, 0xf, 0x5>(v, v), v);   // banks 0,2 <- +4; banks 1,3 <- -4
    else if constexpr (J == 8) return dpp_f32<0x128>(v, v);                 // row_ror:8
    else if constexpr (J == 16) return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), (16 << 10) | 0x1f));
    else {
        float a = v, b = v;
        asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));   // a = {lo, lo}, b = {hi, hi}
        return (threadIdx.x & 32) ? a : b;
    }
}

}  // namespace mvip
