// Fused inference of the reference's second model, NeRF_TCNN.forward (DS_NeRF/run_nerf_helpers_tcnn.py:88-112):
//     point, direction -> hash-grid gather (16 levels) -> 32->64->16 sigma MLP -> SH4(dir) ++ 15 geometry
//     features ++ 1.0 -> 32->64->64->16 colour MLP -> (r, g, b, sigma)
// in one kernel, for the no-grad passes (frame renders, neighbour views, evaluation).  The unfused path writes
// and re-reads ~1.5 KB of activations per point around five skinny library GEMMs; here a wavefront owns 32
// points, the gathered features never leave registers and the five layers run on the matrix pipe in exact fp32
// (v_mfma_f32_32x32x2_f32, 192 steps per 32 points).
//
// Lane (j = lane & 31, h = lane >> 5) owns point j of the tile and HALF of its per-point inputs:
//   * hash-grid levels h, 2+h, .. 14+h (16 features).  K = 2 MFMA step s takes feature 4(s>>1) + 2h + (s&1) from
//     lane (j, h), so the first layer's A operand is that column of W1 -- the k order is a property of the packed
//     image only.
//   * an accumulator tile holds row 8q + 4h + s' in register 4q + s' of lane (j, h); using register r directly as
//     the B operand of the next layer's step contributes hidden units {32t + 8q + s', +4}, again matched by the
//     packed A operand (same trick as csrc/mlp_layout.h).  Activations never move between lanes.
//   * the colour network's 32 inputs: registers 0..7 of the sigma network's output tile are rows
//     {0-3, 8-11} (h = 0) / {4-7, 12-15} (h = 1) = sigma + the 15 geometry features; lane (j, 0) replaces row 0
//     (sigma, not an input) by the constant 1.0 that pads the 31 inputs to 32, and each lane evaluates SH
//     coefficients 8h .. 8h+7 itself.
// The 16-row output layers are padded to 32-row tiles with zero rows (half of their MFMA work is padding; they are
// a third of the steps).  The packed weights (48 KB) sit in LDS, two workgroups per CU, so one wave's gathers
// run under the other wave's MFMAs.
#include "common.h"

namespace mvip {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

constexpr int HF_STEPS = 192;                    // S1 32, S2 32, C1 32, C2 64, C3 32
constexpr int HF_FLOATS = HF_STEPS * 64;         // 12288 floats = 48 KB
constexpr int HF_S2 = 32, HF_C1 = 64, HF_C2 = 96, HF_C3 = 160;

struct HgLevelF { float scale; uint32_t resolution, offset, size; };

// hash-grid feature (row of the 32-feature encoding) that lane half h supplies at first-layer step k
__host__ __device__ constexpr int hf_level(int l, int h) { return 2 * l + h; }
__host__ __device__ constexpr int hf_feature(int k, int h) { return 2 * hf_level(k >> 1, h) + (k & 1); }
// hidden unit that register r of lane half h feeds (see header)
__host__ __device__ constexpr int hf_unit(int k, int h) { return 32 * (k >> 4) + 8 * ((k & 15) >> 2) + 4 * h + (k & 3); }

// packed image: float index 4*(64*b) + 4*lane + s = A operand of step 4b + s for lane (i = lane & 31, h = lane >> 5)
__global__ void hgf_pack_kernel(const float *__restrict__ sig, const float *__restrict__ col, float *__restrict__ img) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= HF_FLOATS) return;
    const int b = idx >> 8, lane = (idx & 255) >> 2, s4 = idx & 3;
    const int i = lane & 31, h = lane >> 5, step = 4 * b + s4;
    const float *W1 = sig, *W2 = sig + 2048, *C1 = col, *C2 = col + 2048, *C3 = col + 6144;   // [out][in] row-major
    float v = 0.f;
    if (step < HF_S2) {
        const int t = step >> 4, k = step & 15;
        v = W1[(32 * t + i) * 32 + hf_feature(k, h)];
    } else if (step < HF_C1) {
        if (i < 16) v = W2[i * 64 + hf_unit(step - HF_S2, h)];
    } else if (step < HF_C2) {
        const int t = (step - HF_C1) >> 4, k = (step - HF_C1) & 15;
        int c;
        if (k < 8) {
            const int row = 8 * (k >> 2) + 4 * h + (k & 3);       // row of the sigma network's output
            c = row == 0 ? 31 : 15 + row;                          // inputs: [sh 0..15, h[1..15], 1.0]
        } else {
            c = 8 * h + (k - 8);
        }
        v = C1[(32 * t + i) * 32 + c];
    } else if (step < HF_C3) {
        const int t = (step - HF_C2) >> 5, k = (step - HF_C2) & 31;
        v = C2[(32 * t + i) * 64 + hf_unit(k, h)];
    } else {
        if (i < 16) v = C3[i * 64 + hf_unit(step - HF_C3, h)];
    }
    img[idx] = v;
}

// Entry indices of the 8 cell corners: the same values as hg_index() in hashgrid.hip (uint32 wrap-around
// arithmetic, then mod table size) without a 32-bit division per corner -- a dense level's index is below the
// table size for every in-range cell, and a hashed level's table is a power of two in every configuration
// tiny-cuda-nn produces (it only hashes when the size was capped at 2^log2_hashmap_size); both fall back to `%`.
__device__ __forceinline__ void hgf_corners(const uint32_t (&c)[3], uint32_t res, uint32_t size, uint32_t off,
                                            uint32_t (&e)[8]) {
    const bool dense = (uint64_t)res * res * res <= (uint64_t)size;
    if (dense) {
        const uint32_t r2 = res * res;
        const uint32_t b = c[0] + c[1] * res + c[2] * r2;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            uint32_t idx = b + (k & 1) + ((k >> 1) & 1) * res + ((k >> 2) & 1) * r2;
            if (idx >= size) idx %= size;
            e[k] = off + idx;
        }
    } else {
        const uint32_t mask = size - 1;
        const bool pow2 = (size & mask) == 0;
        const uint32_t hx[2] = {c[0], c[0] + 1};
        const uint32_t hy[2] = {c[1] * 2654435761u, (c[1] + 1) * 2654435761u};
        const uint32_t hz[2] = {c[2] * 805459861u, (c[2] + 1) * 805459861u};
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const uint32_t hsh = hx[k & 1] ^ hy[(k >> 1) & 1] ^ hz[(k >> 2) & 1];
            e[k] = off + (pow2 ? (hsh & mask) : hsh % size);
        }
    }
}

__device__ __forceinline__ f32x16 hf_mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ void hf_relu(f32x16 &a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
}

// one 32-row output tile over NK k-steps; BLK = first 1-KB block of its A operands; b(k) = B operand of step k
template <int BLK, int NK, class BOp>
__device__ __forceinline__ f32x16 hf_tile(const float *wl, int woff, BOp b) {
    f32x16 acc;
    __builtin_amdgcn_sched_barrier(0);            // keep this tile's operand reads below the previous tile
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int k4 = 0; k4 < NK / 4; ++k4) {
        const f32x4v a = *reinterpret_cast<const f32x4v *>(wl + (BLK + k4) * 256 + woff);
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = hf_mfma(a[s], b(4 * k4 + s), acc);
    }
    return acc;
}

#ifndef HGF_GROUP
#define HGF_GROUP 4
#endif
template <int HGF_OCC, int WAVES>
__global__ void __launch_bounds__(64 * WAVES, HGF_OCC)
hgf_forward_kernel(const float *__restrict__ x, const float *__restrict__ d, const float2 *__restrict__ table,
                   const HgLevelF *__restrict__ levels, const float *__restrict__ img, int64_t P, float bound,
                   float4 *__restrict__ out) {
    __shared__ float wl[HF_FLOATS];
    for (int i = threadIdx.x; i < HF_FLOATS / 4; i += 64 * WAVES)
        reinterpret_cast<float4 *>(wl)[i] = reinterpret_cast<const float4 *>(img)[i];
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int64_t ntiles = (P + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * WAVES + wave; tile < ntiles; tile += (int64_t)gridDim.x * WAVES) {
        // the weights are loop-invariant: without an opaque per-iteration offset the compiler hoists all 48 operand
        // reads (192 registers) out of the tile loop and spills
        int woff = lane * 4;
        asm volatile("" : "+v"(woff));
        const int64_t p = tile * 32 + j;
        const bool valid = p < P;
        const int64_t pc = valid ? p : P - 1;
        float xn[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = x[pc * 3 + a];
            if (bound > 0.f) v = (v + bound) / (2.f * bound);
            xn[a] = v;
        }
        // ---- hash-grid features of levels h, 2+h, .., 14+h
        // (gathers are issued HGF_GROUP levels at a time: all 64 of them in flight would need 128 data + 64
        //  address registers and spill; the byte offset is 32-bit against the uniform table base)
        float f[16];
#pragma unroll
        for (int g = 0; g < 8 / HGF_GROUP; ++g) {
            float2 v[HGF_GROUP][8];
            float w[HGF_GROUP][3];
#pragma unroll
            for (int u = 0; u < HGF_GROUP; ++u) {
                const HgLevelF L = levels[hf_level(HGF_GROUP * g + u, h)];
                uint32_t c[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    const float pos = xn[a] * L.scale + 0.5f;
                    const float fl = floorf(pos);
                    c[a] = (uint32_t)(int)fl;
                    w[u][a] = pos - fl;
                }
                uint32_t e[8];
                hgf_corners(c, L.resolution, L.size, L.offset, e);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    v[u][k] = *reinterpret_cast<const float2 *>(reinterpret_cast<const char *>(table) + e[k] * 8u);
            }
#pragma unroll
            for (int u = 0; u < HGF_GROUP; ++u) {
                float f0 = 0.f, f1 = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float wk = ((k & 1) ? w[u][0] : 1.f - w[u][0]) * (((k >> 1) & 1) ? w[u][1] : 1.f - w[u][1]) *
                                     (((k >> 2) & 1) ? w[u][2] : 1.f - w[u][2]);
                    f0 += wk * v[u][k].x;
                    f1 += wk * v[u][k].y;
                }
                f[2 * (HGF_GROUP * g + u)] = f0;
                f[2 * (HGF_GROUP * g + u) + 1] = f1;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- sigma network: 32 -> 64 (ReLU) -> 16
        f32x16 a0 = hf_tile<0, 16>(wl, woff, [&](int k) { return f[k]; });
        f32x16 a1 = hf_tile<4, 16>(wl, woff, [&](int k) { return f[k]; });
        hf_relu(a0);
        hf_relu(a1);
        const f32x16 hh = hf_tile<HF_S2 / 4, 32>(wl, woff, [&](int k) { return k < 16 ? a0[k & 15] : a1[k & 15]; });
        const float sigma = hh[0];                                   // row 0 lives in lane (j, 0)
        // ---- colour network inputs: 8 rows of hh (row 0 -> the 1.0 pad) + SH coefficients 8h .. 8h+7
        float cin[16];
#pragma unroll
        for (int s = 0; s < 8; ++s) cin[s] = hh[s];
        if (h == 0) cin[0] = 1.f;
        {
            const float dx = ((d[pc * 3 + 0] + 1.f) / 2.f) * 2.f - 1.f;
            const float dy = ((d[pc * 3 + 1] + 1.f) / 2.f) * 2.f - 1.f;
            const float dz = ((d[pc * 3 + 2] + 1.f) / 2.f) * 2.f - 1.f;
            const float xy = dx * dy, xz = dx * dz, yz = dy * dz, x2 = dx * dx, y2 = dy * dy, z2 = dz * dz;
            if (h == 0) {
                cin[8] = 0.28209479177387814f;
                cin[9] = -0.48860251190291987f * dy;
                cin[10] = 0.48860251190291987f * dz;
                cin[11] = -0.48860251190291987f * dx;
                cin[12] = 1.0925484305920792f * xy;
                cin[13] = -1.0925484305920792f * yz;
                cin[14] = 0.94617469575755997f * z2 - 0.31539156525251999f;
                cin[15] = -1.0925484305920792f * xz;
            } else {
                cin[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
                cin[9] = 0.59004358992664352f * dy * (-3.0f * x2 + y2);
                cin[10] = 2.8906114426405538f * xy * dz;
                cin[11] = 0.45704579946446572f * dy * (1.0f - 5.0f * z2);
                cin[12] = 0.3731763325901154f * dz * (5.0f * z2 - 3.0f);
                cin[13] = 0.45704579946446572f * dx * (1.0f - 5.0f * z2);
                cin[14] = 1.4453057213202769f * dz * (x2 - y2);
                cin[15] = 0.59004358992664352f * dx * (-x2 + 3.0f * y2);
            }
        }
        // ---- colour network: 32 -> 64 (ReLU) -> 64 (ReLU) -> 16
        f32x16 c0 = hf_tile<HF_C1 / 4, 16>(wl, woff, [&](int k) { return cin[k]; });
        f32x16 c1 = hf_tile<HF_C1 / 4 + 4, 16>(wl, woff, [&](int k) { return cin[k]; });
        hf_relu(c0);
        hf_relu(c1);
        f32x16 e0 = hf_tile<HF_C2 / 4, 32>(wl, woff, [&](int k) { return k < 16 ? c0[k & 15] : c1[k & 15]; });
        f32x16 e1 = hf_tile<HF_C2 / 4 + 8, 32>(wl, woff, [&](int k) { return k < 16 ? c0[k & 15] : c1[k & 15]; });
        hf_relu(e0);
        hf_relu(e1);
        const f32x16 rgb = hf_tile<HF_C3 / 4, 32>(wl, woff, [&](int k) { return k < 16 ? e0[k & 15] : e1[k & 15]; });
        if (h == 0 && valid) out[p] = make_float4(rgb[0], rgb[1], rgb[2], sigma);
    }
}

}  // namespace mvip

using namespace mvip;

extern "C" int64_t mvip_hashgrid_mlp_packed_floats(void) { return HF_FLOATS; }

// sigma_params: [64x32 | 16x64] floats, colour_params: [64x32 | 64x64 | 16x64] floats ([out][in] row-major, the
// tiny-cuda-nn parameter order); img: HF_FLOATS floats
extern "C" int mvip_hashgrid_mlp_pack(const float *sigma_params, const float *colour_params, float *img, void *stream) {
    if (!sigma_params || !colour_params || !img) return MVIP_EINVAL;
    hipLaunchKernelGGL(hgf_pack_kernel, dim3(HF_FLOATS / 256), dim3(256), 0, as_stream(stream), sigma_params,
                       colour_params, img);
    return check_launch();
}

extern "C" int mvip_hashgrid_nerf_forward(const float *x, const float *dirs, const float *table, const void *levels,
                                          const float *img, int64_t P, float bound, float *raw, void *stream) {
    if (P < 0) return MVIP_EINVAL;
    if (P == 0) return MVIP_OK;
    if (!x || !dirs || !table || !levels || !img || !raw) return MVIP_EINVAL;
    const int64_t ntiles = (P + 31) / 32;
    // two waves per SIMD (2 workgroups of 4 waves per CU).  Measured on the bench frame: 3 waves/SIMD (168 registers,
    // small spills) 16.1 ms, 4 waves/SIMD (8- or 16-wave workgroups, 128 registers) 20.9 ms, this 14.9 ms -- the
    // gathers are throughput-bound in the texture-address / L2 path, more waves only thrash the L2.
    int64_t blocks = (ntiles + 3) / 4;
    if (blocks > 256 * 2) blocks = 256 * 2;                // persistent
    hipLaunchKernelGGL((hgf_forward_kernel<2, 4>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), x, dirs,
                       (const float2 *)table, (const HgLevelF *)levels, img, P, bound, (float4 *)raw);
    return check_launch();
}
