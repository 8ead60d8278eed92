// Multiresolution hash-grid encoding + degree-4 spherical harmonics of the reference's second model,
// NeRF_TCNN (DS_NeRF/run_nerf_helpers_tcnn.py:13-112: 16 levels x 2 features, 2^19 entries per level,
// base resolution 16, per_level_scale = (2048*bound/16)^(1/15)).  The reference gets both encodings from
// tiny-cuda-nn (NVIDIA-only, not in the reference tree); this restates the published algorithm
// (Mueller et al. 2022 / tiny-cuda-nn grid.h): per level
//     scale = 2^(level * log2 b) * N_min - 1,  resolution = ceil(scale) + 1
//     pos = x * scale + 0.5,  cell = floor(pos),  w = pos - cell          (linear interpolation)
//     index(c) = dense   (c.x + c.y*res + c.z*res^2)            if res^3 fits the level's table
//                hashed  (c.x*1 ^ c.y*2654435761 ^ c.z*805459861) otherwise,   then  mod table size
//     feature = sum over the 8 cell corners of w_corner * table[level][index(corner)]
//
// HBM-bound gathers: 16 levels x 8 corners x 8 B per point (1 KB/point algorithmic; the fine levels touch a
// distinct 64-byte sector per corner).  One thread per (point, level), blockIdx.y = level so a workgroup
// stays inside one level's table (the three coarse levels are dense and L2-resident); features are written
// level-major [32][P] so that both this kernel's stores and the following 32->64 GEMM read whole rows.
// Backward = the same walk; contributions to the same entry are first summed per workgroup tile in an LDS
// hash map, then added with fire-and-forget fp32 atomics.
#include "common.h"

namespace mvip {

constexpr int HG_LEVELS = 16;

struct HgLevel {           // one row of the level table (device memory, indexed by blockIdx.y)
    float scale;
    uint32_t resolution;
    uint32_t offset;       // first entry of the level (in entries of 2 floats)
    uint32_t size;         // entries in the level
};

__device__ __forceinline__ uint32_t hg_index(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t res, uint32_t size) {
    uint32_t stride = 1, index = 0;
    const uint32_t c[3] = {cx, cy, cz};
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (stride <= size) { index += c[d] * stride; stride *= res; }
    }
    if (size < stride) index = cx ^ (cy * 2654435761u) ^ (cz * 805459861u);
    return index % size;
}

__device__ __forceinline__ void hg_cell(const float *__restrict__ x, int64_t p, float bound, float scale, uint32_t (&c)[3],
                                        float (&w)[3]) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        float v = x[p * 3 + d];
        if (bound > 0.f) v = (v + bound) / (2.f * bound);          // NeRF_TCNN.forward: to [0, 1]
        const float pos = v * scale + 0.5f;
        const float fl = floorf(pos);
        c[d] = (uint32_t)(int)fl;
        w[d] = pos - fl;
    }
}

// grid (ceil(P/256), 16)
__global__ void __launch_bounds__(256)
hg_forward_kernel(const float *__restrict__ x, const float2 *__restrict__ table, const HgLevel *__restrict__ levels,
                  int64_t P, float bound, float *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const int level = blockIdx.y;
    const HgLevel L = levels[level];
    uint32_t c[3];
    float w[3];
    hg_cell(x, p, bound, L.scale, c, w);
    const float2 *tab = table + L.offset;
    float f0 = 0.f, f1 = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t cx = c[0] + (k & 1), cy = c[1] + ((k >> 1) & 1), cz = c[2] + ((k >> 2) & 1);
        const float wk = ((k & 1) ? w[0] : 1.f - w[0]) * (((k >> 1) & 1) ? w[1] : 1.f - w[1]) *
                         (((k >> 2) & 1) ? w[2] : 1.f - w[2]);
        const float2 v = tab[hg_index(cx, cy, cz, L.resolution, L.size)];
        f0 += wk * v.x;
        f1 += wk * v.y;
    }
    out[(int64_t)(2 * level) * P + p] = f0;
    out[(int64_t)(2 * level + 1) * P + p] = f1;
}

// Table-gradient accumulation.  Scene coordinates occupy a small part of the [-bound, bound] box, so on the
// coarse and middle levels thousands of samples hit the same few entries and plain global atomics serialise
// (measured: 547 ms for 8.4 M points).  Each workgroup therefore walks a tile of HG_TILE points of ONE level and
// first sums contributions per entry in an LDS hash map (8192 slots, open addressing, 4 probes, ds_add_f32); at the end of
// the tile every occupied slot is flushed with one pair of global atomics.  Contributions that find no slot
// (fine levels: nearly every entry distinct) go to global memory directly.
#ifndef HG_TILE_POINTS
#define HG_TILE_POINTS 4096
#endif
constexpr int HG_TILE = HG_TILE_POINTS;
#ifndef HG_SLOTS_LOG2
#define HG_SLOTS_LOG2 13      // 8192 slots = 96 KB: one workgroup per CU, but the middle levels aggregate (iteration 14.9 -> 14.4 ms vs 2048)
#endif
constexpr int HG_SLOTS = 1 << HG_SLOTS_LOG2;
constexpr uint32_t HG_EMPTY = 0xFFFFFFFFu;

//
// HALF2 (opt-in, NeRF_TCNN.table_grad_atomics = 'half2'): the memory-side atomic COUNT bounds this kernel,
// and gfx950 has a packed pair atomic only for 16-bit floats.  Contributions that miss the LDS map (the scattered
// fine levels) then go out as ONE global_atomic_pk_add_f16 carrying both features, scaled by a power of two taken
// from max|dout| (`scale2`, device memory), into a separate half-precision table; the per-tile sums of the LDS map
// stay fp32 atomics into the fp32 table.  The caller adds table_h / scale to the fp32 table.  This is the arithmetic
// tiny-cuda-nn itself uses for these gradients (__half2 atomics); fp32 pairs remain the default.
typedef _Float16 hg_h2 __attribute__((ext_vector_type(2)));

template <bool HALF2>
__global__ void __launch_bounds__(256)
hg_backward_kernel(const float *__restrict__ x, const float *__restrict__ dout, const HgLevel *__restrict__ levels,
                   int64_t P, float bound, float *__restrict__ dtable, hg_h2 *__restrict__ dtable_h,
                   const float *__restrict__ scale2) {
    __shared__ uint32_t keys[HG_SLOTS];
    __shared__ float acc0[HG_SLOTS], acc1[HG_SLOTS];
    for (int s = threadIdx.x; s < HG_SLOTS; s += 256) { keys[s] = HG_EMPTY; acc0[s] = 0.f; acc1[s] = 0.f; }
    __syncthreads();
    const int level = blockIdx.y;
    const HgLevel L = levels[level];
    float *tab = dtable + 2 * (int64_t)L.offset;
    hg_h2 *tab_h = HALF2 ? dtable_h + L.offset : nullptr;
    const float hs = HALF2 ? scale2[0] * 0.0625f : 1.f;        // max|dout| * hs in [2^5, 2^6): 2^10 of headroom
    const int64_t p0 = (int64_t)blockIdx.x * HG_TILE;
#pragma unroll 1
    for (int it = 0; it < HG_TILE / 256; ++it) {
        const int64_t p = p0 + it * 256 + threadIdx.x;
        if (p >= P) break;
        const float g0 = dout[(int64_t)(2 * level) * P + p], g1 = dout[(int64_t)(2 * level + 1) * P + p];
        if (g0 == 0.f && g1 == 0.f) continue;
        uint32_t c[3];
        float w[3];
        hg_cell(x, p, bound, L.scale, c, w);
#pragma unroll 1
        for (int k = 0; k < 8; ++k) {
            const uint32_t cx = c[0] + (k & 1), cy = c[1] + ((k >> 1) & 1), cz = c[2] + ((k >> 2) & 1);
            const float wk = ((k & 1) ? w[0] : 1.f - w[0]) * (((k >> 1) & 1) ? w[1] : 1.f - w[1]) *
                             (((k >> 2) & 1) ? w[2] : 1.f - w[2]);
            const uint32_t idx = hg_index(cx, cy, cz, L.resolution, L.size);
            uint32_t slot = (idx * 2654435761u) >> (32 - HG_SLOTS_LOG2);
            bool done = false;
#pragma unroll 1
            for (int probe = 0; probe < 4 && !done; ++probe) {
                const uint32_t prev = atomicCAS(&keys[slot], HG_EMPTY, idx);
                if (prev == HG_EMPTY || prev == idx) {
                    atomicAdd(&acc0[slot], wk * g0);
                    atomicAdd(&acc1[slot], wk * g1);
                    done = true;
                }
                slot = (slot + 1) & (HG_SLOTS - 1);
            }
            if (!done) {
                if constexpr (HALF2) {
                    const hg_h2 v = {(_Float16)(wk * g0 * hs), (_Float16)(wk * g1 * hs)};
                    __builtin_amdgcn_global_atomic_fadd_v2f16((__attribute__((address_space(1))) hg_h2 *)(tab_h + idx), v);
                } else {
                    unsafeAtomicAdd(tab + 2 * (int64_t)idx, wk * g0);
#ifndef MVIP_EXPERIMENT_HG_ONE_ATOMIC
                    unsafeAtomicAdd(tab + 2 * (int64_t)idx + 1, wk * g1);
#endif
                }
            }
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < HG_SLOTS; s += 256) {
        const uint32_t k = keys[s];
        if (k != HG_EMPTY) {
            unsafeAtomicAdd(tab + 2 * (int64_t)k, acc0[s]);
#ifndef MVIP_EXPERIMENT_HG_ONE_ATOMIC
            unsafeAtomicAdd(tab + 2 * (int64_t)k + 1, acc1[s]);
#endif
        }
    }
}

// degree-4 spherical harmonics (16 coefficients, tiny-cuda-nn / instant-ngp basis and sign convention) of
// the direction after NeRF_TCNN's (d+1)/2 and the encoder's 2u-1 round trip; out is [16][P].
__global__ void __launch_bounds__(256)
sh4_kernel(const float *__restrict__ d, int64_t P, float *__restrict__ out) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= P) return;
    const float x = ((d[p * 3 + 0] + 1.f) / 2.f) * 2.f - 1.f;
    const float y = ((d[p * 3 + 1] + 1.f) / 2.f) * 2.f - 1.f;
    const float z = ((d[p * 3 + 2] + 1.f) / 2.f) * 2.f - 1.f;
    const float xy = x * y, xz = x * z, yz = y * z, x2 = x * x, y2 = y * y, z2 = z * z;
    float o[16];
    o[0] = 0.28209479177387814f;
    o[1] = -0.48860251190291987f * y;
    o[2] = 0.48860251190291987f * z;
    o[3] = -0.48860251190291987f * x;
    o[4] = 1.0925484305920792f * xy;
    o[5] = -1.0925484305920792f * yz;
    o[6] = 0.94617469575755997f * z2 - 0.31539156525251999f;
    o[7] = -1.0925484305920792f * xz;
    o[8] = 0.54627421529603959f * x2 - 0.54627421529603959f * y2;
    o[9] = 0.59004358992664352f * y * (-3.0f * x2 + y2);
    o[10] = 2.8906114426405538f * xy * z;
    o[11] = 0.45704579946446572f * y * (1.0f - 5.0f * z2);
    o[12] = 0.3731763325901154f * z * (5.0f * z2 - 3.0f);
    o[13] = 0.45704579946446572f * x * (1.0f - 5.0f * z2);
    o[14] = 1.4453057213202769f * z * (x2 - y2);
    o[15] = 0.59004358992664352f * x * (-x2 + 3.0f * y2);
#pragma unroll
    for (int k = 0; k < 16; ++k) out[(int64_t)k * P + p] = o[k];
}

}  // namespace mvip

using namespace mvip;

// levels: [16][4] 32-bit words {scale (float bits), resolution, offset, size} in device memory
extern "C" int mvip_hashgrid_forward(const float *x, const float *table, const void *levels, int64_t P, float bound,
                                     float *features, void *stream) {
    if (P < 0) return MVIP_EINVAL;
    if (P == 0) return MVIP_OK;
    if (!x || !table || !levels || !features) return MVIP_EINVAL;
    const dim3 grid((unsigned)((P + 255) / 256), HG_LEVELS);
    hipLaunchKernelGGL(hg_forward_kernel, grid, dim3(256), 0, as_stream(stream), x, (const float2 *)table,
                       (const HgLevel *)levels, P, bound, features);
    return check_launch();
}

extern "C" int mvip_hashgrid_backward(const float *x, const float *d_features, const void *levels, int64_t P,
                                      float bound, float *d_table, void *stream) {
    if (P < 0) return MVIP_EINVAL;
    if (P == 0) return MVIP_OK;
    if (!x || !d_features || !levels || !d_table) return MVIP_EINVAL;
    const dim3 grid((unsigned)((P + HG_TILE - 1) / HG_TILE), HG_LEVELS);
    hipLaunchKernelGGL((hg_backward_kernel<false>), grid, dim3(256), 0, as_stream(stream), x, d_features,
                       (const HgLevel *)levels, P, bound, d_table, (hg_h2 *)nullptr, (const float *)nullptr);
    return check_launch();
}

// opt-in variant: d_table (fp32, [n_entries][2]) receives the per-tile sums, d_table_h2 ([n_entries] half pairs) the
// scattered contributions multiplied by scale2[0] / 16; scale2 = {s, 1/s} from mvip_absmax_scale(d_features).
// Zero both tables first; the table gradient is d_table + d_table_h2 * (16 * scale2[1]).
extern "C" int mvip_hashgrid_backward_half2(const float *x, const float *d_features, const void *levels, int64_t P,
                                            float bound, const float *scale2, float *d_table, void *d_table_h2,
                                            void *stream) {
    if (P < 0) return MVIP_EINVAL;
    if (P == 0) return MVIP_OK;
    if (!x || !d_features || !levels || !scale2 || !d_table || !d_table_h2) return MVIP_EINVAL;
    const dim3 grid((unsigned)((P + HG_TILE - 1) / HG_TILE), HG_LEVELS);
    hipLaunchKernelGGL((hg_backward_kernel<true>), grid, dim3(256), 0, as_stream(stream), x, d_features,
                       (const HgLevel *)levels, P, bound, d_table, (hg_h2 *)d_table_h2, scale2);
    return check_launch();
}

extern "C" int mvip_sh4(const float *dirs, int64_t P, float *out, void *stream) {
    if (P < 0) return MVIP_EINVAL;
    if (P == 0) return MVIP_OK;
    if (!dirs || !out) return MVIP_EINVAL;
    hipLaunchKernelGGL(sh4_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, as_stream(stream), dirs, P, out);
    return check_launch();
}
