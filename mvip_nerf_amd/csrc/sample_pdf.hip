// Hierarchical inverse-CDF sampling fused with the coarse+fine depth merge
// (DS_NeRF/run_nerf_helpers.py:304-347 sample_pdf; DS_NeRF/run.py:1809-1816 mids / sort(cat);
//  :1836 z_std).
//
// One wavefront per ray.  Element e of a per-ray vector lives in lane e%64, register e/64
// ("strided"), so every global access is a contiguous 256-byte run per register.  The reference configuration (64 coarse
// depths, <= 64 new samples) takes the one-register-per-lane path of sample_pdf_device.h:
//   * pdf normaliser = fp64 wave sum, CDF = fp64 wave scan on DPP row operations, rounded to fp32 per entry (what torch's
//     CPU sum / cumsum do for fp32: acc_type = double);
//   * searchsorted(right=True) = a six-step binary search on cross-lane reads of the sorted CDF, the 2-point gathers of
//     cdf / bins are cross-lane reads;
//   * sort(cat[z, z_samples]) = a merge BY RANK of two sorted lists (the inverse CDF's own interval index is the rank hint;
//     random uniforms are sorted first by a 21-step DPP bitonic network); the general / unsorted case keeps the bitonic
//     network over registers.
// The kernel is bound by VALU issue, not by bandwidth (~320 vector instructions per ray on the deterministic path after
// round 5: scalar row bases through readfirstlane, zero-filling DPP moves in the fp64 scans, v_rcp / v_sqrt for the
// statistic, LDS address space spelled out for the rank row; ~400 before).
// HBM traffic per ray (Nc=Nf=64): 512 B in (z, weights) + 256 B (u) and 512+256+4 B out.
#include "sample_pdf_device.h"
#include <stdlib.h>

namespace mvip {

// RPW rays per wavefront, consecutive, the next ray's rows requested BEFORE the current ray is worked on.  One ray per wave
// (RPW = 1) is the right shape while the rows come out of the L2 / Infinity Cache (a frame's 190,512 rays: 0.079 / 0.125 ms
// against 0.085 / 0.137 with four rays per wave, round 5); past the Infinity Cache the launch is bound by
// latency x bytes in flight (32 waves per CU x 768 B = 24 KB per CU: 2.5 TB/s whatever the arithmetic does, round 6) and the
// prefetching form doubles what a wave keeps in flight.  The host picks RPW by the ray count (MVIP_SAMPLE_RPW overrides).
// counting: 0 = sort unsorted samples (the round-5 route; MVIP_SAMPLE_COUNTING=0, A/B switch), 1 = counting_merge64.
template <int IT, int RPW>
__global__ __launch_bounds__(256) void sample_pdf_merge_kernel(
    const float *__restrict__ z, const float *__restrict__ weights, const float *__restrict__ u, int u_is_row,
    int64_t B, int Nc, int Nf, float *__restrict__ z_samples, float *__restrict__ z_merged,
    float *__restrict__ z_std, int64_t *__restrict__ inds_out, float *__restrict__ cdf_out, int counting) {
    __shared__ int rank_rows[4][RANK_LDS_WORDS];     // per wave: the count row of rank_merge64 + the rows of counting_merge64
    // the wave index through readfirstlane: the ray number and every row base derived from it are then SCALAR values (the
    // compiler cannot know that threadIdx.x >> 6 is wave-uniform; as a vector value each of the eight row accesses cost ~8
    // VALU instructions of 64-bit address arithmetic)
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t ray0 = ((int64_t)blockIdx.x * 4 + wave_in_block) * RPW;
    if (ray0 >= B) return;
    const int l = lane_id();
    const int nb = Nc - 1;                           // midpoints
    auto load = [&](int64_t ray, float (&zc)[IT], float (&wts)[IT], float (&uu)[IT]) {
#pragma unroll
        for (int i = 0; i < IT; ++i) {
            const int e = i * 64 + l;
            zc[i] = e < Nc ? z[ray * Nc + e] : 0.f;
            // weights[..., 1:-1]: weight e of the pdf is coarse weight e+1
            wts[i] = e < nb - 1 ? weights[ray * Nc + e + 1] : 0.f;
            uu[i] = e < Nf ? (u_is_row ? u[e] : u[ray * Nf + e]) : 2.f;
        }
    };
    int *lds_row = counting ? rank_rows[wave_in_block] : nullptr;
    float zc[IT], wts[IT], uu[IT];
    load(ray0, zc, wts, uu);
    if constexpr (RPW == 1) {
        sample_merge_ray<IT>(zc, wts, uu, ray0, Nc, Nf, z_samples, z_merged, z_std, inds_out, cdf_out, rank_rows[wave_in_block], counting != 0);
    } else {
        (void)lds_row;
#pragma unroll 1
        for (int k = 0; k < RPW; ++k) {
            const int64_t ray = ray0 + k;
            if (ray >= B) break;
            float zn[IT], wn[IT], un[IT];
            const bool more = k + 1 < RPW && ray + 1 < B;
            if (more) load(ray + 1, zn, wn, un);
            sample_merge_ray<IT>(zc, wts, uu, ray, Nc, Nf, z_samples, z_merged, z_std, inds_out, cdf_out, rank_rows[wave_in_block], counting != 0);
            if (more) {
#pragma unroll
                for (int i = 0; i < IT; ++i) { zc[i] = zn[i]; wts[i] = wn[i]; uu[i] = un[i]; }
            }
        }
    }
}

template <int IT>
__global__ __launch_bounds__(256) void sample_pdf_kernel(
    const float *__restrict__ bins_in, const float *__restrict__ weights, const float *__restrict__ u,
    int u_is_row, int64_t B, int Nb, int Nf, float *__restrict__ samples, int64_t *__restrict__ inds_out,
    float *__restrict__ cdf_out) {
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // see sample_pdf_merge_kernel
    const int64_t ray = (int64_t)blockIdx.x * 4 + wave_in_block;
    if (ray >= B) return;
    const int l = lane_id();
    float bins[IT], wts[IT], uu[IT], smp[IT], cdf[IT];
    int inds[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int e = i * 64 + l;
        bins[i] = e < Nb ? bins_in[ray * Nb + e] : 0.f;
        wts[i] = e < Nb - 1 ? weights[ray * (Nb - 1) + e] : 0.f;
        uu[i] = e < Nf ? (u_is_row ? u[e] : u[ray * Nf + e]) : 2.f;
    }
    inverse_cdf<IT>(bins, wts, Nb, uu, Nf, smp, inds, cdf);
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int e = i * 64 + l;
        if (e < Nf) {
            samples[ray * Nf + e] = smp[i];
            if (inds_out) inds_out[ray * Nf + e] = inds[i];
        }
        if (cdf_out && e < Nb) cdf_out[ray * Nb + e] = cdf[i];
    }
}

}  // namespace mvip

using namespace mvip;

extern "C" int mvip_sample_pdf_merge(const float *z, const float *weights, const float *u, int u_is_row,
                                     int64_t B, int Nc, int Nf, float *z_samples, float *z_merged, float *z_std,
                                     int64_t *inds, float *cdf, void *stream) {
    if (B < 0 || Nc < 3 || Nf < 1) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    if (!z || !weights || !u || !z_samples || !z_merged || !z_std) return MVIP_EINVAL;
    const int mx = Nc > Nf ? Nc : Nf;
    static const int counting = [] { const char *e = getenv("MVIP_SAMPLE_COUNTING"); return e ? atoi(e) : 1; }();     // A/B switch
    static const int rpw_env = [] { const char *e = getenv("MVIP_SAMPLE_RPW"); return e ? atoi(e) : 0; }();           // tuning
    // rays per wave: 1 while a launch's rows (1,540 B per ray) stay inside the 256 MB Infinity Cache, 4 with prefetch beyond
    const int rpw = mx > 64 ? 1 : (rpw_env == 1 || rpw_env == 2 || rpw_env == 4 ? rpw_env : (B * (int64_t)(Nc + 2 * Nf + Nc + Nf) * 4 > ((int64_t)200 << 20) ? 4 : 1));
    const dim3 grid((unsigned)((B + 4 * rpw - 1) / (4 * rpw))), block(256);
#define CALL(I, R) hipLaunchKernelGGL((sample_pdf_merge_kernel<I, R>), grid, block, 0, as_stream(stream), z, weights, u, \
                                      u_is_row, B, Nc, Nf, z_samples, z_merged, z_std, inds, cdf, counting)
    if (mx <= 64) { if (rpw == 4) CALL(1, 4); else if (rpw == 2) CALL(1, 2); else CALL(1, 1); }
    else if (mx <= 128) { CALL(2, 1); } else if (mx <= 256) { CALL(4, 1); } else return MVIP_EUNSUP;
#undef CALL
    return check_launch();
}

extern "C" int mvip_sample_pdf(const float *bins, const float *weights, const float *u, int u_is_row, int64_t B,
                               int Nb, int Nf, float *samples, int64_t *inds, float *cdf, void *stream) {
    if (B < 0 || Nb < 2 || Nf < 1) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    if (!bins || !weights || !u || !samples) return MVIP_EINVAL;
    const int mx = Nb > Nf ? Nb : Nf;
    const dim3 grid((unsigned)((B + 3) / 4)), block(256);
#define CALL(I) hipLaunchKernelGGL(sample_pdf_kernel<I>, grid, block, 0, as_stream(stream), bins, weights, u, \
                                   u_is_row, B, Nb, Nf, samples, inds, cdf)
    if (mx <= 64) { CALL(1); } else if (mx <= 128) { CALL(2); } else if (mx <= 256) { CALL(4); } else return MVIP_EUNSUP;
#undef CALL
    return check_launch();
}
