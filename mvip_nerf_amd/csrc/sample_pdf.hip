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

namespace mvip {

// One ray per wavefront.  (Round 5 tried four consecutive rays per wave with the next ray's rows requested before the current
// ray is worked on: 0.085 / 0.137 ms against 0.079 / 0.125 -- the launch is not bound by latency x occupancy.)
template <int IT>
__global__ __launch_bounds__(256) void sample_pdf_merge_kernel(
    const float *__restrict__ z, const float *__restrict__ weights, const float *__restrict__ u, int u_is_row,
    int64_t B, int Nc, int Nf, float *__restrict__ z_samples, float *__restrict__ z_merged,
    float *__restrict__ z_std, int64_t *__restrict__ inds_out, float *__restrict__ cdf_out) {
    __shared__ int rank_rows[4][80];                 // per wave: the 65-word row of rank_merge64's prefix-maximum count
    // the wave index through readfirstlane: the ray number and every row base derived from it are then SCALAR values (the
    // compiler cannot know that threadIdx.x >> 6 is wave-uniform; as a vector value each of the eight row accesses cost ~8
    // VALU instructions of 64-bit address arithmetic)
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t ray = (int64_t)blockIdx.x * 4 + wave_in_block;
    if (ray >= B) return;
    const int l = lane_id();
    const int nb = Nc - 1;                           // midpoints
    float zc[IT], wts[IT], uu[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int e = i * 64 + l;
        zc[i] = e < Nc ? z[ray * Nc + e] : 0.f;
        // weights[..., 1:-1]: weight e of the pdf is coarse weight e+1
        wts[i] = e < nb - 1 ? weights[ray * Nc + e + 1] : 0.f;
        uu[i] = e < Nf ? (u_is_row ? u[e] : u[ray * Nf + e]) : 2.f;
    }
    sample_merge_ray<IT>(zc, wts, uu, ray, Nc, Nf, z_samples, z_merged, z_std, inds_out, cdf_out, rank_rows[wave_in_block]);
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
    const dim3 grid((unsigned)((B + 3) / 4)), block(256);
#define CALL(I) hipLaunchKernelGGL(sample_pdf_merge_kernel<I>, grid, block, 0, as_stream(stream), z, weights, u, \
                                   u_is_row, B, Nc, Nf, z_samples, z_merged, z_std, inds, cdf)
    if (mx <= 64) { CALL(1); } else if (mx <= 128) { CALL(2); } else if (mx <= 256) { CALL(4); } else return MVIP_EUNSUP;
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
