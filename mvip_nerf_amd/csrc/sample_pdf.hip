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
// Round 6: the one-ray-per-wave kernel is bound by the LATENCY of a ray's dependent chain times the eight waves a SIMD holds
// (dynamic-LDS occupancy probe: time x1.7 from 2 to 4 waves per SIMD, x1.5 from 4 to 8; r6_sample_merge_occupancy_probe.json);
// the reference configuration therefore runs sample_pdf_merge_pair_kernel: N = 2 rays per wave on a branch-free route whose
// steps are written ray 0, ray 1, ray 0, ... (sample_pdf_device.h::rays_fast), after which the launch is bound by VALU ISSUE --
// 2, 3 and 4 rays per wave run in the same time -- and the instruction count is what matters: the sorting network on signed
// keys (2 instructions per step instead of 3-7), wave totals without masked steps (252 vector instructions per ray on the
// random-uniform route, 191 on the deterministic one; 300 / 210 before).  Measured (r6_sample_merge_rays_per_wave_ab.jsonl):
// random uniforms 0.120 -> 0.093 ms per 190,512 rays (0.305 -> 0.395 of 8 TB/s), 8 frames 0.99 -> 0.80 ms (0.295 -> 0.365).
// HBM traffic per ray (Nc=Nf=64): 512 B in (z, weights) + 256 B (u) and 512+256+4 B out.
#include "sample_pdf_device.h"
#include <stdlib.h>

namespace mvip {

// One ray per wavefront: every shape but 64 + 64.  (Round 5 tried four CONSECUTIVE rays per wave with the next ray's rows requested
// before the current ray is worked on: 0.085 / 0.137 ms against 0.079 / 0.125 -- prefetching the rows does not shorten the chain.)
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

// Two rays per wavefront for the reference configuration (64 + 64), both on rays_fast's branch-free route, their chains
// interleaved step by step (sample_pdf_device.h): twice the independent work per wave where the one-ray kernel is bound by latency x the eight waves a
// SIMD holds.  SORT: random uniforms (the new samples are sorted by the network); otherwise the shared deterministic row.
template <bool SORT, int N>
__global__ __launch_bounds__(256) void sample_pdf_merge_pair_kernel(
    const float *__restrict__ z, const float *__restrict__ weights, const float *__restrict__ u, int u_is_row, int64_t B,
    float *__restrict__ z_samples, float *__restrict__ z_merged, float *__restrict__ z_std, int64_t *__restrict__ inds_out,
    float *__restrict__ cdf_out) {
    __shared__ int rank_rows[4][N][80];
    const int wave_in_block = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t ray0 = ((int64_t)blockIdx.x * 4 + wave_in_block) * N;
    if (ray0 >= B) return;
    const int l = lane_id();
    const int lw = min(l + 1, 63);                   // weights[..., 1:-1]: weight e of the pdf is coarse weight e + 1 (e < 62)
    float zz[N], ww[N], uu[N];
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int64_t ray = min(ray0 + k, B - 1);    // a short tail: the spare chains redo the last ray and store nothing
        zz[k] = z[ray * 64 + l];
        const float w = weights[ray * 64 + lw];
        ww[k] = l < 62 ? w : 0.f;
        uu[k] = u_is_row ? u[l] : u[ray * 64 + l];
    }
    RaysFast<N> R;
    rays_fast<SORT, N>(zz, ww, uu, &rank_rows[wave_in_block][0][0], 80, R);
#pragma unroll
    for (int k = 0; k < N; ++k) {
        const int64_t ray = ray0 + k;
        if (ray >= B) break;
        if (R.ok[k]) {
            if (z_samples) z_samples[ray * 64 + l] = R.smp[k];
            if (inds_out) inds_out[ray * 64 + l] = R.ind[k];
            if (cdf_out && l < 63) cdf_out[ray * 63 + l] = R.cdf[k];
            if (l == 0) z_std[ray] = R.zstd[k];
            float *out_row = z_merged + ray * 128;
            out_row[R.pa[k]] = zz[k];
            out_row[R.pb[k]] = R.b_key[k];
        } else {                                     // negative pdf entry, NaN, unsorted depths, a hint that does not bracket: the general route
            const float z1[1] = {zz[k]}, w1[1] = {ww[k]}, u1[1] = {uu[k]};
            sample_merge_ray<1>(z1, w1, u1, ray, 64, 64, z_samples, z_merged, z_std, inds_out, cdf_out, rank_rows[wave_in_block][k]);
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
    static const int pair_env = [] { const char *e = getenv("MVIP_SAMPLE_PAIR"); return e ? atoi(e) : 1; }();      // A/B switch
    if (Nc == 64 && Nf == 64 && pair_env) {
        const dim3 block2(256);
#define PAIR(SORT, N) hipLaunchKernelGGL((sample_pdf_merge_pair_kernel<SORT, N>), dim3((unsigned)((B + 4 * N - 1) / (4 * N))), block2, 0, \
                                         as_stream(stream), z, weights, u, u_is_row, B, z_samples, z_merged, z_std, inds, cdf)
        if (pair_env == 3) { if (u_is_row) PAIR(false, 3); else PAIR(true, 3); }
        else if (pair_env == 4) { if (u_is_row) PAIR(false, 4); else PAIR(true, 4); }
        else { if (u_is_row) PAIR(false, 2); else PAIR(true, 2); }
#undef PAIR
        return check_launch();
    }
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
