// Inference forward of the 8x256 NeRF MLP with TWO waves per SIMD (exact fp32, v_mfma_f32_16x16x4_f32).
// The NeRF module's no-grad exact-fp32 path (NeRF.two_wave_inference): 202.9 ms on the bench's fine pass = 142.7
// TFLOP/s = 0.907 of peak, against 205.5 ms / 0.896 for the 32-point kernel of mlp_fwd.hip.
//
// mlp_fwd.hip gives every wave 32 points: 256 units x 32 points of activations are 128 registers per set and
// the kernel needs ~450 registers, i.e. ONE wave per SIMD -- and with one wave per SIMD nothing co-issues with
// that wave's own MFMAs: every LDS read, wait, DMA set-up and epilogue instruction costs ~4 cycles of
// matrix-pipe idle time (measured: 0.90 of peak after trimming them to ~950 per layer).  Here a wave owns 16
// points and works on 16x16 accumulator tiles (4 registers): an activation set is 64 registers, the kernel fits
// 256, a workgroup is 8 waves = 2 per SIMD sharing ONE weight ring, and each wave's non-MFMA instructions issue
// under the other wave's MFMAs.  Same MFMA rate (1024 MACs per 32 cycles), same L2->LDS weight traffic per point
// (128 points per workgroup), twice the LDS operand reads per point (64 B/clk/CU of 128).
//
// Register trick, 16x16 edition: accumulator register i of lane (n = lane & 15, g = lane >> 4) holds row 4g+i of
// the tile, column (point) n -- which is a legal B operand of a K=4 step whose k-slot g is unit 16T + 4g + i.
// The packed image stores the matching A operands: block (to, ti) = 64 lanes x 4 floats, lane (m, g) holds
// W[16 to + m][16 ti + 4g + 0..3], i.e. four consecutive input units: one ds_read_b128 feeds four MFMAs.
// Blocks are streamed layer by layer, output tile by output tile: a 256-wide layer is 16 chunks of 16 KB.
// STASH = true is the training forward: every activation tile is also written to the stash the backward kernels read
// ([row tile of 32 units][point tile of 32][32][32] fp32, mlp_device.h) -- a wave's 16 x 16 tile is four stores of four
// 64-byte row segments, the two waves that share a point tile filling the other half of each 128-byte row.
#include <stdlib.h>
#include "mlp_device16.h"
#include "rays_device.h"
#include "composite_device.h"
#include "sample_pdf_device.h"

namespace mvip {
using namespace mlp;

namespace f16p {

// FUSE (rays form only; DS_NeRF/run.py:1703-1847 render_rays as TWO launches per chunk instead of six):
//   1  the COARSE pass: 64 samples per ray, a workgroup = two rays.  The depths are computed here (stratified_point, no z
//      tensor); after the network every wave evaluates the exponentials of raw2outputs for its 16 points, then the first
//      wave of each ray composites its 64 samples from those terms in LDS (scan + sums: round 5, see the tail), draws the
//      fine samples by inverse CDF from the weights still in its registers and
//      writes the merged 128 depths: rgb0 / disp0 / acc0 (/ alpha0), z_std, z_merged -- no raw, weights or depth tensor of
//      the coarse pass ever exists;
//   2  the FINE pass: 128 samples per ray, a workgroup = one ray; wave 0 composites after the network.
// Both run the same device functions as the stand-alone kernels (composite_device.h, sample_pdf_device.h,
// rays_device.h), so every output is bit-identical to the unfused path.
struct FuseArgs {
    const float *t_vals, *t_rand, *noise, *u;
    int u_is_row, lindisp, flags, Nf;
    float *rgb, *disp, *acc, *depth, *weights, *alpha, *z_merged, *z_std;
};

template <bool FROM_RAYS, bool STASH = false, int FUSE = 0>
__global__ void __launch_bounds__(512, 2)
mlp_forward16_kernel(const float *__restrict__ packed, const float *__restrict__ in_a, const float *__restrict__ in_b,
                     int64_t P, int S, float *__restrict__ raw, float *__restrict__ stash = nullptr, int64_t n_pt = 0,
                     const FuseArgs fa = FuseArgs{}) {
    __shared__ __attribute__((aligned(16))) float lds[LDS16_FLOATS];
    // The fused tail's copy of the workgroup's raw values lives in a ring slot the weight stream no longer uses: every wave
    // that has left the last layer is past the barrier that closed chunk TOTAL_CHUNKS - 2, so that chunk's slot is read by
    // nobody.  (A separate 2 KB array made the kernel 4-5 % slower: 80,896 instead of 78,848 bytes of LDS per workgroup --
    // measured 210 vs 201.5 ms on the fine pass with identical instructions in the network part; with 78,848 bytes the
    // next workgroup's waves evidently start flowing in while this one's last waves finish.)
    float *raw_s = lds + ((TOTAL_CHUNKS - 2) % NSLOT16) * CHUNK_FLOATS;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    int64_t p = (int64_t)blockIdx.x * WG_POINTS + wave * 16 + n;
    const bool live = p < P;
    if (!live) p = P - 1;

    Stream16 st{packed, lds, wave, lane};
    for (int b = wave; b < SEC_B_FLOATS / BLOCK_FLOATS; b += 8)
        glds<0>(packed + SEC_A_FLOATS + b * BLOCK_FLOATS + lane * 4, lds + RING16_FLOATS + b * BLOCK_FLOATS);
    st.issue_chunk(0, 0);
    st.issue_chunk(1, 1);

    float px, py, pz, vx, vy, vz;
    // fused tail (FUSE != 0): what it will need is fetched NOW and carried in a few registers through the network -- the next
    // sample's depth, this point's density noise and the ray's direction norm (every wave evaluates the exponentials of ITS
    // points after the network, see below), the tail wave's uniform; at the end of the workgroup a global-memory round trip
    // would be fully exposed
    float zz = 0.f, zn = 0.f, nz = 0.f, dnorm = 0.f, tail_u = 2.f;
    int s_idx = 0;
    if constexpr (FROM_RAYS) {
        const int64_t ray = p / S;
        const float *row = in_a + ray * 11;
        s_idx = (int)(p - ray * S);
        if constexpr (FUSE == 1) {
            zz = stratified_point(row[6], row[7], fa.t_vals, s_idx, S, fa.lindisp, fa.t_rand ? fa.t_rand + p : nullptr);
            if (s_idx + 1 < S) zn = stratified_point(row[6], row[7], fa.t_vals, s_idx + 1, S, fa.lindisp, fa.t_rand ? fa.t_rand + p + 1 : nullptr);
        } else {
            zz = in_b[p];
            if constexpr (FUSE == 2) { if (s_idx + 1 < S) zn = in_b[p + 1]; }
        }
        px = row[0] + row[3] * zz; py = row[1] + row[4] * zz; pz = row[2] + row[5] * zz;
        vx = row[8]; vy = row[9]; vz = row[10];
        if constexpr (FUSE != 0) {
            constexpr int WPR_ = FUSE == 1 ? 4 : 8, SR_ = FUSE == 1 ? 64 : 128;
            dnorm = dir_norm(row);
            if (fa.noise) nz = fa.noise[p];
            if (FUSE == 1 && wave % WPR_ == 0) {             // this wave draws the fine samples of its ray at the end: lane = uniform
                const int64_t tray = (int64_t)blockIdx.x * (8 / WPR_) + wave / WPR_;
                if (tray * SR_ < P && lane < fa.Nf) tail_u = fa.u_is_row ? fa.u[lane] : fa.u[tray * fa.Nf + lane];
            }
        }
    } else {
        px = in_a[p * 3]; py = in_a[p * 3 + 1]; pz = in_a[p * 3 + 2];
        vx = in_b[p * 3]; vy = in_b[p * 3 + 1]; vz = in_b[p * 3 + 2];
    }
    f32x4 emb[4], edir[2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) emb[t][i] = enc_channel<63>(px, py, pz, 16 * t + 4 * g + i);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) edir[t][i] = enc_channel<27>(vx, vy, vz, 16 * t + 4 * g + i);

    // stash: 16-unit tile `t16` (two per 32-unit row tile) of this wave's 16 points -> rows 16 (t16 & 1) + 4 g + i of the
    // block (row tile t16 >> 1, point tile 4 blockIdx + wave / 2), columns 16 (wave & 1) + n.  The block address is
    // wave-uniform (scalar base), the lane part a constant.
    const int64_t pt_wave = (int64_t)blockIdx.x * 4 + (wave >> 1);
    const int stash_lane = (4 * g) * 32 + 16 * (wave & 1) + n;
    auto stash16 = [&](int t16, const f32x4 &t) {
        if constexpr (STASH) {
            float *q = stash + ((int64_t)(t16 >> 1) * n_pt + pt_wave) * 1024 + (t16 & 1) * 512 + stash_lane;
            q[0] = t[0]; q[32] = t[1]; q[64] = t[2]; q[96] = t[3];
        }
    };
#pragma unroll
    for (int t = 0; t < 4; ++t) stash16(2 * AT_EMB + t, emb[t]);
#pragma unroll
    for (int t = 0; t < 2; ++t) stash16(2 * AT_EDIR + t, edir[t]);

    __syncthreads();                                   // chunks 0, 1 and section B have landed
    const float *sb = lds + RING16_FLOATS;
    f32x4 a = st.read_block<0>();
    f32x4 h[16], o[16];

    // layer 0: 63(+1) -> 256
    layer16<OFF_L0, 16, NTI_L0, false>(st, a, sb + SB_BIAS, [&](auto ti) { return emb[ti.value]; },
        [&](auto to, const f32x4 &acc) { o[to.value] = act16<true>(acc); stash16(2 * AT_H + to.value, o[to.value]); });
#pragma unroll
    for (int t = 0; t < 16; ++t) h[t] = o[t];
    // layers 1..4
    static_for<4>([&](auto li) {
        constexpr int l = 1 + decltype(li)::value;
        layer16<OFF_L1 + (l - 1) * LH_BLOCKS, 16, NTI_LH, false>(st, a, sb + SB_BIAS + l * 256, [&](auto ti) { return h[ti.value]; },
            [&](auto to, const f32x4 &acc) { o[to.value] = act16<true>(acc); stash16(2 * (AT_H + 8 * l) + to.value, o[to.value]); });
#pragma unroll
        for (int t = 0; t < 16; ++t) h[t] = o[t];
    });
    // layer 5: cat[encoded point (64), h4 (256)] -> 256
    layer16<OFF_L5, 16, NTI_L5, false>(st, a, sb + SB_BIAS + 5 * 256,
        [&](auto ti) { if constexpr (ti.value < 4) return emb[ti.value]; else return h[ti.value - 4]; },
        [&](auto to, const f32x4 &acc) { o[to.value] = act16<true>(acc); stash16(2 * (AT_H + 40) + to.value, o[to.value]); });
#pragma unroll
    for (int t = 0; t < 16; ++t) h[t] = o[t];
    // layers 6, 7; sigma = alpha_linear(h7) is accumulated tile by tile in layer 7's epilogue (one weight quad
    // live at a time: reading all 16 up front made the register allocator spill)
    float sigma = 0.f;
    static_for<2>([&](auto li) {
        constexpr int l = 6 + decltype(li)::value;
        layer16<OFF_L6 + (l - 6) * LH_BLOCKS, 16, NTI_LH, false>(st, a, sb + SB_BIAS + l * 256, [&](auto ti) { return h[ti.value]; },
            [&](auto to, const f32x4 &acc) {
                o[to.value] = act16<true>(acc);
                stash16(2 * (AT_H + 8 * l) + to.value, o[to.value]);
                if constexpr (l == 7) {
                    const f32x4 w = *reinterpret_cast<const f32x4 *>(sb + SB_WALPHA + 16 * to.value + 4 * g);
#pragma unroll
                    for (int i = 0; i < 4; ++i) sigma = fmaf(w[i], o[to.value][i], sigma);
                }
            });
#pragma unroll
        for (int t = 0; t < 16; ++t) h[t] = o[t];
    });
    sigma += __shfl_xor(sigma, 16, 64);
    sigma += __shfl_xor(sigma, 32, 64);
    sigma += sb[SB_BALPHA];
    // feature = feature_linear(h7), no activation
    layer16<OFF_FEAT, 16, NTI_LH, false>(st, a, sb + SB_BFEAT, [&](auto ti) { return h[ti.value]; },
        [&](auto to, const f32x4 &acc) { o[to.value] = act16<false>(acc); stash16(2 * AT_FEAT + to.value, o[to.value]); });
    // view branch: cat[feature (256), encoded dir (27+5)] -> 128, relu
    // rgb = rgb_linear(v), accumulated in the view layer's epilogue
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    layer16<OFF_VIEWS, 8, NTI_LV, true>(st, a, sb + SB_BVIEWS,
        [&](auto ti) { if constexpr (ti.value < 16) return o[ti.value]; else return edir[ti.value - 16]; },
        [&](auto to, const f32x4 &acc) {
            const f32x4 v = act16<true>(acc);
            stash16(2 * AT_V + to.value, v);
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(sb + SB_WRGB + 16 * to.value + 4 * g);
            const f32x4 w1 = *reinterpret_cast<const f32x4 *>(sb + SB_WRGB + 128 + 16 * to.value + 4 * g);
            const f32x4 w2 = *reinterpret_cast<const f32x4 *>(sb + SB_WRGB + 256 + 16 * to.value + 4 * g);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                r0 = fmaf(w0[i], v[i], r0);
                r1 = fmaf(w1[i], v[i], r1);
                r2 = fmaf(w2[i], v[i], r2);
            }
        });
    r0 += __shfl_xor(r0, 16, 64); r1 += __shfl_xor(r1, 16, 64); r2 += __shfl_xor(r2, 16, 64);
    r0 += __shfl_xor(r0, 32, 64); r1 += __shfl_xor(r1, 32, 64); r2 += __shfl_xor(r2, 32, 64);
    const float4 out4 = make_float4(r0 + sb[SB_BRGB], r1 + sb[SB_BRGB + 1], r2 + sb[SB_BRGB + 2], sigma);
    if (live && g == 0 && raw) reinterpret_cast<float4 *>(raw)[p] = out4;
    if constexpr (FUSE != 0) {
        // ---- the rest of the pass.  The exponentials of raw2outputs (DS_NeRF/run_nerf_helpers.py:373-393) are evaluated HERE, by
        // all eight waves on their own 16 points: lane group g = 0 forms e = exp(-relu(sigma + noise) dist), groups 1..3 the
        // sigmoid of one colour channel each (every lane holds the point's four raw values after the head reductions).  What
        // remains for the ONE wave per ray that composites while the other seven have left -- its latency is exposed in full --
        // is the transmittance scan and the five sums: round 4's tail evaluated all 8 x 128 exponentials and quotients itself.
        // The terms are those of composite_device.h (comp_*), so every output stays bit-identical to the stand-alone kernels.
        float *terms_s = raw_s, *z_s = raw_s + WG_POINTS * 4;     // {e, c0, c1, c2} per point, the points' depths next to them
        {
            const float rc = g == 1 ? out4.x : (g == 2 ? out4.y : out4.z);
            const float x = g == 0 ? comp_neg_exponent(out4.w + nz, comp_dist(zz, zn, s_idx == S - 1, dnorm)) : -rc;
            const float e = expf(x);
            terms_s[(wave * 16 + n) * 4 + g] = g == 0 ? e : comp_sigmoid_from_exp(e);
            if (g == 0) z_s[wave * 16 + n] = zz;
        }
        __syncthreads();
        constexpr int RAYS = FUSE == 1 ? 2 : 1, WPR = 8 / RAYS, SR = WG_POINTS / RAYS;      // rays, waves and samples per ray
        if (wave % WPR != 0) return;
#ifdef MVIP_EXPERIMENT_NO_FUSE_TAIL        // timing experiment only (results are then missing): the network part alone
        return;
#endif
        const int64_t ray = (int64_t)blockIdx.x * RAYS + wave / WPR;
        if (ray * SR >= P) return;
        constexpr int IT = SR / 64;
        RayState<IT> stt;
        float sums[5];
        ray_forward_terms<IT>(terms_s + (wave / WPR) * SR * 4, z_s + (wave / WPR) * SR, stt, sums);
        composite_store<IT>(stt, sums, ray, SR, fa.flags, fa.rgb, fa.disp, fa.acc, fa.depth, fa.weights, fa.alpha);
        if constexpr (FUSE == 1) {
            // inverse-CDF resampling + merge from the weights in registers: weight e of the pdf = coarse weight e + 1
            float zc[1] = {stt.z[0]}, wts[1], uu[1] = {tail_u};
            const float wn = __shfl_down(stt.w[0], 1, 64);
            wts[0] = lane < SR - 2 ? wn : 0.f;
            sample_merge_ray<1>(zc, wts, uu, ray, SR, fa.Nf, nullptr, fa.z_merged, fa.z_std, nullptr, nullptr);
        }
    }
}

// ---- packing ---------------------------------------------------------------------------------------------------
struct ParamPtrsC16 { const float *p[P_COUNT]; };

__global__ void mlp_pack16_kernel(ParamPtrsC16 pp, float *__restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= SEC_A_FLOATS) return;
    const int blk = idx / BLOCK_FLOATS, r = idx % BLOCK_FLOATS;
    const int lane = r / 4, s = r % 4, m = lane & 15, g = lane >> 4;
    int local, nti, param;
    int layer = -1;                                   // 0..7 hidden, 8 feature, 9 views
    if (blk < OFF_L1) { local = blk; nti = NTI_L0; layer = 0; }
    else if (blk < OFF_L5) { layer = 1 + (blk - OFF_L1) / LH_BLOCKS; local = (blk - OFF_L1) % LH_BLOCKS; nti = NTI_LH; }
    else if (blk < OFF_L6) { layer = 5; local = blk - OFF_L5; nti = NTI_L5; }
    else if (blk < OFF_FEAT) { layer = 6 + (blk - OFF_L6) / LH_BLOCKS; local = (blk - OFF_L6) % LH_BLOCKS; nti = NTI_LH; }
    else if (blk < OFF_VIEWS) { layer = 8; local = blk - OFF_FEAT; nti = NTI_LH; }
    else { layer = 9; local = blk - OFF_VIEWS; nti = NTI_LV; }
    const int to = local / nti, ti = local % nti;
    const int row = 16 * to + m, col = 16 * ti + 4 * g + s;
    float v = 0.f;
    if (layer == 0) { if (col < 63) v = pp.p[P_W0][row * 63 + col]; }
    else if (layer == 5) {
        if (col < 63) v = pp.p[10][row * 319 + col];
        else if (col >= 64) v = pp.p[10][row * 319 + 63 + (col - 64)];
    }
    else if (layer == 8) v = pp.p[P_WF][row * 256 + col];
    else if (layer == 9) { if (col < 283) v = pp.p[P_WV][row * 283 + col]; }
    else { param = 2 * layer; v = pp.p[param][row * 256 + col]; }
    packed[idx] = v;
}

}  // namespace f16p
}  // namespace mvip

using namespace mvip;
using namespace mvip::f16p;

// packed16 = [section A in 16-point block order | section B copied from the 32-point image]
extern "C" int mvip_mlp_pack16(const float *const *params_host, const float *packed32, float *packed16, void *stream) {
    if (!params_host || !packed32 || !packed16) return MVIP_EINVAL;
    ParamPtrsC16 pp;
    for (int i = 0; i < mlp::P_COUNT; ++i) {
        if (!params_host[i]) return MVIP_EINVAL;
        pp.p[i] = params_host[i];
    }
    hipStream_t st = as_stream(stream);
    hipLaunchKernelGGL(mlp_pack16_kernel, dim3((mlp::SEC_A_FLOATS + 255) / 256), dim3(256), 0, st, pp, packed16);
    if (hipMemcpyAsync(packed16 + mlp::SEC_A_FLOATS, packed32 + mlp::SEC_A_FLOATS, mlp::SEC_B_FLOATS * sizeof(float),
                       hipMemcpyDeviceToDevice, st) != hipSuccess)
        return check_launch();
    return check_launch();
}

extern "C" int mvip_mlp_forward_rays16(const float *packed16, const float *rows, const float *z, int64_t B, int S,
                                       float *raw, void *stream) {
    if (B < 0 || S <= 0) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    if (!packed16 || !rows || !z || !raw) return MVIP_EINVAL;
    const int64_t P = B * S;
    const int64_t tiles = (P + WG_POINTS - 1) / WG_POINTS;
    hipLaunchKernelGGL((mlp_forward16_kernel<true>), dim3((unsigned)tiles), dim3(512), 0,
                       as_stream(stream), packed16, rows, z, P, S, raw);
    return check_launch();
}

// Training forward on the two-wave kernel: raw AND the activation stash of mvip_mlp_stash_floats(B*S) floats that
// mvip_mlp_backward_stash consumes (same layout as mvip_mlp_forward_rays_stash writes; precision 0 only).
extern "C" int mvip_mlp_forward_rays_stash16(const float *packed16, const float *rows, const float *z, int64_t B, int S,
                                             float *raw, float *stash, void *stream) {
    if (B < 0 || S <= 0) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    if (!packed16 || !rows || !z || !raw || !stash) return MVIP_EINVAL;
    const int64_t P = B * S;
    const int64_t wgs = (P + WG_POINTS - 1) / WG_POINTS;
    hipLaunchKernelGGL((mlp_forward16_kernel<true, true>), dim3((unsigned)wgs), dim3(512), 0, as_stream(stream), packed16,
                       rows, z, P, S, raw, stash, wgs * 4);
    return check_launch();
}

extern "C" int mvip_mlp_forward_points16(const float *packed16, const float *pts, const float *dirs, int64_t P,
                                         float *raw, void *stream) {
    if (P < 0) return MVIP_EINVAL;
    if (P == 0) return MVIP_OK;
    if (!packed16 || !pts || !dirs || !raw) return MVIP_EINVAL;
    hipLaunchKernelGGL((mlp_forward16_kernel<false>), dim3((unsigned)((P + WG_POINTS - 1) / WG_POINTS)), dim3(512), 0,
                       as_stream(stream), packed16, pts, dirs, P, 1, raw);
    return check_launch();
}

// ---- render_rays in two launches (no-grad renders of the native 8x256 networks; DS_NeRF/run.py:1703-1847) --------------
// Coarse pass, 64 samples per ray: stratified depths (t_vals [64], t_rand [B,64] or NULL) -> network -> raw2outputs (noise
// [B,64] or NULL, flags as mvip_composite_forward) -> inverse-CDF resampling with Nf <= 64 uniforms (u [B,Nf], or one row
// when u_is_row) -> merged depths.  Outputs: rgb0 [B,3], disp0 [B], acc0 [B], z_merged [B,64+Nf], z_std [B]; alpha0 [B,64],
// depth0 [B], weights0 [B,64] optional (NULL = not wanted).  Every value is bit-identical to the chain
// mvip_stratified_z -> mvip_mlp_forward_rays16 -> mvip_composite_forward -> mvip_sample_pdf_merge.
extern "C" int mvip_render_coarse_fused(const float *packed16, const float *rows, int64_t B, const float *t_vals, int lindisp,
                                        const float *t_rand, const float *noise, const float *u, int u_is_row, int Nf, int flags,
                                        float *rgb0, float *disp0, float *acc0, float *depth0, float *weights0, float *alpha0,
                                        float *z_merged, float *z_std, void *stream) {
    if (B < 0 || Nf < 1 || Nf > 64) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    if (!packed16 || !rows || !t_vals || !u || !rgb0 || !disp0 || !acc0 || !z_merged || !z_std) return MVIP_EINVAL;
    FuseArgs fa;
    fa.t_vals = t_vals; fa.t_rand = t_rand; fa.noise = noise; fa.u = u; fa.u_is_row = u_is_row; fa.lindisp = lindisp;
    fa.flags = flags; fa.Nf = Nf;
    fa.rgb = rgb0; fa.disp = disp0; fa.acc = acc0; fa.depth = depth0; fa.weights = weights0; fa.alpha = alpha0;
    fa.z_merged = z_merged; fa.z_std = z_std;
    const int64_t P = B * 64;
    hipLaunchKernelGGL((mlp_forward16_kernel<true, false, 1>), dim3((unsigned)((P + WG_POINTS - 1) / WG_POINTS)), dim3(512), 0,
                       as_stream(stream), packed16, rows, (const float *)nullptr, P, 64, (float *)nullptr, (float *)nullptr,
                       (int64_t)0, fa);
    return check_launch();
}

// Fine pass, 128 samples per ray at the depths z [B,128]: network -> raw2outputs.  Outputs as mvip_composite_forward
// (weights required, alpha optional) plus raw [B,128,4] (optional); bit-identical to mvip_mlp_forward_rays16 ->
// mvip_composite_forward.
extern "C" int mvip_render_fine_fused(const float *packed16, const float *rows, const float *z, int64_t B, const float *noise,
                                      int flags, float *raw, float *rgb, float *disp, float *acc, float *depth, float *weights,
                                      float *alpha, void *stream) {
    if (B < 0) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    if (!packed16 || !rows || !z || !rgb || !disp || !acc || !depth || !weights) return MVIP_EINVAL;
    FuseArgs fa;
    fa.t_vals = nullptr; fa.t_rand = nullptr; fa.noise = noise; fa.u = nullptr; fa.u_is_row = 0; fa.lindisp = 0;
    fa.flags = flags; fa.Nf = 0;
    fa.rgb = rgb; fa.disp = disp; fa.acc = acc; fa.depth = depth; fa.weights = weights; fa.alpha = alpha;
    fa.z_merged = nullptr; fa.z_std = nullptr;
    const int64_t P = B * 128;
    hipLaunchKernelGGL((mlp_forward16_kernel<true, false, 2>), dim3((unsigned)(P / WG_POINTS)), dim3(512), 0, as_stream(stream),
                       packed16, rows, z, P, 128, raw, (float *)nullptr, (int64_t)0, fa);
    return check_launch();
}
