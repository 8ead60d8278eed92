// Split-precision fused NeRF MLP forward (precision = 1, "f16x3").
//
// Same algorithm, geometry and register-resident activation scheme as mlp_fwd.hip, but every
// contraction runs on v_mfma_f32_32x32x16_f16 (16x the fp32 MFMA rate) with BOTH operands split
// into two fp16 terms:   W = Wh + Wl,  X = Xh + Xl,   Wh = f16(W), Wl = f16(W - Wh)  (likewise X)
//        W.X  ~=  Wh.Xh + Wh.Xl + Wl.Xh          (three MFMAs; the dropped Wl.Xl term is ~2^-22 |W.X|)
// Each fp16 x fp16 product is exact in the fp32 accumulator, the split residuals are exact
// (x - f16(x) has <= 13 significant bits), so the only deviations from the fp32 kernel are the
// 2^-22 representation error of the lo terms and the dropped lo.lo products: relative error
// ~2-4e-7 per contraction, the same order as fp32 rounding itself (measured in tests/).
// Encodings, biases, ReLU, the sigma/rgb heads and all accumulation stay fp32.
//
// A 32x32 fp32 accumulator tile is turned into the B operands of the next layer in registers: for
// k-step s (K = 16), element j of lane (col, h) is accumulator register 8s+j = tile row
// 16s + 8(j>>2) + 4h + (j&3); the packed weight image stores the A fragments in that k order.
// Weight stream: [Ah | Al] 1-KB blocks per k-step, same byte size and layer offsets as the fp32
// image, staged by LDS-DMA through a 2 x 64 KB ring (one barrier per 64 KB = 96 MFMAs per wave).
#include "common.h"
#include "mlp_layout.h"
#include "mlp_device.h"
#include "mlp_device_f16.h"

namespace mvip {
using namespace mlp;

// ---------------------------------------------------------------------------------------------- pack
struct ParamPtrsC { const float *p[P_COUNT]; };

__device__ __forceinline__ float weight_at(const ParamPtrsC &pp, int layer_blk_off, int row, int k) {
    // (layer identified by its block offset) -> W[row][k] with the fp32 image's K padding rules
    if (layer_blk_off == OFF_L0) return k < 63 ? pp.p[P_W0][row * 63 + k] : 0.f;
    if (layer_blk_off == OFF_L5) {
        if (k < 63) return pp.p[10][row * 319 + k];
        if (k >= 64) return pp.p[10][row * 319 + 63 + (k - 64)];
        return 0.f;
    }
    if (layer_blk_off == OFF_VIEWS) return k < 283 ? pp.p[P_WV][row * 283 + k] : 0.f;
    if (layer_blk_off == OFF_FEAT) return pp.p[P_WF][row * 256 + k];
    if (layer_blk_off >= OFF_L6) return pp.p[2 * (6 + (layer_blk_off - OFF_L6) / LH_BLOCKS)][row * 256 + k];
    return pp.p[2 * (1 + (layer_blk_off - OFF_L1) / LH_BLOCKS)][row * 256 + k];
}

__global__ void mlp_pack_f16x3_kernel(ParamPtrsC pp, _Float16 *__restrict__ img) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // one fp16 element of section A
    if (idx >= SEC_A_FLOATS * 2) return;
    const int blk = idx / 512, e = idx % 512;                        // 512 halves per 1-KB block
    const int lane = e / 8, j = e % 8, h = lane / 32, i = lane % 32;
    int off, ksn;                                                     // layer block offset, k-steps per tile
    if (blk < OFF_L1) { off = OFF_L0; ksn = L0_KG / 2; }
    else if (blk < OFF_L5) { off = OFF_L1 + ((blk - OFF_L1) / LH_BLOCKS) * LH_BLOCKS; ksn = LH_KG / 2; }
    else if (blk < OFF_L6) { off = OFF_L5; ksn = L5_KG / 2; }
    else if (blk < OFF_FEAT) { off = OFF_L6 + ((blk - OFF_L6) / LH_BLOCKS) * LH_BLOCKS; ksn = LH_KG / 2; }
    else if (blk < OFF_VIEWS) { off = OFF_FEAT; ksn = LH_KG / 2; }
    else { off = OFF_VIEWS; ksn = LV_KG / 2; }
    const int local = blk - off, pair = local / 2, lo = local % 2;   // [Ah | Al] per k-step
    const int ti = pair / ksn, ks = pair % ksn;
    const int k = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);
    const float w = weight_at(pp, off, 32 * ti + i, k);
    const _Float16 wh = (_Float16)w;
    img[idx] = lo ? (_Float16)(w - (float)wh) : wh;
}

template <bool FROM_RAYS, bool STASH>
__global__ __launch_bounds__(256, 1) void mlp_forward_f16x3_kernel(
    const float *__restrict__ img, const float *__restrict__ in_a, const float *__restrict__ in_b, int64_t p_begin,
    int64_t p_count, int S, float *__restrict__ raw, float *__restrict__ stash, int64_t n_pt) {
    __shared__ __attribute__((aligned(16))) float lds[F_LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hh = lane >> 5;
    const int64_t pt = (int64_t)blockIdx.x * 4 + wave;
    int64_t pl = pt * 32 + j;
    const bool live = pl < p_count;
    if (!live) pl = p_count - 1;
    const int64_t p = p_begin + pl;
    auto stash_tile = [&](int row_tile, const f32x16 &t) {          // fp32 activations, same stash as precision 0
        if constexpr (STASH) {
            store_tile<true>(stash_block(stash, row_tile, n_pt, pt), t, j, hh);
            // + the ReLU sign mask of the trunk / view-branch tiles: all the delta kernel reads of them (mlp_device.h)
            if (row_tile < AT_FEAT) *mask_slot(stash, row_tile, n_pt, pt, lane) = (unsigned short)tile_sign_mask(t);
            else if (row_tile >= AT_V && row_tile < AT_V + 4)
                *mask_slot(stash, 64 + row_tile - AT_V, n_pt, pt, lane) = (unsigned short)tile_sign_mask(t);
        }
    };

    StreamF st{img, lds, wave, lane, TOTAL_BLOCKS};
    st.init_bases();
    for (int b = wave; b < SEC_B_FLOATS / BLOCK_FLOATS; b += 4)        // section B (fp32 small vectors)
        glds16(img + SEC_A_FLOATS + b * BLOCK_FLOATS + lane * 4, lds + F_RING_FLOATS + b * BLOCK_FLOATS);
    st.issue_group(0);

    float px, py, pz, vx, vy, vz;
    load_point<FROM_RAYS>(in_a, in_b, p, S, px, py, pz, vx, vy, vz);
    Frag emb[2];
    {
        f32x16 t;
        encode_tile<63>(px, py, pz, hh, 0, t); emb[0] = split_tile(t); stash_tile(AT_EMB, t);
        encode_tile<63>(px, py, pz, hh, 1, t); emb[1] = split_tile(t); stash_tile(AT_EMB + 1, t);
        if constexpr (STASH) { encode_tile<27>(vx, vy, vz, hh, 0, t); stash_tile(AT_EDIR, t); }
    }
    __syncthreads();
    const float *sb = lds + F_RING_FLOATS;
    APair a0{st.read_block<0>(), st.read_block<1>()}, a1{st.read_block<2>(), st.read_block<3>()};

    Frag h[8], o[8];
    auto from = [](const Frag *x, int t, int s) { return FragPair{x[t].hi[s], x[t].lo[s]}; };

    // layer 0
    run_layer_f<OFF_L0, L0_NT, L0_KG / 2, false>(st, a0, a1,
        [&](auto ks) { return from(emb, ks.value >> 1, ks.value & 1); },
        NoPre{}, [&](auto ti, const f32x16 &acc, int) {
            const f32x16 a = act_only<true>(acc);
            stash_tile(AT_H + ti.value, a);
            o[ti.value] = split_tile(a);
        }, sb + SB_BIAS);
#pragma unroll
    for (int t = 0; t < 8; ++t) h[t] = o[t];
    // layers 1..4
    static_for<4>([&](auto li) {
        constexpr int l = 1 + decltype(li)::value;
        run_layer_f<OFF_L1 + (l - 1) * LH_BLOCKS, LH_NT, LH_KG / 2, false>(st, a0, a1,
            [&](auto ks) { return from(h, ks.value >> 1, ks.value & 1); },
            NoPre{}, [&](auto ti, const f32x16 &acc, int) {
                const f32x16 a = act_only<true>(acc);
                stash_tile(AT_H + 8 * l + ti.value, a);
                o[ti.value] = split_tile(a);
            }, sb + SB_BIAS + l * 256);
#pragma unroll
        for (int t = 0; t < 8; ++t) h[t] = o[t];
    });
    // layer 5: cat[encoded point, h4]
    run_layer_f<OFF_L5, L5_NT, L5_KG / 2, false>(st, a0, a1,
        [&](auto ks) {
            if constexpr (ks.value < 4) return from(emb, ks.value >> 1, ks.value & 1);
            else return from(h, (ks.value - 4) >> 1, (ks.value - 4) & 1);
        },
        NoPre{}, [&](auto ti, const f32x16 &acc, int) {
            const f32x16 a = act_only<true>(acc);
            stash_tile(AT_H + 40 + ti.value, a);
            o[ti.value] = split_tile(a);
        }, sb + SB_BIAS + 5 * 256);
#pragma unroll
    for (int t = 0; t < 8; ++t) h[t] = o[t];
    // layer 6
    run_layer_f<OFF_L6, LH_NT, LH_KG / 2, false>(st, a0, a1,
        [&](auto ks) { return from(h, ks.value >> 1, ks.value & 1); },
        NoPre{}, [&](auto ti, const f32x16 &acc, int) {
            const f32x16 a = act_only<true>(acc);
            stash_tile(AT_H + 48 + ti.value, a);
            o[ti.value] = split_tile(a);
        }, sb + SB_BIAS + 6 * 256);
#pragma unroll
    for (int t = 0; t < 8; ++t) h[t] = o[t];
    // layer 7 (+ the fp32 sigma head accumulated from the fp32 activations in the epilogue)
    float sigma = 0.f;
    run_layer_f<OFF_L6 + LH_BLOCKS, LH_NT, LH_KG / 2, false>(st, a0, a1,
        [&](auto ks) { return from(h, ks.value >> 1, ks.value & 1); },
        NoPre{}, [&](auto ti, const f32x16 &acc, int) {
            const f32x16 a7 = act_only<true>(acc);
            stash_tile(AT_H + 56 + ti.value, a7);
            sigma += dot_tiles<1>(&a7, sb + SB_WALPHA + 32 * ti.value, hh);
            o[ti.value] = split_tile(a7);
        }, sb + SB_BIAS + 7 * 256);
#pragma unroll
    for (int t = 0; t < 8; ++t) h[t] = o[t];
    sigma += __shfl_xor(sigma, 32, 64);
    sigma += sb[SB_BALPHA];
    // feature (no activation)
    run_layer_f<OFF_FEAT, LH_NT, LH_KG / 2, false>(st, a0, a1,
        [&](auto ks) { return from(h, ks.value >> 1, ks.value & 1); },
        NoPre{}, [&](auto ti, const f32x16 &acc, int) {
            const f32x16 a = act_only<false>(acc);
            stash_tile(AT_FEAT + ti.value, a);
            o[ti.value] = split_tile(a);
        }, sb + SB_BFEAT);
    // view branch (+ the fp32 rgb head); the direction encoding is formed only now (16 fewer live
    // registers through the trunk)
    Frag edir;
    {
        f32x16 t;
        encode_tile<27>(vx, vy, vz, hh, 0, t);
        edir = split_tile(t);
    }
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    run_layer_f<OFF_VIEWS, LV_NT, LV_KG / 2, true>(st, a0, a1,
        [&](auto ks) {
            if constexpr (ks.value < 16) return from(o, ks.value >> 1, ks.value & 1);
            else return FragPair{edir.hi[ks.value - 16], edir.lo[ks.value - 16]};
        },
        NoPre{}, [&](auto ti, const f32x16 &acc, int) {
            const f32x16 v = act_only<true>(acc);
            stash_tile(AT_V + ti.value, v);
            r0 += dot_tiles<1>(&v, sb + SB_WRGB + 32 * ti.value, hh);
            r1 += dot_tiles<1>(&v, sb + SB_WRGB + 128 + 32 * ti.value, hh);
            r2 += dot_tiles<1>(&v, sb + SB_WRGB + 256 + 32 * ti.value, hh);
        }, sb + SB_BVIEWS);
    r0 += __shfl_xor(r0, 32, 64);
    r1 += __shfl_xor(r1, 32, 64);
    r2 += __shfl_xor(r2, 32, 64);
    if (raw && live && hh == 0)
        reinterpret_cast<float4 *>(raw)[p] = make_float4(r0 + sb[SB_BRGB], r1 + sb[SB_BRGB + 1], r2 + sb[SB_BRGB + 2], sigma);
}

int mlp_forward_f16x3_launch(const float *img, const float *a, const float *b, int64_t p_begin, int64_t p_count,
                             int S, float *raw, float *stash, int64_t n_pt, bool from_rays, void *stream) {
    if (p_count == 0) return MVIP_OK;
    const dim3 grid((unsigned)((p_count + 127) / 128)), block(256);
    hipStream_t s = as_stream(stream);
    if (from_rays) {
        if (stash) hipLaunchKernelGGL((mlp_forward_f16x3_kernel<true, true>), grid, block, 0, s, img, a, b, p_begin, p_count, S, raw, stash, n_pt);
        else hipLaunchKernelGGL((mlp_forward_f16x3_kernel<true, false>), grid, block, 0, s, img, a, b, p_begin, p_count, S, raw, stash, n_pt);
    } else {
        if (stash) hipLaunchKernelGGL((mlp_forward_f16x3_kernel<false, true>), grid, block, 0, s, img, a, b, p_begin, p_count, S, raw, stash, n_pt);
        else hipLaunchKernelGGL((mlp_forward_f16x3_kernel<false, false>), grid, block, 0, s, img, a, b, p_begin, p_count, S, raw, stash, n_pt);
    }
    return check_launch();
}

static int launch_f16x3(const float *img, const float *a, const float *b, int64_t P, int S, float *raw, bool from_rays,
                        void *stream) {
    return mlp_forward_f16x3_launch(img, a, b, 0, P, S, raw, nullptr, 0, from_rays, stream);
}

}  // namespace mvip

using namespace mvip;

extern "C" int mvip_mlp_pack_f16x3(const float *const *params_host, float *image, const float *packed_f32,
                                   void *stream) {
    if (!params_host || !image || !packed_f32) return MVIP_EINVAL;
    ParamPtrsC pp;
    for (int i = 0; i < mlp::P_COUNT; ++i) {
        if (!params_host[i]) return MVIP_EINVAL;
        pp.p[i] = params_host[i];
    }
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(mlp_pack_f16x3_kernel, dim3((mlp::SEC_A_FLOATS * 2 + 255) / 256), dim3(256), 0, s, pp,
                       reinterpret_cast<_Float16 *>(image));
    // section B (fp32 biases / head rows) is shared with the fp32 image
    hipMemcpyAsync(image + mlp::SEC_A_FLOATS, packed_f32 + mlp::SEC_A_FLOATS, sizeof(float) * mlp::SEC_B_FLOATS,
                   hipMemcpyDeviceToDevice, s);
    return check_launch();
}

extern "C" int mvip_mlp_forward_rays_f16x3(const float *image, const float *rows, const float *z, int64_t B, int S,
                                           float *raw, void *stream) {
    if (B < 0 || S <= 0) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    if (!image || !rows || !z || !raw) return MVIP_EINVAL;
    return launch_f16x3(image, rows, z, B * S, S, raw, true, stream);
}

extern "C" int mvip_mlp_forward_points_f16x3(const float *image, const float *pts, const float *dirs, int64_t P,
                                             float *raw, void *stream) {
    if (P < 0) return MVIP_EINVAL;
    if (P == 0) return MVIP_OK;
    if (!image || !pts || !dirs || !raw) return MVIP_EINVAL;
    return launch_f16x3(image, pts, dirs, P, 1, raw, false, stream);
}
