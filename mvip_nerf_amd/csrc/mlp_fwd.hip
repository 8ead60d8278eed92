// Fused NeRF MLP forward for gfx950: ray point -> sinusoidal encoding -> 8x256 MLP (+skip) ->
// sigma / feature / view branch -> raw[4], one kernel, activations never leave registers.
// Replaces run_network + Embedder.embed + NeRF.forward
// (DS_NeRF/run.py:1108-1124, DS_NeRF/run_nerf_helpers.py:22-52, :104-127).
//
// Geometry.  256 threads = 4 wavefronts, one per SIMD (the kernel uses ~330 of the 512 unified
// VGPR/AGPRs).  Each wave owns 32 points for the whole network: its activations are a
// [256 units x 32 points] matrix held as 8 accumulator tiles (128 registers), and because a
// 32x32 MFMA accumulator register is directly a valid K=2 B operand (see mlp_layout.h) the
// output of one layer feeds the next layer's v_mfma_f32_32x32x2_f32 with no LDS round trip, no
// shuffles and no HBM traffic.  Arithmetic is exact fp32 (k-ordered fma chains), i.e. the same
// precision class as the reference's fp32 nn.Linear.
//
// Weights.  The 2.3 MB packed image is streamed through a 4-slot x 16 KB LDS ring by
// global_load_lds_dwordx4 (LDS-DMA), two chunks ahead of use, one __syncthreads per chunk
// (64 MFMAs = 4096 cycles per wave); all four waves consume the same A operands with one
// conflict-free ds_read_b128 per four MFMAs.  Roofline: FLOP-bound -- 2*606,208 padded MACs per
// point against ~20 B/point of HBM traffic; L2->LDS traffic is 2.3 MB per 128 points.
#include "common.h"
#include "mlp_layout.h"
#include "mlp_device.h"

namespace mvip {
using namespace mlp;

template <bool FROM_RAYS, bool STASH>
__global__ __launch_bounds__(256, 1) void mlp_forward_kernel(
    const float *__restrict__ packed, const float *__restrict__ in_a, const float *__restrict__ in_b,
    int64_t p_begin, int64_t p_count, int S, float *__restrict__ raw, float *__restrict__ stash, int64_t n_pt,
    unsigned long long *__restrict__ clock_dbg) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    // diagnostics only (clock_dbg == nullptr in every product call): shader-clock and 100 MHz
    // real-time stamps around the whole workgroup -> sustained clock and cycles per workgroup
    unsigned long long t0c = 0, t0r = 0, ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (clock_dbg) { t0c = __builtin_amdgcn_s_memtime(); t0r = __builtin_amdgcn_s_memrealtime(); }
#define MVIP_STAMP(k) do { if (clock_dbg) ts[k] = __builtin_amdgcn_s_memtime() - t0c; } while (0)
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hh = lane >> 5;
    const int64_t pt = (int64_t)blockIdx.x * 4 + wave;           // 32-point tile of this wave
    int64_t pl = pt * 32 + j;                                     // index inside [0, p_count)
    const bool live = pl < p_count;
    if (!live) pl = p_count - 1;
    const int64_t p = p_begin + pl;

    Stream st{packed, lds, wave, lane};
    st.prologue(packed + SEC_A_FLOATS);

    // ---- inputs: point and unit view direction of this lane's column ----
    float px, py, pz, vx, vy, vz;
    load_point<FROM_RAYS>(in_a, in_b, p, S, px, py, pz, vx, vy, vz);

    f32x16 emb[2], edir;
    encode_tile<63>(px, py, pz, hh, 0, emb[0]);
    encode_tile<63>(px, py, pz, hh, 1, emb[1]);
    encode_tile<27>(vx, vy, vz, hh, 0, edir);
    auto stash_tile = [&](int row_tile, const f32x16 &t) {
        if constexpr (STASH) store_tile(stash_block(stash, row_tile, n_pt, pt), t, j, hh);
    };
    stash_tile(AT_EMB, emb[0]);
    stash_tile(AT_EMB + 1, emb[1]);
    stash_tile(AT_EDIR, edir);

    MVIP_STAMP(0);                                    // inputs loaded + encoded
    __syncthreads();                                  // chunks 0,1 and section B have landed
    const float *sb = lds + RING_FLOATS;
    APair32 a = st.first_pair();
    MVIP_STAMP(1);                                    // weight ring primed

    f32x16 h[8], o[8];

    // ---- layer 0: 63(+1) -> 256 ----
    run_layer<L0_NT, L0_KG, false>(st, 0, a,
        [&](auto kg, auto s) { return emb[kg.value >> 2][4 * (kg.value & 3) + s.value]; },
        NoPre{}, [&](auto ti, const f32x16 &acc, int) {
            o[ti.value] = bias_relu<true>(acc, sb + SB_BIAS + 32 * ti.value, hh);
            stash_tile(AT_H + ti.value, o[ti.value]);
        });
#pragma unroll
    for (int t = 0; t < 8; ++t) h[t] = o[t];

    MVIP_STAMP(2);                                    // layer 0 done (256 MFMAs)
    // ---- layers 1..4: 256 -> 256 (unrolled: inside a runtime loop the 256 loop-carried h/o
    //      registers cost ~5.5k cycles of shuffling per layer, measured with tools/clock_probe.py) ----
    static_for<4>([&](auto li) {
        constexpr int l = 1 + decltype(li)::value;
        run_layer<LH_NT, LH_KG, false>(st, OFF_L1 / CHUNK_BLOCKS + (l - 1) * (LH_BLOCKS / CHUNK_BLOCKS), a,
            [&](auto kg, auto s) { return h[kg.value >> 2][4 * (kg.value & 3) + s.value]; },
            NoPre{}, [&](auto ti, const f32x16 &acc, int) {
                o[ti.value] = bias_relu<true>(acc, sb + SB_BIAS + l * 256 + 32 * ti.value, hh);
                stash_tile(AT_H + 8 * l + ti.value, o[ti.value]);
            });
#pragma unroll
        for (int t = 0; t < 8; ++t) h[t] = o[t];
    });

    MVIP_STAMP(3);                                    // layers 1..4 done (4096 MFMAs)
    // ---- layer 5: cat[encoded point (64), h4 (256)] -> 256 ----
    run_layer<L5_NT, L5_KG, false>(st, OFF_L5 / CHUNK_BLOCKS, a,
        [&](auto kg, auto s) {
            if constexpr (kg.value < 8) return emb[kg.value >> 2][4 * (kg.value & 3) + s.value];
            else return h[(kg.value - 8) >> 2][4 * ((kg.value - 8) & 3) + s.value];
        },
        NoPre{}, [&](auto ti, const f32x16 &acc, int) {
            o[ti.value] = bias_relu<true>(acc, sb + SB_BIAS + 5 * 256 + 32 * ti.value, hh);
            stash_tile(AT_H + 40 + ti.value, o[ti.value]);
        });
#pragma unroll
    for (int t = 0; t < 8; ++t) h[t] = o[t];

    MVIP_STAMP(4);                                    // layer 5 done (1280 MFMAs)
    // ---- layers 6, 7 ----
    static_for<2>([&](auto li) {
        constexpr int l = 6 + decltype(li)::value;
        run_layer<LH_NT, LH_KG, false>(st, OFF_L6 / CHUNK_BLOCKS + (l - 6) * (LH_BLOCKS / CHUNK_BLOCKS), a,
            [&](auto kg, auto s) { return h[kg.value >> 2][4 * (kg.value & 3) + s.value]; },
            NoPre{}, [&](auto ti, const f32x16 &acc, int) {
                o[ti.value] = bias_relu<true>(acc, sb + SB_BIAS + l * 256 + 32 * ti.value, hh);
                stash_tile(AT_H + 8 * l + ti.value, o[ti.value]);
            });
#pragma unroll
        for (int t = 0; t < 8; ++t) h[t] = o[t];
    });

    MVIP_STAMP(5);                                    // layers 6, 7 done (2048 MFMAs)
    // ---- sigma = alpha_linear(h7): a 256-long dot product per point, on the VALU ----
    float sigma = dot_tiles<8>(h, sb + SB_WALPHA, hh);
    sigma += __shfl_xor(sigma, 32, 64);
    sigma += sb[SB_BALPHA];

    // ---- feature = feature_linear(h7) (no activation) ----
    run_layer<LH_NT, LH_KG, false>(st, OFF_FEAT / CHUNK_BLOCKS, a,
        [&](auto kg, auto s) { return h[kg.value >> 2][4 * (kg.value & 3) + s.value]; },
        NoPre{}, [&](auto ti, const f32x16 &acc, int) {
            o[ti.value] = bias_relu<false>(acc, sb + SB_BFEAT + 32 * ti.value, hh);
            stash_tile(AT_FEAT + ti.value, o[ti.value]);
        });

    MVIP_STAMP(6);                                    // sigma + feature layer done (1024 MFMAs)
    // ---- view branch: cat[feature (256), encoded dir (27+5)] -> 128, relu ----
    f32x16 v[4];
    run_layer<LV_NT, LV_KG, true>(st, OFF_VIEWS / CHUNK_BLOCKS, a,
        [&](auto kg, auto s) {
            if constexpr (kg.value < 32) return o[kg.value >> 2][4 * (kg.value & 3) + s.value];
            else return edir[4 * (kg.value - 32) + s.value];
        },
        NoPre{}, [&](auto ti, const f32x16 &acc, int) {
            v[ti.value] = bias_relu<true>(acc, sb + SB_BVIEWS + 32 * ti.value, hh);
            stash_tile(AT_V + ti.value, v[ti.value]);
        });

    MVIP_STAMP(7);                                    // view branch done (576 MFMAs)
    // ---- rgb = rgb_linear(v): three 128-long dot products ----
    float r0 = dot_tiles<4>(v, sb + SB_WRGB, hh);
    float r1 = dot_tiles<4>(v, sb + SB_WRGB + 128, hh);
    float r2 = dot_tiles<4>(v, sb + SB_WRGB + 256, hh);
    r0 += __shfl_xor(r0, 32, 64);
    r1 += __shfl_xor(r1, 32, 64);
    r2 += __shfl_xor(r2, 32, 64);
    if (raw && live && hh == 0) {
        float4 out = make_float4(r0 + sb[SB_BRGB], r1 + sb[SB_BRGB + 1], r2 + sb[SB_BRGB + 2], sigma);
        reinterpret_cast<float4 *>(raw)[p] = out;
    }
    if (clock_dbg && threadIdx.x == 0) {
        unsigned long long *o = clock_dbg + 10 * (int64_t)blockIdx.x;
        o[0] = __builtin_amdgcn_s_memtime() - t0c;
        o[1] = __builtin_amdgcn_s_memrealtime() - t0r;
#pragma unroll
        for (int k = 0; k < 8; ++k) o[2 + k] = ts[k];
    }
}

// p_begin/p_count select a sub-range of the P points (used by the tiled backward); `stash`
// (n_pt = p_count/32 rounded up to a multiple of 4 point tiles) receives the activations.
int mlp_forward_launch(const float *packed, const float *a, const float *b, int64_t p_begin, int64_t p_count,
                       int S, float *raw, float *stash, int64_t n_pt, bool from_rays, void *stream,
                       unsigned long long *clock_dbg = nullptr) {
    if (p_count == 0) return MVIP_OK;
    const dim3 grid((unsigned)((p_count + 127) / 128)), block(256);
    hipStream_t s = as_stream(stream);
    if (from_rays) {
        if (stash) hipLaunchKernelGGL((mlp_forward_kernel<true, true>), grid, block, 0, s, packed, a, b, p_begin, p_count, S, raw, stash, n_pt, clock_dbg);
        else hipLaunchKernelGGL((mlp_forward_kernel<true, false>), grid, block, 0, s, packed, a, b, p_begin, p_count, S, raw, stash, n_pt, clock_dbg);
    } else {
        if (stash) hipLaunchKernelGGL((mlp_forward_kernel<false, true>), grid, block, 0, s, packed, a, b, p_begin, p_count, S, raw, stash, n_pt, clock_dbg);
        else hipLaunchKernelGGL((mlp_forward_kernel<false, false>), grid, block, 0, s, packed, a, b, p_begin, p_count, S, raw, stash, n_pt, clock_dbg);
    }
    return check_launch();
}

}  // namespace mvip

using namespace mvip;

extern "C" int mvip_mlp_forward_rays_f16x3(const float *, const float *, const float *, int64_t, int, float *, void *);
extern "C" int mvip_mlp_forward_points_f16x3(const float *, const float *, const float *, int64_t, float *, void *);

extern "C" int mvip_mlp_forward_rays(const float *packed, const float *rows, const float *z, int64_t B, int S,
                                     float *raw, int precision, void *stream) {
    if (B < 0 || S <= 0) return MVIP_EINVAL;
    if (precision == 1) return mvip_mlp_forward_rays_f16x3(packed, rows, z, B, S, raw, stream);
    if (precision != 0) return MVIP_EUNSUP;
    if (B == 0) return MVIP_OK;
    if (!packed || !rows || !z || !raw) return MVIP_EINVAL;
    return mlp_forward_launch(packed, rows, z, 0, B * S, S, raw, nullptr, 0, true, stream);
}

extern "C" int mvip_mlp_forward_points(const float *packed, const float *pts, const float *dirs, int64_t P,
                                       float *raw, int precision, void *stream) {
    if (P < 0) return MVIP_EINVAL;
    if (precision == 1) return mvip_mlp_forward_points_f16x3(packed, pts, dirs, P, raw, stream);
    if (precision != 0) return MVIP_EUNSUP;
    if (P == 0) return MVIP_OK;
    if (!packed || !pts || !dirs || !raw) return MVIP_EINVAL;
    return mlp_forward_launch(packed, pts, dirs, 0, P, 1, raw, nullptr, 0, false, stream);
}

/* diagnostics: as mvip_mlp_forward_rays, additionally writing per-workgroup (shader cycles, 100 MHz ticks) */
extern "C" int mvip_debug_forward_clock(const float *packed, const float *rows, const float *z, int64_t B, int S,
                                        float *raw, unsigned long long *clock_out, void *stream) {
    if (B <= 0 || S <= 0 || !packed || !rows || !z || !raw || !clock_out) return MVIP_EINVAL;
    return mlp_forward_launch(packed, rows, z, 0, B * S, S, raw, nullptr, 0, true, stream, clock_out);
}
