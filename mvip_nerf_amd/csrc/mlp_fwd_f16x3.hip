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

namespace mvip {
using namespace mlp;

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int F_GROUP_BLOCKS = 64;                         // blocks (1 KB) per barrier group
constexpr int F_RING_FLOATS = 2 * F_GROUP_BLOCKS * BLOCK_FLOATS;   // 128 KB
constexpr int F_LDS_FLOATS = F_RING_FLOATS + SEC_B_FLOATS;
constexpr int F_TOTAL_GROUPS = (TOTAL_BLOCKS + F_GROUP_BLOCKS - 1) / F_GROUP_BLOCKS;
static_assert(OFF_L1 % F_GROUP_BLOCKS == 0 && OFF_L5 % F_GROUP_BLOCKS == 0 && OFF_L6 % F_GROUP_BLOCKS == 0 &&
              OFF_FEAT % F_GROUP_BLOCKS == 0 && OFF_VIEWS % F_GROUP_BLOCKS == 0 && LH_BLOCKS % F_GROUP_BLOCKS == 0,
              "layers must start on barrier-group boundaries");

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

// ---------------------------------------------------------------------------------------------- kernel
struct StreamF {
    const float *img;       // f16x3 image viewed as floats (same byte layout granularity: 256 floats = 1 KB)
    float *lds;
    int wave, lane;
    // issue the 64 blocks of barrier group g into ring half (g & 1): 16 blocks per wave
    __device__ __forceinline__ void issue_group(int g) const {
        if (g < F_TOTAL_GROUPS) {
            const int nblk = (TOTAL_BLOCKS - g * F_GROUP_BLOCKS) < F_GROUP_BLOCKS ? (TOTAL_BLOCKS - g * F_GROUP_BLOCKS)
                                                                                 : F_GROUP_BLOCKS;
            const float *src = img + (int64_t)g * F_GROUP_BLOCKS * BLOCK_FLOATS + lane * 4;
            float *dst = lds + (g & 1) * (F_GROUP_BLOCKS * BLOCK_FLOATS);
            for (int b = wave; b < nblk; b += 4) glds16(src + b * BLOCK_FLOATS, dst + b * BLOCK_FLOATS);
        }
    }
    template <int BLK>      // BLK = absolute block index in the stream (compile time)
    __device__ __forceinline__ h16x8 read_block() const {
        constexpr int off = (BLK % (2 * F_GROUP_BLOCKS)) * BLOCK_FLOATS;
        return *reinterpret_cast<const h16x8 *>(lds + off + lane * 4);
    }
};

__device__ __forceinline__ f32x16 mfma16(h16x8 a, h16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

struct Frag { h16x8 hi[2], lo[2]; };            // one 32-unit activation tile as B operands (k-steps 0,1)

__device__ __forceinline__ Frag split_tile(const f32x16 &x) {
    Frag f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = x[8 * s + j];
            const _Float16 hv = (_Float16)v;
            f.hi[s][j] = hv;
            f.lo[s][j] = (_Float16)(v - (float)hv);
        }
    return f;
}

// bias + (optional) ReLU for this mode: v_max_f32 instead of compare+select
template <bool RELU>
__device__ __forceinline__ f32x16 bias_act(const f32x16 &acc, const float *bias32, int hh) {
    f32x16 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(bias32 + 8 * q + 4 * hh);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float x = acc[4 * q + s] + b[s];
            r[4 * q + s] = RELU ? fmaxf(x, 0.f) : x;
        }
    }
    return r;
}

struct APair { h16x8 h, l; };      // [Ah | Al] fragments of one k-step

// One layer: NT output tiles, KS k-steps of 16; BASE = absolute block offset of the layer.
// bfrag(ks) -> (hi, lo) B fragments of k-step ks.  A operands are read from LDS TWO k-steps ahead
// (a k-step is only 96 MFMA cycles, less than the loaded LDS latency); `a0`/`a1` carry the fragments
// of the current and the next k-step across tiles and layers.  Reads never cross a barrier-group
// boundary early: the next group is only guaranteed to have landed after its barrier.
template <int BASE, int NT, int KS, bool LAST, class BFrag, class Epi>
__device__ __forceinline__ void run_layer_f(const StreamF &st, APair &a0, APair &a1, BFrag bfrag, Epi epi) {
    f32x16 accs[2];
    static_for<NT>([&](auto ti) {
        constexpr int T = decltype(ti)::value;
        f32x16 &acc = accs[T & 1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        static_for<KS>([&](auto ks) {
            constexpr int K = decltype(ks)::value;
            constexpr int blk = BASE + 2 * (T * KS + K);                     // Ah block of this k-step
            constexpr int left = LAST ? (NT * KS - (T * KS + K) - 1) : 1000;  // k-steps after this one
            constexpr bool group_end = (blk + 2) % F_GROUP_BLOCKS == 0;       // this is the last k-step of its group
            constexpr bool next_is_group_end = (blk + 4) % F_GROUP_BLOCKS == 0;
            if constexpr (blk % F_GROUP_BLOCKS == 0) st.issue_group(blk / F_GROUP_BLOCKS + 1);
            APair a2 = a1;
            // fragments of k-step +2 live in the same group iff neither this nor the next step ends it
            if constexpr (left >= 2 && !group_end && !next_is_group_end)
                a2 = APair{st.template read_block<blk + 4>(), st.template read_block<blk + 5>()};
            const auto b = bfrag(ks);
            acc = mfma16(a0.h, b.first, acc);
            acc = mfma16(a0.h, b.second, acc);
            acc = mfma16(a0.l, b.first, acc);
            if constexpr (group_end) {
                __syncthreads();                                             // next group landed, this half is free
                if constexpr (left >= 1) a1 = APair{st.template read_block<blk + 2>(), st.template read_block<blk + 3>()};
                if constexpr (left >= 2) a2 = APair{st.template read_block<blk + 4>(), st.template read_block<blk + 5>()};
            } else if constexpr (next_is_group_end) {
                a2 = a1;                                                      // refilled after the next barrier
            }
            a0 = a1; a1 = a2;
            if constexpr (K == 1 && T > 0) epi(ic<T - 1>{}, accs[(T - 1) & 1]);
        });
    });
    epi(ic<NT - 1>{}, accs[(NT - 1) & 1]);
}

struct FragPair { h16x8 first, second; };

template <bool FROM_RAYS>
__global__ __launch_bounds__(256, 1) void mlp_forward_f16x3_kernel(
    const float *__restrict__ img, const float *__restrict__ in_a, const float *__restrict__ in_b, int64_t P, int S,
    float *__restrict__ raw) {
    __shared__ __attribute__((aligned(16))) float lds[F_LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hh = lane >> 5;
    int64_t p = (int64_t)blockIdx.x * 128 + wave * 32 + j;
    const bool live = p < P;
    if (!live) p = P - 1;

    StreamF st{img, lds, wave, lane};
    for (int b = wave; b < SEC_B_FLOATS / BLOCK_FLOATS; b += 4)        // section B (fp32 small vectors)
        glds16(img + SEC_A_FLOATS + b * BLOCK_FLOATS + lane * 4, lds + F_RING_FLOATS + b * BLOCK_FLOATS);
    st.issue_group(0);

    float px, py, pz, vx, vy, vz;
    load_point<FROM_RAYS>(in_a, in_b, p, S, px, py, pz, vx, vy, vz);
    Frag emb[2];
    {
        f32x16 t;
        encode_tile<63>(px, py, pz, hh, 0, t); emb[0] = split_tile(t);
        encode_tile<63>(px, py, pz, hh, 1, t); emb[1] = split_tile(t);
    }
    __syncthreads();
    const float *sb = lds + F_RING_FLOATS;
    APair a0{st.read_block<0>(), st.read_block<1>()}, a1{st.read_block<2>(), st.read_block<3>()};

    Frag h[8], o[8];
    auto from = [](const Frag *x, int t, int s) { return FragPair{x[t].hi[s], x[t].lo[s]}; };

    // layer 0
    run_layer_f<OFF_L0, L0_NT, L0_KG / 2, false>(st, a0, a1,
        [&](auto ks) { return from(emb, ks.value >> 1, ks.value & 1); },
        [&](auto ti, const f32x16 &acc) { o[ti.value] = split_tile(bias_act<true>(acc, sb + SB_BIAS + 32 * ti.value, hh)); });
#pragma unroll
    for (int t = 0; t < 8; ++t) h[t] = o[t];
    // layers 1..4
    static_for<4>([&](auto li) {
        constexpr int l = 1 + decltype(li)::value;
        run_layer_f<OFF_L1 + (l - 1) * LH_BLOCKS, LH_NT, LH_KG / 2, false>(st, a0, a1,
            [&](auto ks) { return from(h, ks.value >> 1, ks.value & 1); },
            [&](auto ti, const f32x16 &acc) {
                o[ti.value] = split_tile(bias_act<true>(acc, sb + SB_BIAS + l * 256 + 32 * ti.value, hh));
            });
#pragma unroll
        for (int t = 0; t < 8; ++t) h[t] = o[t];
    });
    // layer 5: cat[encoded point, h4]
    run_layer_f<OFF_L5, L5_NT, L5_KG / 2, false>(st, a0, a1,
        [&](auto ks) {
            if constexpr (ks.value < 4) return from(emb, ks.value >> 1, ks.value & 1);
            else return from(h, (ks.value - 4) >> 1, (ks.value - 4) & 1);
        },
        [&](auto ti, const f32x16 &acc) {
            o[ti.value] = split_tile(bias_act<true>(acc, sb + SB_BIAS + 5 * 256 + 32 * ti.value, hh));
        });
#pragma unroll
    for (int t = 0; t < 8; ++t) h[t] = o[t];
    // layer 6
    run_layer_f<OFF_L6, LH_NT, LH_KG / 2, false>(st, a0, a1,
        [&](auto ks) { return from(h, ks.value >> 1, ks.value & 1); },
        [&](auto ti, const f32x16 &acc) {
            o[ti.value] = split_tile(bias_act<true>(acc, sb + SB_BIAS + 6 * 256 + 32 * ti.value, hh));
        });
#pragma unroll
    for (int t = 0; t < 8; ++t) h[t] = o[t];
    // layer 7 (+ the fp32 sigma head accumulated from the fp32 activations in the epilogue)
    float sigma = 0.f;
    run_layer_f<OFF_L6 + LH_BLOCKS, LH_NT, LH_KG / 2, false>(st, a0, a1,
        [&](auto ks) { return from(h, ks.value >> 1, ks.value & 1); },
        [&](auto ti, const f32x16 &acc) {
            const f32x16 a7 = bias_act<true>(acc, sb + SB_BIAS + 7 * 256 + 32 * ti.value, hh);
            sigma += dot_tiles<1>(&a7, sb + SB_WALPHA + 32 * ti.value, hh);
            o[ti.value] = split_tile(a7);
        });
#pragma unroll
    for (int t = 0; t < 8; ++t) h[t] = o[t];
    sigma += __shfl_xor(sigma, 32, 64);
    sigma += sb[SB_BALPHA];
    // feature (no activation)
    run_layer_f<OFF_FEAT, LH_NT, LH_KG / 2, false>(st, a0, a1,
        [&](auto ks) { return from(h, ks.value >> 1, ks.value & 1); },
        [&](auto ti, const f32x16 &acc) { o[ti.value] = split_tile(bias_act<false>(acc, sb + SB_BFEAT + 32 * ti.value, hh)); });
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
        [&](auto ti, const f32x16 &acc) {
            const f32x16 v = bias_act<true>(acc, sb + SB_BVIEWS + 32 * ti.value, hh);
            r0 += dot_tiles<1>(&v, sb + SB_WRGB + 32 * ti.value, hh);
            r1 += dot_tiles<1>(&v, sb + SB_WRGB + 128 + 32 * ti.value, hh);
            r2 += dot_tiles<1>(&v, sb + SB_WRGB + 256 + 32 * ti.value, hh);
        });
    r0 += __shfl_xor(r0, 32, 64);
    r1 += __shfl_xor(r1, 32, 64);
    r2 += __shfl_xor(r2, 32, 64);
    if (live && hh == 0)
        reinterpret_cast<float4 *>(raw)[p] = make_float4(r0 + sb[SB_BRGB], r1 + sb[SB_BRGB + 1], r2 + sb[SB_BRGB + 2], sigma);
}

static int launch_f16x3(const float *img, const float *a, const float *b, int64_t P, int S, float *raw, bool from_rays,
                        void *stream) {
    if (P == 0) return MVIP_OK;
    const dim3 grid((unsigned)((P + 127) / 128)), block(256);
    if (from_rays) hipLaunchKernelGGL((mlp_forward_f16x3_kernel<true>), grid, block, 0, as_stream(stream), img, a, b, P, S, raw);
    else hipLaunchKernelGGL((mlp_forward_f16x3_kernel<false>), grid, block, 0, as_stream(stream), img, a, b, P, S, raw);
    return check_launch();
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
