// Delta propagation of the fused NeRF MLP backward with TWO waves per SIMD (exact fp32, v_mfma_f32_16x16x4_f32):
// the 16-points-per-wave counterpart of mlp_delta_kernel (mlp_bwd.hip), as mlp_fwd16.hip is of the forward.
// (Autograd through NeRF.forward, DS_NeRF/run_nerf_helpers.py:104-127.)
//
//   G_l = relu'(h_l) . W_{l+1}^T G_{l+1}
// A wave owns 16 points; its [256 x 16] gradient matrix lives in 64 registers as sixteen 16 x 16 accumulator tiles, each
// directly the B operand of the next layer's K = 4 steps (mlp_device16.h); the TRANSPOSED weights stream through the
// shared LDS ring from an image in 16-point block order (mlp_pack_transposed16_kernel).  ReLU masks come from the
// activation stash, pre-activation gradients go to the gradient stash -- both in the [row tile][point tile][32][32]
// layout of the 32-point kernels, so the weight-gradient kernel reads them unchanged.  With one wave per SIMD the
// 32-point kernel's stash loads, stores and epilogues each cost matrix-pipe idle time (0.80 of peak); here they issue
// under the SIMD partner's MFMAs.
#include "mlp_device16.h"

namespace mvip {
using namespace mlp;

// the transposed stream: views^T (feature part), feature^T, then layers 7,6,5(h4 part),4,3,2,1 transposed
constexpr int T16_VIEWS_BLOCKS = 16 * 8;               // 16 out tiles (256 feature units) x 8 in tiles (128 view units)
constexpr int T16_LAYER_BLOCKS = 16 * 16;
constexpr int T16_TOTAL_BLOCKS = T16_VIEWS_BLOCKS + 8 * T16_LAYER_BLOCKS;       // 2176, as the 32-point image
constexpr int T16_TOTAL_CHUNKS = T16_TOTAL_BLOCKS / CHUNK_BLOCKS;
constexpr int T16_FLOATS = T16_TOTAL_BLOCKS * BLOCK_FLOATS;

__device__ __forceinline__ float packed_w32(const float *__restrict__ packed, int blk_off, int KG, int out, int k) {
    const int blk = blk_off + block_pos(out >> 5, k >> 3, KG);
    return packed[(int64_t)blk * BLOCK_FLOATS + ((k & 7) >> 2) * 128 + (out & 31) * 4 + (k & 3)];
}

// block (to, ti): lane (m, g) holds W^T[16 to + m][16 ti + 4 g + s] = W[16 ti + 4 g + s][16 to + m], s = 0..3
__global__ void mlp_pack_transposed16_kernel(const float *__restrict__ packed, float *__restrict__ pt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T16_FLOATS) return;
    const int blk = idx / BLOCK_FLOATS, r = idx % BLOCK_FLOATS;
    const int lane = r / 4, s = r % 4, m = lane & 15, g = lane >> 4;
    float v;
    if (blk < T16_VIEWS_BLOCKS) {
        const int to = blk / 8, ti = blk % 8;
        v = packed_w32(packed, OFF_VIEWS, LV_KG, 16 * ti + 4 * g + s, 16 * to + m);
    } else {
        const int lm = (blk - T16_VIEWS_BLOCKS) / T16_LAYER_BLOCKS;        // 0: feature, 1..7: layers 7..1
        const int local = (blk - T16_VIEWS_BLOCKS) % T16_LAYER_BLOCKS;
        const int to = local / 16, ti = local % 16;
        const int out = 16 * ti + 4 * g + s, in = 16 * to + m;
        if (lm == 0) v = packed_w32(packed, OFF_FEAT, LH_KG, out, in);
        else {
            const int l = 8 - lm;
            if (l >= 6) v = packed_w32(packed, OFF_L6 + (l - 6) * LH_BLOCKS, LH_KG, out, in);
            else if (l == 5) v = packed_w32(packed, OFF_L5, L5_KG, out, 64 + in);
            else v = packed_w32(packed, OFF_L1 + (l - 1) * LH_BLOCKS, LH_KG, out, in);
        }
    }
    pt[idx] = v;
}

namespace f16p {

__global__ void __launch_bounds__(512, 2)
mlp_delta16_kernel(const float *__restrict__ packed_t, const float *__restrict__ secb, const float *__restrict__ d_raw,
                   int64_t p_begin, int64_t p_count, const float *__restrict__ act, int64_t act_n_pt, int64_t act_pt0,
                   float *__restrict__ gst, int64_t n_pt) {
    __shared__ __attribute__((aligned(16))) float lds[LDS16_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    const int64_t pl = (int64_t)blockIdx.x * WG_POINTS + wave * 16 + n;       // index inside [0, p_count)
    const bool live = pl < p_count;

    Stream16 st{packed_t, lds, wave, lane, T16_TOTAL_CHUNKS};
    for (int b = wave; b < SEC_B_FLOATS / BLOCK_FLOATS; b += 8)
        glds<0>(secb + b * BLOCK_FLOATS + lane * 4, lds + RING16_FLOATS + b * BLOCK_FLOATS);
    st.issue_chunk(0, 0);
    st.issue_chunk(1, 1);

    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) d = reinterpret_cast<const float4 *>(d_raw)[p_begin + pl];

    // stash addressing as in mlp_forward16_kernel<STASH>: wave-uniform block base + a per-lane constant
    const int64_t pt_wave = (int64_t)blockIdx.x * 4 + (wave >> 1);
    const int lane_off = (4 * g) * 32 + 16 * (wave & 1) + n;
    auto put = [&](int t16, const f32x4 &t) {
        float *q = gst + ((int64_t)(t16 >> 1) * n_pt + pt_wave) * 1024 + (t16 & 1) * 512 + lane_off;
        q[0] = t[0]; q[32] = t[1]; q[64] = t[2]; q[96] = t[3];
    };
    auto act_tile = [&](int t16) {
        const float *q = act + ((int64_t)(t16 >> 1) * act_n_pt + act_pt0 + pt_wave) * 1024 + (t16 & 1) * 512 + lane_off;
        return f32x4{q[0], q[32], q[64], q[96]};
    };
    {   // d_raw^T: rows 0..3 of row tile GT_D (registers 0..3 of the lanes with g = 0), the rest of the tile zero
        const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
        put(2 * GT_D, g == 0 ? f32x4{d.x, d.y, d.z, d.w} : zero);
        put(2 * GT_D + 1, zero);
    }
    f32x4 vt[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) vt[t] = act_tile(2 * AT_V + t);

    __syncthreads();                                   // chunks 0, 1 and section B have landed
    const float *sb = lds + RING16_FLOATS;
    f32x4 a = st.read_block<0>();

    // grad wrt the view-branch pre-activation: relu'(v) . (W_rgb^T d_rgb)
    f32x4 gv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const int u0 = 16 * t + 4 * g;
        const f32x4 w0 = *reinterpret_cast<const f32x4 *>(sb + SB_WRGB + u0);
        const f32x4 w1 = *reinterpret_cast<const f32x4 *>(sb + SB_WRGB + 128 + u0);
        const f32x4 w2 = *reinterpret_cast<const f32x4 *>(sb + SB_WRGB + 256 + u0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float val = fmaf(w2[i], d.z, fmaf(w1[i], d.y, w0[i] * d.x));
            gv[t][i] = vt[t][i] > 0.f ? val : 0.f;
        }
        put(2 * GT_V + t, gv[t]);
    }

    // grad wrt feature = W_views[:, :256]^T gv   (feature_linear has no activation)
    f32x4 gg[16], gn[16];
    layer16x<0, 16, 8, false, false>(st, a, nullptr, [&](auto ti) { return gv[ti.value]; }, NoPre16{},
        [&](auto to, const f32x4 &acc, int) { gg[to.value] = acc; put(2 * GT_F + to.value, acc); });

    // G_7 = relu'(h7) . (W_feat^T g_feat + w_alpha d_sigma)
    layer16x<T16_VIEWS_BLOCKS, 16, 16, false, false>(st, a, nullptr, [&](auto ti) { return gg[ti.value]; },
        [&](auto to) { return act_tile(2 * (AT_H + 56) + to.value); },
        [&](auto to, const f32x4 &acc, const f32x4 &hv) {
            const f32x4 wa = *reinterpret_cast<const f32x4 *>(sb + SB_WALPHA + 16 * to.value + 4 * g);
            f32x4 o;
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = hv[i] > 0.f ? fmaf(wa[i], d.w, acc[i]) : 0.f;
            gn[to.value] = o;
            put(2 * (GT_G + 56) + to.value, o);
        });
#pragma unroll
    for (int t = 0; t < 16; ++t) gg[t] = gn[t];

    // G_m = relu'(h_m) . W_{m+1}^T G_{m+1},  m = 6..0   (layer 5 contributes its h4 columns only)
    static_for<7>([&](auto mi) {
        constexpr int m = 6 - decltype(mi)::value;
        layer16x<T16_VIEWS_BLOCKS + (7 - m) * T16_LAYER_BLOCKS, 16, 16, m == 0, false>(st, a, nullptr,
            [&](auto ti) { return gg[ti.value]; },
            [&](auto to) { return act_tile(2 * (AT_H + 8 * m) + to.value); },
            [&](auto to, const f32x4 &acc, const f32x4 &hv) {
                f32x4 o;
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = hv[i] > 0.f ? acc[i] : 0.f;
                gn[to.value] = o;
                put(2 * (GT_G + 8 * m) + to.value, o);
            });
#pragma unroll
        for (int t = 0; t < 16; ++t) gg[t] = gn[t];
    });
}

}  // namespace f16p

// image_t16: T16_FLOATS floats (the same size as the 32-point transposed image, so it takes its place in the workspace)
int mlp_delta16_prepare(const float *packed, float *image_t16, void *stream) {
    hipLaunchKernelGGL(mlp_pack_transposed16_kernel, dim3((T16_FLOATS + 255) / 256), dim3(256), 0, as_stream(stream), packed,
                       image_t16);
    return check_launch();
}

// n_pt: point tiles of the gradient stash (a multiple of 4: one workgroup = 128 points = 4 point tiles)
int mlp_delta16_launch(const float *image_t16, const float *secb, const float *d_raw, int64_t p0, int64_t pc,
                       const float *act, int64_t act_n_pt, int64_t act_pt0, float *gst, int64_t n_pt, void *stream) {
    hipLaunchKernelGGL(f16p::mlp_delta16_kernel, dim3((unsigned)(n_pt / 4)), dim3(512), 0, as_stream(stream), image_t16, secb,
                       d_raw, p0, pc, act, act_n_pt, act_pt0, gst, n_pt);
    return check_launch();
}

}  // namespace mvip
