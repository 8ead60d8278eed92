// Split-precision ("f16x3", precision = 1) versions of the backward's delta kernel and of the
// transposed weight image; see mlp_bwd.hip for the algorithm and mlp_fwd_f16x3.hip for the
// arithmetic.  Gradients G_l are produced, masked, stored (fp32 stash) and re-split in registers
// exactly like activations in the forward.
#include "common.h"
#include "mlp_layout.h"
#include "mlp_device.h"
#include "mlp_device_f16.h"

namespace mvip {
using namespace mlp;

constexpr int T16_VIEWS_BLOCKS = 8 * 8 * 2;            // 8 out tiles (feature units) x 8 k-steps x [hi|lo]
constexpr int T16_LAYER_BLOCKS = 8 * 16 * 2;
constexpr int T16_TOTAL_BLOCKS = T16_VIEWS_BLOCKS + 8 * T16_LAYER_BLOCKS;     // 2176, as the fp32 transposed image
constexpr int T16_FLOATS = T16_TOTAL_BLOCKS * BLOCK_FLOATS;
static_assert(T16_VIEWS_BLOCKS % F_GROUP_BLOCKS == 0 && T16_LAYER_BLOCKS % F_GROUP_BLOCKS == 0, "group alignment");

// W[out][k] of a layer (identified by its block offset / k-steps per tile) read back from the
// forward f16x3 image: hi + lo is exact in fp32
__device__ __forceinline__ float image16_w(const _Float16 *__restrict__ img, int blk_off, int ksn, int out, int k) {
    const int ks = k >> 4, r = k & 15;
    const int j = ((r >> 3) << 2) | (r & 3), h = (r & 7) >> 2;
    const int64_t blk = blk_off + 2 * ((out >> 5) * ksn + ks);
    const int64_t e = blk * 512 + ((h * 32 + (out & 31)) * 8 + j);
    return (float)img[e] + (float)img[e + 512];
}

__global__ void mlp_pack_transposed_f16x3_kernel(const _Float16 *__restrict__ img, _Float16 *__restrict__ out) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T16_FLOATS * 2) return;
    const int blk = idx / 512, e = idx % 512;
    const int lane = e / 8, j = e % 8, h = lane / 32, i = lane % 32;
    int local, ksn;
    float w;
    if (blk < T16_VIEWS_BLOCKS) { local = blk; ksn = 8; }
    else { local = (blk - T16_VIEWS_BLOCKS) % T16_LAYER_BLOCKS; ksn = 16; }
    const int pair = local / 2, lo = local % 2, ti = pair / ksn, ks = pair % ksn;
    const int in = 32 * ti + i;                                   // row of W^T = input unit
    const int o = 16 * ks + 8 * (j >> 2) + 4 * h + (j & 3);       // column of W^T = output unit
    if (blk < T16_VIEWS_BLOCKS) w = image16_w(img, OFF_VIEWS, LV_KG / 2, o, in);
    else {
        const int m = (blk - T16_VIEWS_BLOCKS) / T16_LAYER_BLOCKS;        // 0: feature, 1..7: layers 7..1
        if (m == 0) w = image16_w(img, OFF_FEAT, LH_KG / 2, o, in);
        else {
            const int l = 8 - m;
            if (l >= 6) w = image16_w(img, OFF_L6 + (l - 6) * LH_BLOCKS, LH_KG / 2, o, in);
            else if (l == 5) w = image16_w(img, OFF_L5, L5_KG / 2, o, 64 + in);
            else w = image16_w(img, OFF_L1 + (l - 1) * LH_BLOCKS, LH_KG / 2, o, in);
        }
    }
    const _Float16 wh = (_Float16)w;
    out[idx] = lo ? (_Float16)(w - (float)wh) : wh;
}

__global__ __launch_bounds__(256, 1) void mlp_delta_f16x3_kernel(
    const float *__restrict__ img_t, const float *__restrict__ secb, const float *__restrict__ d_raw,
    int64_t p_begin, int64_t p_count, const float *__restrict__ act, int64_t act_n_pt, int64_t act_pt0,
    float *__restrict__ gst, int64_t n_pt) {
    __shared__ __attribute__((aligned(16))) float lds[F_LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hh = lane >> 5;
    const int64_t pt = (int64_t)blockIdx.x * 4 + wave;
    const int64_t pl = pt * 32 + j;
    const bool live = pl < p_count;

    StreamF st{img_t, lds, wave, lane, T16_TOTAL_BLOCKS};
    st.init_bases();
    for (int b = wave; b < SEC_B_FLOATS / BLOCK_FLOATS; b += 4)
        glds16(secb + b * BLOCK_FLOATS + lane * 4, lds + F_RING_FLOATS + b * BLOCK_FLOATS);
    st.issue_group(0);

    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) d = reinterpret_cast<const float4 *>(d_raw)[p_begin + pl];
    {
        f32x16 dt;
#pragma unroll
        for (int r = 0; r < 16; ++r) dt[r] = 0.f;
        if (hh == 0) { dt[0] = d.x; dt[1] = d.y; dt[2] = d.z; dt[3] = d.w; }
        store_tile<true>(stash_block(gst, GT_D, n_pt, pt), dt, j, hh);
    }
    // Per-point power-of-two scaling.  Delta propagation is linear in each point's d_raw, and a
    // column scale of the B operand is the same column scale of the MFMA result, so every point is
    // normalised to max|d| in [16, 32) on entry (exact) and un-scaled (exact) when its gradients are
    // stored.  Without it the small training gradients (1e-3 .. 1e-8) push the fp16 lo terms into
    // the subnormal range and the split loses its low bits.
    float inv_s = 1.f;
    {
        const float mx = fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w)));
        int e = ((__float_as_int(mx) >> 23) & 255) - 127;           // floor(log2(mx)) for normal mx
        if (!(mx > 0.f) || mx != mx || e > 120) e = 4;               // zero / NaN / huge: leave as is
        if (e < -116) e = -116;
        const float s = __int_as_float((127 + 4 - e) << 23);
        inv_s = __int_as_float((127 + e - 4) << 23);
        d.x *= s; d.y *= s; d.z *= s; d.w *= s;
    }
    // relu'(h) = [h > 0]: the forward left one sign bit per (unit, point) next to the activations -- 128 B per tile and 32
    // points instead of the tile's 4 KB (this kernel's reads were 9.7 KB per point, half of its traffic)
    auto act_mask = [&](int mi) -> unsigned {
        return *mask_slot(const_cast<float *>(act), mi, act_n_pt, act_pt0 + pt, lane);
    };
    auto put = [&](int row_tile, const f32x16 &t) {
        f32x16 u;
#pragma unroll
        for (int r = 0; r < 16; ++r) u[r] = t[r] * inv_s;
        store_tile<true>(stash_block(gst, row_tile, n_pt, pt), u, j, hh);
    };

    unsigned vt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) vt[t] = act_mask(64 + t);
    __syncthreads();
    const float *sb = lds + F_RING_FLOATS;
    APair a0{st.read_block<0>(), st.read_block<1>()}, a1{st.read_block<2>(), st.read_block<3>()};

    Frag gv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        f32x16 gt;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u0 = 32 * t + 8 * q + 4 * hh;
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(sb + SB_WRGB + u0);
            const f32x4 w1 = *reinterpret_cast<const f32x4 *>(sb + SB_WRGB + 128 + u0);
            const f32x4 w2 = *reinterpret_cast<const f32x4 *>(sb + SB_WRGB + 256 + u0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float val = fmaf(w2[s], d.z, fmaf(w1[s], d.y, w0[s] * d.x));
                gt[4 * q + s] = ((vt[t] >> (4 * q + s)) & 1u) ? val : 0.f;
            }
        }
        put(GT_V + t, gt);
        gv[t] = split_tile(gt);
    }

    Frag g[8], gn[8];
    // grad wrt feature = W_views[:, :256]^T gv
    run_layer_f<0, 8, 8, false>(st, a0, a1,
        [&](auto ks) { return FragPair{gv[ks.value >> 1].hi[ks.value & 1], gv[ks.value >> 1].lo[ks.value & 1]}; },
        NoPre{}, [&](auto ti, const f32x16 &acc, int) { put(GT_F + ti.value, acc); g[ti.value] = split_tile(acc); });
    // G_7 = relu'(h7) . (W_feat^T g_feat + w_alpha d_sigma)
    run_layer_f<T16_VIEWS_BLOCKS, 8, 16, false>(st, a0, a1,
        [&](auto ks) { return FragPair{g[ks.value >> 1].hi[ks.value & 1], g[ks.value >> 1].lo[ks.value & 1]}; },
        [&](auto ti) { return act_mask(56 + ti.value); },
        [&](auto ti, const f32x16 &acc, unsigned hv) {
            f32x16 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wa = *reinterpret_cast<const f32x4 *>(sb + SB_WALPHA + 32 * ti.value + 8 * q + 4 * hh);
#pragma unroll
                for (int s = 0; s < 4; ++s) o[4 * q + s] = ((hv >> (4 * q + s)) & 1u) ? fmaf(wa[s], d.w, acc[4 * q + s]) : 0.f;
            }
            put(GT_G + 56 + ti.value, o);
            gn[ti.value] = split_tile(o);
        });
#pragma unroll
    for (int t = 0; t < 8; ++t) g[t] = gn[t];
    // G_m = relu'(h_m) . W_{m+1}^T G_{m+1},  m = 6..0
    static_for<7>([&](auto mi) {
        constexpr int idx = decltype(mi)::value, m = 6 - idx;
        run_layer_f<T16_VIEWS_BLOCKS + (1 + idx) * T16_LAYER_BLOCKS, 8, 16, (idx == 6)>(st, a0, a1,
            [&](auto ks) { return FragPair{g[ks.value >> 1].hi[ks.value & 1], g[ks.value >> 1].lo[ks.value & 1]}; },
            [&](auto ti) { return act_mask(8 * m + ti.value); },
            [&](auto ti, const f32x16 &acc, unsigned hv) {
                f32x16 o;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = ((hv >> r) & 1u) ? acc[r] : 0.f;
                put(GT_G + 8 * m + ti.value, o);
                gn[ti.value] = split_tile(o);
            });
#pragma unroll
        for (int t = 0; t < 8; ++t) g[t] = gn[t];
    });
}

// used by backward_impl (mlp_bwd.hip) when precision == 1
int mlp_delta_f16x3_prepare(const float *image16, float *image_t16, void *stream) {
    hipLaunchKernelGGL(mlp_pack_transposed_f16x3_kernel, dim3((T16_FLOATS * 2 + 255) / 256), dim3(256), 0,
                       as_stream(stream), reinterpret_cast<const _Float16 *>(image16),
                       reinterpret_cast<_Float16 *>(image_t16));
    return check_launch();
}

int mlp_delta_f16x3_launch(const float *image_t16, const float *secb, const float *d_raw, int64_t p0, int64_t pc,
                           const float *act, int64_t act_n_pt, int64_t act_pt0, float *gst, int64_t n_pt,
                           void *stream) {
    hipLaunchKernelGGL(mlp_delta_f16x3_kernel, dim3((unsigned)(n_pt / 4)), dim3(256), 0, as_stream(stream), image_t16,
                       secb, d_raw, p0, pc, act, act_n_pt, act_pt0, gst, n_pt);
    return check_launch();
}

}  // namespace mvip
