// raw2outputs forward + backward (DS_NeRF/run_nerf_helpers.py:350-404).
//
// One 64-lane wavefront per ray, ITEMS consecutive samples per lane, so the [S,4] raw rows of a
// ray are read as one contiguous 16*S-byte run (16 B per lane per item) and the transmittance
// cumprod / suffix sums are wave-level scans on shuffles: no LDS, no atomics.  HBM-bound:
// forward reads 16+4(+4) B and writes 4(+4) B per sample (+24 B per ray).
#include "composite_device.h"

namespace mvip {

template <int ITEMS>
__global__ __launch_bounds__(256) void composite_fwd_kernel(
    const float *__restrict__ raw, const float *__restrict__ z, const float *__restrict__ rows, int ncols,
    const float *__restrict__ noise, int64_t B, int S, int flags, float *__restrict__ rgb,
    float *__restrict__ disp, float *__restrict__ acc, float *__restrict__ depth,
    float *__restrict__ weights, float *__restrict__ alpha) {
    // wave index through readfirstlane: the ray number and the row bases are scalar values, not 64-bit vector arithmetic per access
    const int64_t ray = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (ray >= B) return;
    const int l = lane_id();
    RayState<ITEMS> st;
    float sums[5];
    ray_forward<ITEMS>(raw + ray * S * 4, z + ray * S, noise ? noise + ray * S : nullptr,
                       dir_norm(rows + ray * ncols), S, st, sums);
    composite_store<ITEMS>(st, sums, ray, S, flags, rgb, disp, acc, depth, weights, alpha);
}

template <int ITEMS>
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float *__restrict__ raw, const float *__restrict__ z, const float *__restrict__ rows, int ncols,
    const float *__restrict__ noise, int64_t B, int S, int flags, const float *__restrict__ g_rgb,
    const float *__restrict__ g_disp, const float *__restrict__ g_acc, const float *__restrict__ g_depth,
    const float *__restrict__ g_w, const float *__restrict__ g_alpha, float *__restrict__ d_raw) {
    // wave index through readfirstlane: the ray number and the row bases are scalar values, not 64-bit vector arithmetic per access
    const int64_t ray = (int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (ray >= B) return;
    const int l = lane_id();
    RayState<ITEMS> st;
    float sums[5];
    ray_forward<ITEMS>(raw + ray * S * 4, z + ray * S, noise ? noise + ray * S : nullptr,
                       dir_norm(rows + ray * ncols), S, st, sums);
    const float a = sums[0], d = sums[1];
    float gc[3] = {0.f, 0.f, 0.f};
    if (g_rgb) { gc[0] = g_rgb[ray * 3]; gc[1] = g_rgb[ray * 3 + 1]; gc[2] = g_rgb[ray * 3 + 2]; }
    // disp = 1 / max(1e-10, depth/acc)
    float gq = 0.f;
    if (g_disp) {
        const float q = d / a;
        const float m = (q != q) ? q : fmaxf(1e-10f, q);
        const float gm = -g_disp[ray] / (m * m);
        gq = (1e-10f > q) ? 0.f : (q == 1e-10f ? .5f * gm : gm);
    }
    float gd = (g_depth ? g_depth[ray] : 0.f);
    float ga = (g_acc ? g_acc[ray] : 0.f);
    if (g_disp) { gd += gq / a; ga -= gq * d / (a * a); }
    if (flags & MVIP_COMP_WHITE) ga -= (gc[0] + gc[1]) + gc[2];
    const bool detach = flags & MVIP_COMP_DETACHW;

    float Gw[ITEMS], lane_sfx = 0.f;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int s = l * ITEMS + i;
        float g = gd * st.z[i] + ga;
        if (g_w && st.valid[i]) g += g_w[ray * S + s];
        if (!detach) g += (gc[0] * st.c[i][0] + gc[1] * st.c[i][1]) + gc[2] * st.c[i][2];
        Gw[i] = st.valid[i] ? g : 0.f;
        lane_sfx += Gw[i] * st.w[i];
    }
    // exclusive suffix sum of G_w*w over samples after s
    float incl = wave_incl_suffix_sum(lane_sfx);
    float after = __shfl_down(incl, 1, 64);
    if (l == 63) after = 0.f;
    float run = after;
#pragma unroll
    for (int i = ITEMS - 1; i >= 0; --i) {
        const int s = l * ITEMS + i;
        const float R = run;                         // sum over k > s
        run += Gw[i] * st.w[i];
        if (!st.valid[i]) continue;
        float da = Gw[i] * st.T[i] - R / st.t[i];
        if (g_alpha) da += g_alpha[ray * S + s];
        const float dsig = da * st.dist[i] * st.e[i];
        float4 o;
        const float wv = st.w[i];
        o.x = gc[0] * wv * st.c[i][0] * (1.f - st.c[i][0]);
        o.y = gc[1] * wv * st.c[i][1] * (1.f - st.c[i][1]);
        o.z = gc[2] * wv * st.c[i][2] * (1.f - st.c[i][2]);
        o.w = st.sig[i] > 0.f ? dsig : 0.f;
        reinterpret_cast<float4 *>(d_raw)[ray * S + s] = o;
    }
}

}  // namespace mvip

using namespace mvip;

#define DISPATCH_ITEMS(S, CALL)                         \
    if ((S) <= 64) { CALL(1); }                         \
    else if ((S) <= 128) { CALL(2); }                   \
    else if ((S) <= 256) { CALL(4); }                   \
    else if ((S) <= 512) { CALL(8); }                   \
    else return MVIP_EUNSUP;

extern "C" int mvip_composite_forward(const float *raw, const float *z, const float *rows, int ncols,
                                      const float *noise, int64_t B, int S, int flags, float *rgb, float *disp,
                                      float *acc, float *depth, float *weights, float *alpha, void *stream) {
    if (B < 0 || S <= 0 || ncols < 6) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    if (!raw || !z || !rows || !rgb || !disp || !acc || !depth || !weights) return MVIP_EINVAL;
    const dim3 grid((unsigned)((B + 3) / 4)), block(256);
#define CALL(I) hipLaunchKernelGGL(composite_fwd_kernel<I>, grid, block, 0, as_stream(stream), raw, z, rows, \
                                   ncols, noise, B, S, flags, rgb, disp, acc, depth, weights, alpha)
    DISPATCH_ITEMS(S, CALL)
#undef CALL
    return check_launch();
}

extern "C" int mvip_composite_backward(const float *raw, const float *z, const float *rows, int ncols,
                                       const float *noise, int64_t B, int S, int flags, const float *g_rgb,
                                       const float *g_disp, const float *g_acc, const float *g_depth,
                                       const float *g_weights, const float *g_alpha, float *d_raw,
                                       void *stream) {
    if (B < 0 || S <= 0 || ncols < 6) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    if (!raw || !z || !rows || !d_raw) return MVIP_EINVAL;
    const dim3 grid((unsigned)((B + 3) / 4)), block(256);
#define CALL(I) hipLaunchKernelGGL(composite_bwd_kernel<I>, grid, block, 0, as_stream(stream), raw, z, rows, \
                                   ncols, noise, B, S, flags, g_rgb, g_disp, g_acc, g_depth, g_weights,       \
                                   g_alpha, d_raw)
    DISPATCH_ITEMS(S, CALL)
#undef CALL
    return check_launch();
}
