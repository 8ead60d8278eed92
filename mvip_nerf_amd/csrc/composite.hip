// raw2outputs forward + backward (DS_NeRF/run_nerf_helpers.py:350-404).
//
// One 64-lane wavefront per ray, ITEMS consecutive samples per lane, so the [S,4] raw rows of a
// ray are read as one contiguous 16*S-byte run (16 B per lane per item) and the transmittance
// cumprod / suffix sums are wave-level scans on shuffles: no LDS, no atomics.  HBM-bound:
// forward reads 16+4(+4) B and writes 4(+4) B per sample (+24 B per ray).
#include "common.h"

namespace mvip {

template <int ITEMS>
struct RayState {
    float z[ITEMS], dist[ITEMS], sig[ITEMS], e[ITEMS], alpha[ITEMS], t[ITEMS], T[ITEMS], w[ITEMS];
    float c[ITEMS][3];
    bool valid[ITEMS];
};

// Recomputes everything the forward defines for one ray.  Returns (acc, depth, rgb sums).
template <int ITEMS>
__device__ __forceinline__ void ray_forward(const float *__restrict__ raw, const float *__restrict__ z,
                                            const float *__restrict__ noise, float dnorm, int S,
                                            RayState<ITEMS> &st, float sums[5]) {
    const int l = lane_id();
    float zfirst_next;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int s = l * ITEMS + i;
        st.valid[i] = s < S;
        st.z[i] = st.valid[i] ? z[s] : 0.f;
    }
    zfirst_next = __shfl_down(st.z[0], 1, 64);
    float lane_prod = 1.f;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int s = l * ITEMS + i;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        float nz = 0.f;
        if (st.valid[i]) {
            r = reinterpret_cast<const float4 *>(raw)[s];
            if (noise) nz = noise[s];
        }
        const float znext = (i + 1 < ITEMS) ? st.z[(i + 1) % ITEMS] : zfirst_next;
        float d = (s == S - 1) ? 1e10f : (znext - st.z[i]);
        d = d * dnorm;
        const float pre = r.w + nz;
        const float sg = pre > 0.f ? pre : 0.f;                 // relu
        const float ee = st.valid[i] ? expf(-sg * d) : 1.f;
        const float a = 1.f - ee;                               // raw2alpha
        st.dist[i] = d; st.sig[i] = pre; st.e[i] = ee; st.alpha[i] = a;
        st.t[i] = (1.f - a) + 1e-10f;
        st.c[i][0] = 1.f / (1.f + expf(-r.x));                  // sigmoid
        st.c[i][1] = 1.f / (1.f + expf(-r.y));
        st.c[i][2] = 1.f / (1.f + expf(-r.z));
        st.T[i] = lane_prod;                                    // exclusive product inside the lane
        lane_prod *= st.t[i];
    }
    float incl = wave_incl_prod(lane_prod);
    float excl = __shfl_up(incl, 1, 64);
    if (l == 0) excl = 1.f;
    float a_sum = 0.f, d_sum = 0.f, c0 = 0.f, c1 = 0.f, c2 = 0.f;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        st.T[i] *= excl;
        st.w[i] = st.valid[i] ? st.alpha[i] * st.T[i] : 0.f;
        a_sum += st.w[i];
        d_sum += st.w[i] * st.z[i];
        c0 += st.w[i] * st.c[i][0];
        c1 += st.w[i] * st.c[i][1];
        c2 += st.w[i] * st.c[i][2];
    }
    sums[0] = wave_sum(a_sum); sums[1] = wave_sum(d_sum);
    sums[2] = wave_sum(c0); sums[3] = wave_sum(c1); sums[4] = wave_sum(c2);
}

__device__ __forceinline__ float dir_norm(const float *__restrict__ row) {
    const float x = row[3], y = row[4], zc = row[5];
    return sqrtf((x * x + y * y) + zc * zc);
}

template <int ITEMS>
__global__ __launch_bounds__(256) void composite_fwd_kernel(
    const float *__restrict__ raw, const float *__restrict__ z, const float *__restrict__ rows, int ncols,
    const float *__restrict__ noise, int64_t B, int S, int flags, float *__restrict__ rgb,
    float *__restrict__ disp, float *__restrict__ acc, float *__restrict__ depth,
    float *__restrict__ weights, float *__restrict__ alpha) {
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= B) return;
    const int l = lane_id();
    RayState<ITEMS> st;
    float sums[5];
    ray_forward<ITEMS>(raw + ray * S * 4, z + ray * S, noise ? noise + ray * S : nullptr,
                       dir_norm(rows + ray * ncols), S, st, sums);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int s = l * ITEMS + i;
        if (st.valid[i]) {
            weights[ray * S + s] = st.w[i];
            if (alpha) alpha[ray * S + s] = st.alpha[i];
        }
    }
    if (l == 0) {
        const float a = sums[0], d = sums[1];
        const float q = d / a;
        const float m = (q != q) ? q : fmaxf(1e-10f, q);        // torch.max propagates NaN (0/0 rays)
        const float white = (flags & MVIP_COMP_WHITE) ? (1.f - a) : 0.f;
        rgb[ray * 3 + 0] = sums[2] + white;
        rgb[ray * 3 + 1] = sums[3] + white;
        rgb[ray * 3 + 2] = sums[4] + white;
        disp[ray] = 1.f / m;
        acc[ray] = a;
        depth[ray] = d;
    }
}

template <int ITEMS>
__global__ __launch_bounds__(256) void composite_bwd_kernel(
    const float *__restrict__ raw, const float *__restrict__ z, const float *__restrict__ rows, int ncols,
    const float *__restrict__ noise, int64_t B, int S, int flags, const float *__restrict__ g_rgb,
    const float *__restrict__ g_disp, const float *__restrict__ g_acc, const float *__restrict__ g_depth,
    const float *__restrict__ g_w, const float *__restrict__ g_alpha, float *__restrict__ d_raw) {
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
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
