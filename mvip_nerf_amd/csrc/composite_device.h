// Device side of raw2outputs (DS_NeRF/run_nerf_helpers.py:350-404), shared by csrc/composite.hip (stand-alone launches,
// forward and backward) and the fused render kernels of csrc/mlp_fwd16.hip: one 64-lane wavefront per ray, ITEMS
// consecutive samples per lane, the transmittance cumprod and the five ray sums as wave-level scans on DPP (round 5: a
// cross-lane step is an operand modifier of an ordinary VALU instruction, ~8 cycles, where the ds_bpermute shuffles of the
// earlier version were an LDS-crossbar round trip each, 37 of them per ray -- the fused render kernel's tail runs this on ONE
// wave while the workgroup's other seven have left, so its latency is exposed in full).
#pragma once
#include "common.h"

namespace mvip {

template <int ITEMS>
struct RayState {
    float z[ITEMS], dist[ITEMS], sig[ITEMS], e[ITEMS], alpha[ITEMS], t[ITEMS], T[ITEMS], w[ITEMS];
    float c[ITEMS][3];
    bool valid[ITEMS];
};

// ---- the per-sample terms of raw2outputs, as functions: the stand-alone kernels evaluate them lane by lane inside ray_forward,
// the fused render kernel (mlp_fwd16.hip) evaluates them on ALL waves of the workgroup before its compositing tail -- the same
// calls on the same values, hence the same bits.
__device__ __forceinline__ float comp_dist(float z, float znext, bool last, float dnorm) {
    const float d = last ? 1e10f : (znext - z);
    return d * dnorm;
}
__device__ __forceinline__ float comp_neg_exponent(float pre, float d) {         // -relu(sigma + noise) * dist
    const float sg = pre > 0.f ? pre : 0.f;
    return -sg * d;
}
__device__ __forceinline__ float comp_sigmoid_from_exp(float e) { return 1.f / (1.f + e); }     // e = expf(-x)

// Second half of the forward: st.valid / z / e / c are filled (per sample: e = exp(-relu(sigma) dist), c = sigmoid(rgb));
// alpha, t, the transmittance T, the weights and the five ray sums follow.
template <int ITEMS>
__device__ __forceinline__ void ray_finish(RayState<ITEMS> &st, float sums[5]) {
    float lane_prod = 1.f;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const float a = 1.f - st.e[i];                          // raw2alpha
        st.alpha[i] = a;
        st.t[i] = (1.f - a) + 1e-10f;
        st.T[i] = lane_prod;                                    // exclusive product inside the lane
        lane_prod *= st.t[i];
    }
    const float incl = dpp_incl_prod(lane_prod);
    const float excl = dpp_from_prev(incl, 1.f);                // product over the lanes before this one (lane 0: 1)
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
    sums[0] = dpp_wave_sum(a_sum); sums[1] = dpp_wave_sum(d_sum);
    sums[2] = dpp_wave_sum(c0); sums[3] = dpp_wave_sum(c1); sums[4] = dpp_wave_sum(c2);
}

// Recomputes everything the forward defines for one ray.  Returns (acc, depth, rgb sums).
// z == nullptr: the caller has already put the ray's depths into st.z (a fused kernel that computed them itself).
// noise_reg != nullptr: this lane's ITEMS noise values, already in registers (a fused kernel that loaded them early).
template <int ITEMS>
__device__ __forceinline__ void ray_forward(const float *__restrict__ raw, const float *__restrict__ z,
                                            const float *__restrict__ noise, float dnorm, int S,
                                            RayState<ITEMS> &st, float sums[5], const float *noise_reg = nullptr) {
    const int l = lane_id();
    float zfirst_next;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int s = l * ITEMS + i;
        st.valid[i] = s < S;
        if (z) st.z[i] = st.valid[i] ? z[s] : 0.f;
        else if (!st.valid[i]) st.z[i] = 0.f;
    }
    zfirst_next = dpp_from_next(st.z[0], st.z[0]);          // lane l + 1's first depth (lane 63: its own, never used: s = S - 1)
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int s = l * ITEMS + i;
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
        float nz = 0.f;
        if (st.valid[i]) {
            r = reinterpret_cast<const float4 *>(raw)[s];
            if (noise_reg) nz = noise_reg[i];
            else if (noise) nz = noise[s];
        }
        const float znext = (i + 1 < ITEMS) ? st.z[(i + 1) % ITEMS] : zfirst_next;
        const float d = comp_dist(st.z[i], znext, s == S - 1, dnorm);
        const float pre = r.w + nz;
        st.dist[i] = d; st.sig[i] = pre;
        st.e[i] = st.valid[i] ? expf(comp_neg_exponent(pre, d)) : 1.f;
        st.c[i][0] = comp_sigmoid_from_exp(expf(-r.x));
        st.c[i][1] = comp_sigmoid_from_exp(expf(-r.y));
        st.c[i][2] = comp_sigmoid_from_exp(expf(-r.z));
    }
    ray_finish<ITEMS>(st, sums);
}

// The fused kernel's form: `terms` holds, per sample, {e, c0, c1, c2} as the workgroup's waves computed them (comp_* above),
// `zs` the depths; every sample of the ray exists (S = 64 ITEMS).  Same values as ray_forward on the raw rows they came from.
template <int ITEMS>
__device__ __forceinline__ void ray_forward_terms(const float *__restrict__ terms, const float *__restrict__ zs,
                                                  RayState<ITEMS> &st, float sums[5]) {
    const int l = lane_id();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int s = l * ITEMS + i;
        const float4 t = reinterpret_cast<const float4 *>(terms)[s];
        st.valid[i] = true;
        st.z[i] = zs[s];
        st.e[i] = t.x;
        st.c[i][0] = t.y; st.c[i][1] = t.z; st.c[i][2] = t.w;
    }
    ray_finish<ITEMS>(st, sums);
}

__device__ __forceinline__ float dir_norm(const float *__restrict__ row) {
    const float x = row[3], y = row[4], zc = row[5];
    return sqrtf((x * x + y * y) + zc * zc);
}


// The forward's outputs of one ray from the state ray_forward left (lane 0 writes the per-ray values): what
// composite_fwd_kernel writes, as a function so that a fused kernel produces the same bits.
template <int ITEMS>
__device__ __forceinline__ void composite_store(const RayState<ITEMS> &st, const float sums[5], int64_t ray, int S, int flags,
                                                float *__restrict__ rgb, float *__restrict__ disp, float *__restrict__ acc,
                                                float *__restrict__ depth, float *__restrict__ weights,
                                                float *__restrict__ alpha) {
    const int l = lane_id();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int s = l * ITEMS + i;
        if (st.valid[i]) {
            if (weights) weights[ray * S + s] = st.w[i];
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
        if (depth) depth[ray] = d;
    }
}

}  // namespace mvip
