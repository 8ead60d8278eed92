// Elementwise core of the score-distillation step (DS_NeRF/guidance/sd_utils.py:406-413 and the
// scheduler.add_noise the pipeline's prepare_latents applies, pipeline_sd_inpainting.py:668).
// 16 K-element tensors: launch-bound, so each is ONE fused kernel instead of ~8 torch ops.
#include "common.h"
#include <float.h>

namespace mvip {

__global__ void sds_add_noise_kernel(const float *__restrict__ x0, const float *__restrict__ noise, float sa,
                                     float sb, int64_t n, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sa * x0[i] + sb * noise[i];
}

__device__ __forceinline__ float nan_to_num(float v) {
    if (v != v) return 0.f;
    if (v > FLT_MAX) return FLT_MAX;
    if (v < -FLT_MAX) return -FLT_MAX;
    return v;
}

// eps = e_u + s*(e_c - e_u);  grad (+)= w*(eps - noise);  grad = nan_to_num(grad)
__global__ void sds_grad_kernel(const float *__restrict__ eu, const float *__restrict__ ec,
                                const float *__restrict__ noise, float s, float w, int64_t n, int accumulate,
                                float *__restrict__ grad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float u = eu[i];
    const float e = ec ? u + s * (ec[i] - u) : u;
    float g = w * (e - noise[i]);
    if (accumulate) g = grad[i] + g;
    grad[i] = nan_to_num(g);
}

// Variants that read the timestep-dependent scalars from device memory, so a captured hipGraph of the
// whole SDS step can be replayed for any t: scal = {sqrt(abar), sqrt(1-abar), 1-abar}.
__global__ void sds_add_noise_dev_kernel(const float *__restrict__ x0, const float *__restrict__ noise,
                                         const float *__restrict__ scal, int64_t n, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = scal[0] * x0[i] + scal[1] * noise[i];
}

__global__ void sds_grad_dev_kernel(const float *__restrict__ eu, const float *__restrict__ ec,
                                    const float *__restrict__ noise, float s, const float *__restrict__ scal,
                                    int64_t n, int accumulate, float *__restrict__ grad) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float u = eu[i];
    const float e = ec ? u + s * (ec[i] - u) : u;
    float g = scal[2] * (e - noise[i]);
    if (accumulate) g = grad[i] + g;
    grad[i] = nan_to_num(g);
}

// ---- posterior sample of the VAE encoder (pipeline _encode_vae_image: scaling_factor * latent_dist.sample(),
// DS_NeRF/guidance/sd_utils.py:207 through the inpainting pipeline) and its adjoint, one launch each instead of the seven
// (forward) / fifteen (autograd backward: clamp mask, exp, products, the zero-filled halves of chunk's cat) elementwise
// launches:  moments [N][2C][HW] = (mean | logvar);  lv = clamp(logvar, -30, 20);  std = exp(0.5 lv);
//   out = sf * (mean + std * noise)            d_mean = sf g,   d_logvar = 0.5 (sf g noise) std inside the clamp, else 0
__global__ void vae_sample_kernel(const float *__restrict__ moments, const float *__restrict__ noise, float sf, int64_t n,
                                  int64_t chw, float *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t b = i / chw, r = i - b * chw;
    const float mean = moments[b * 2 * chw + r];
    float lv = moments[b * 2 * chw + chw + r];
    lv = fminf(fmaxf(lv, -30.f), 20.f);
    const float sd = expf(0.5f * lv);
    out[i] = __fmul_rn(sf, __fadd_rn(mean, __fmul_rn(sd, noise[i])));       // torch's three roundings (no fused multiply-add)
}

__global__ void vae_sample_bwd_kernel(const float *__restrict__ moments, const float *__restrict__ noise,
                                      const float *__restrict__ g, float sf, int64_t n, int64_t chw,
                                      float *__restrict__ d_moments) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t b = i / chw, r = i - b * chw;
    const float lv = moments[b * 2 * chw + chw + r];
    const bool inside = lv >= -30.f && lv <= 20.f;
    const float sd = expf(0.5f * fminf(fmaxf(lv, -30.f), 20.f));
    const float gs = __fmul_rn(g[i], sf);
    d_moments[b * 2 * chw + r] = gs;
    d_moments[b * 2 * chw + chw + r] = inside ? __fmul_rn(__fmul_rn(__fmul_rn(gs, noise[i]), sd), 0.5f) : 0.f;
}

// ---- sinusoidal timestep embedding (diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0) in
// front of the UNet's time MLP): out[n] = (cos(t_n f_k) | sin(t_n f_k)) for the half = dim / 2 frequencies f (computed
// once by the caller), one launch instead of product + cos + sin + cat
__global__ void timestep_sincos_kernel(const float *__restrict__ t, const float *__restrict__ freqs, int N, int half,
                                       float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * half) return;
    const int n = i / half, k = i - n * half;
    const float a = __fmul_rn(t[n], freqs[k]);
    out[(int64_t)n * 2 * half + k] = cosf(a);
    out[(int64_t)n * 2 * half + half + k] = sinf(a);
}

// ---- bilinear resize, align_corners = False (the F.interpolate in front of vae.encode, DS_NeRF/guidance/sd_utils.py:282-284)
// torch's convention: scale = in / out (float), src = scale * (dst + 0.5) - 0.5 clamped at 0, i0 = (int)src,
// i1 = i0 + (i0 < in - 1), lambda1 = src - i0, lambda0 = 1 - lambda1.
struct Tap { int i0, i1; float l0, l1; };
__device__ __forceinline__ Tap resize_tap(int dst, float scale, int in) {
    float src = scale * ((float)dst + 0.5f) - 0.5f;
    if (src < 0.f) src = 0.f;
    Tap t;
    t.i0 = (int)src;
    if (t.i0 > in - 1) t.i0 = in - 1;
    t.i1 = t.i0 + (t.i0 < in - 1 ? 1 : 0);
    t.l1 = src - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}

__global__ void resize_bilinear_fwd_kernel(const float *__restrict__ x, int64_t planes, int H, int W, int OH, int OW,
                                           float *__restrict__ y) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= planes * OH * OW) return;
    const int ox = (int)(idx % OW), oy = (int)((idx / OW) % OH);
    const int64_t pl = idx / ((int64_t)OW * OH);
    const Tap ty = resize_tap(oy, (float)H / (float)OH, H), tx = resize_tap(ox, (float)W / (float)OW, W);
    const float *xp = x + pl * H * W;
    const float top = tx.l0 * xp[(int64_t)ty.i0 * W + tx.i0] + tx.l1 * xp[(int64_t)ty.i0 * W + tx.i1];
    const float bot = tx.l0 * xp[(int64_t)ty.i1 * W + tx.i0] + tx.l1 * xp[(int64_t)ty.i1 * W + tx.i1];
    y[idx] = ty.l0 * top + ty.l1 * bot;
}

// Adjoint as a GATHER (deterministic; torch scatters with atomics): input pixel (iy, ix) collects every output pixel
// whose taps touch it.  The candidate range of output rows / columns comes from inverting src(dst) with a margin and
// every candidate is then tested with the forward's own tap computation.
__device__ __forceinline__ void resize_range(int i, float scale, int out, int &lo, int &hi) {
    const float inv = 1.f / scale;
    lo = (int)floorf(((float)i - 1.0f + 0.5f) * inv - 0.5f) - 1;
    hi = (int)ceilf(((float)i + 1.0f + 0.5f) * inv - 0.5f) + 1;
    if (i == 0) lo = 0;                                  // src is clamped at 0: every dst below maps here
    if (lo < 0) lo = 0;
    if (hi > out - 1) hi = out - 1;
}

__global__ void resize_bilinear_bwd_kernel(const float *__restrict__ dy, int64_t planes, int H, int W, int OH, int OW,
                                           float *__restrict__ dx) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= planes * H * W) return;
    const int ix = (int)(idx % W), iy = (int)((idx / W) % H);
    const int64_t pl = idx / ((int64_t)W * H);
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    int y0, y1, x0, x1;
    resize_range(iy, sy, OH, y0, y1);
    resize_range(ix, sx, OW, x0, x1);
    const float *dp = dy + pl * OH * OW;
    float acc = 0.f;
    for (int oy = y0; oy <= y1; ++oy) {
        const Tap ty = resize_tap(oy, sy, H);
        float wy = 0.f;
        if (ty.i0 == iy) wy += ty.l0;
        if (ty.i1 == iy) wy += ty.l1;
        if (wy == 0.f) continue;
        float row = 0.f;
        for (int ox = x0; ox <= x1; ++ox) {
            const Tap tx = resize_tap(ox, sx, W);
            float wx = 0.f;
            if (tx.i0 == ix) wx += tx.l0;
            if (tx.i1 == ix) wx += tx.l1;
            if (wx != 0.f) row += wx * dp[(int64_t)oy * OW + ox];
        }
        acc += wy * row;
    }
    dx[idx] = acc;
}

}  // namespace mvip

using namespace mvip;

extern "C" int mvip_resize_bilinear(const float *x, int64_t planes, int64_t H, int64_t W, int64_t OH, int64_t OW, float *y,
                                    void *stream) {
    if (planes < 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || H > (1 << 20) || W > (1 << 20) || OH > (1 << 20) ||
        OW > (1 << 20))
        return MVIP_EINVAL;
    if (planes == 0) return MVIP_OK;
    if (!x || !y) return MVIP_EINVAL;
    const int64_t n = planes * OH * OW;
    hipLaunchKernelGGL(resize_bilinear_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), x,
                       planes, (int)H, (int)W, (int)OH, (int)OW, y);
    return check_launch();
}

extern "C" int mvip_resize_bilinear_backward(const float *dy, int64_t planes, int64_t H, int64_t W, int64_t OH, int64_t OW,
                                             float *dx, void *stream) {
    if (planes < 0 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || H > (1 << 20) || W > (1 << 20) || OH > (1 << 20) ||
        OW > (1 << 20))
        return MVIP_EINVAL;
    if (planes == 0) return MVIP_OK;
    if (!dy || !dx) return MVIP_EINVAL;
    const int64_t n = planes * H * W;
    hipLaunchKernelGGL(resize_bilinear_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), dy,
                       planes, (int)H, (int)W, (int)OH, (int)OW, dx);
    return check_launch();
}

extern "C" int mvip_sds_add_noise_dev(const float *x0, const float *noise, const float *scal, int64_t n,
                                      float *latents, void *stream) {
    if (n < 0 || (n > 0 && (!x0 || !noise || !scal || !latents))) return MVIP_EINVAL;
    if (n == 0) return MVIP_OK;
    hipLaunchKernelGGL(sds_add_noise_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       x0, noise, scal, n, latents);
    return check_launch();
}

extern "C" int mvip_sds_grad_dev(const float *eps_uncond, const float *eps_cond, const float *noise,
                                 float guidance_scale, const float *scal, int64_t n, int accumulate, float *grad,
                                 void *stream) {
    if (n < 0 || (n > 0 && (!eps_uncond || !noise || !scal || !grad))) return MVIP_EINVAL;
    if (n == 0) return MVIP_OK;
    hipLaunchKernelGGL(sds_grad_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       eps_uncond, eps_cond, noise, guidance_scale, scal, n, accumulate, grad);
    return check_launch();
}


extern "C" int mvip_vae_sample(const float *moments, const float *noise, float scaling_factor, int64_t N, int64_t C,
                               int64_t HW, float *out, void *stream) {
    if (N < 0 || C <= 0 || HW < 0) return MVIP_EINVAL;
    const int64_t n = N * C * HW;
    if (n == 0) return MVIP_OK;
    if (!moments || !noise || !out) return MVIP_EINVAL;
    hipLaunchKernelGGL(vae_sample_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), moments, noise,
                       scaling_factor, n, C * HW, out);
    return check_launch();
}

extern "C" int mvip_vae_sample_backward(const float *moments, const float *noise, const float *d_out, float scaling_factor,
                                        int64_t N, int64_t C, int64_t HW, float *d_moments, void *stream) {
    if (N < 0 || C <= 0 || HW < 0) return MVIP_EINVAL;
    const int64_t n = N * C * HW;
    if (n == 0) return MVIP_OK;
    if (!moments || !noise || !d_out || !d_moments) return MVIP_EINVAL;
    hipLaunchKernelGGL(vae_sample_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream), moments, noise,
                       d_out, scaling_factor, n, C * HW, d_moments);
    return check_launch();
}

extern "C" int mvip_timestep_sincos(const float *t, const float *freqs, int64_t N, int64_t half, float *out, void *stream) {
    if (N < 0 || half <= 0 || N * half > (1 << 30)) return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    if (!t || !freqs || !out) return MVIP_EINVAL;
    hipLaunchKernelGGL(timestep_sincos_kernel, dim3((unsigned)((N * half + 255) / 256)), dim3(256), 0, as_stream(stream), t, freqs,
                       (int)N, (int)half, out);
    return check_launch();
}


extern "C" int mvip_sds_add_noise(const float *x0, const float *noise, float sqrt_abar, float sqrt_1m_abar,
                                  int64_t n, float *latents, void *stream) {
    if (n < 0 || (n > 0 && (!x0 || !noise || !latents))) return MVIP_EINVAL;
    if (n == 0) return MVIP_OK;
    hipLaunchKernelGGL(sds_add_noise_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       x0, noise, sqrt_abar, sqrt_1m_abar, n, latents);
    return check_launch();
}

extern "C" int mvip_sds_grad(const float *eps_uncond, const float *eps_cond, const float *noise,
                             float guidance_scale, float w, int64_t n, int accumulate, float *grad,
                             void *stream) {
    if (n < 0 || (n > 0 && (!eps_uncond || !noise || !grad))) return MVIP_EINVAL;
    if (n == 0) return MVIP_OK;
    hipLaunchKernelGGL(sds_grad_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, as_stream(stream),
                       eps_uncond, eps_cond, noise, guidance_scale, w, n, accumulate, grad);
    return check_launch();
}

// ---- row softmax of the VAE mid-block attention (AutoencoderKL's single-head attention over 64 x 64 tokens inside
// vae.encode, DS_NeRF/guidance/sd_utils.py:207): P[i][:] = softmax(scale * S[i][:]) and its adjoint
// dS[i][j] = scale * P[i][j] * (dP[i][j] - sum_k dP[i][k] P[i][k]).  One workgroup per row, the row held in registers
// (cols <= 256 * 32); maxima / sums by wave shuffles + one LDS exchange, in a fixed order (bit-reproducible).
namespace mvip {

__device__ __forceinline__ float sm_block_reduce(float v, bool is_max, float *sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(v, o, 64); v = is_max ? fmaxf(v, t) : v + t; }
    const int w = threadIdx.x >> 6;
    __syncthreads();                                 // sh may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float r = sh[0];
#pragma unroll
    for (int k = 1; k < 4; ++k) r = is_max ? fmaxf(r, sh[k]) : r + sh[k];
    return r;
}

constexpr int SM_MAX_PER_THREAD = 32;

__global__ void __launch_bounds__(256) softmax_rows_kernel(const float *__restrict__ s, int64_t cols, float scale,
                                                           float *__restrict__ p) {
    __shared__ float sh[4];
    const float *sr = s + (int64_t)blockIdx.x * cols;
    float *pr = p + (int64_t)blockIdx.x * cols;
    float v[SM_MAX_PER_THREAD];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < SM_MAX_PER_THREAD; ++k) {
        const int64_t c = (int64_t)k * 256 + threadIdx.x;
        v[k] = c < cols ? sr[c] * scale : -INFINITY;
        mx = fmaxf(mx, v[k]);
    }
    mx = sm_block_reduce(mx, true, sh);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < SM_MAX_PER_THREAD; ++k) { v[k] = expf(v[k] - mx); sum += v[k]; }     // exp(-inf) = 0 beyond cols
    sum = sm_block_reduce(sum, false, sh);
    const float inv = 1.f / sum;
#pragma unroll
    for (int k = 0; k < SM_MAX_PER_THREAD; ++k) {
        const int64_t c = (int64_t)k * 256 + threadIdx.x;
        if (c < cols) pr[c] = v[k] * inv;
    }
}

__global__ void __launch_bounds__(256) softmax_rows_backward_kernel(const float *__restrict__ p, const float *__restrict__ dp,
                                                                    int64_t cols, float scale, float *__restrict__ ds) {
    __shared__ float sh[4];
    const int64_t r0 = (int64_t)blockIdx.x * cols;
    float pv[SM_MAX_PER_THREAD], dv[SM_MAX_PER_THREAD];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < SM_MAX_PER_THREAD; ++k) {
        const int64_t c = (int64_t)k * 256 + threadIdx.x;
        pv[k] = c < cols ? p[r0 + c] : 0.f;
        dv[k] = c < cols ? dp[r0 + c] : 0.f;
        dot += pv[k] * dv[k];
    }
    dot = sm_block_reduce(dot, false, sh);
#pragma unroll
    for (int k = 0; k < SM_MAX_PER_THREAD; ++k) {
        const int64_t c = (int64_t)k * 256 + threadIdx.x;
        if (c < cols) ds[r0 + c] = pv[k] * (dv[k] - dot) * scale;
    }
}

}  // namespace mvip

extern "C" int mvip_softmax_rows(const float *s, int64_t rows, int64_t cols, float scale, float *p, void *stream) {
    if (rows < 0 || cols <= 0 || cols > 256 * mvip::SM_MAX_PER_THREAD || rows > 0x7fffffffLL) return MVIP_EINVAL;
    if (rows == 0) return MVIP_OK;
    if (!s || !p) return MVIP_EINVAL;
    hipLaunchKernelGGL(mvip::softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, mvip::as_stream(stream), s, cols, scale, p);
    return mvip::check_launch();
}

extern "C" int mvip_softmax_rows_backward(const float *p, const float *dp, int64_t rows, int64_t cols, float scale, float *ds,
                                          void *stream) {
    if (rows < 0 || cols <= 0 || cols > 256 * mvip::SM_MAX_PER_THREAD || rows > 0x7fffffffLL) return MVIP_EINVAL;
    if (rows == 0) return MVIP_OK;
    if (!p || !dp || !ds) return MVIP_EINVAL;
    hipLaunchKernelGGL(mvip::softmax_rows_backward_kernel, dim3((unsigned)rows), dim3(256), 0, mvip::as_stream(stream), p, dp, cols,
                       scale, ds);
    return mvip::check_launch();
}
