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

}  // namespace mvip

using namespace mvip;

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
