// Token-side kernels of the SD UNet's transformer blocks (unet(...) inside the SDS step,
// DS_NeRF/guidance/sd_utils.py:390-403; BasicTransformerBlock = LayerNorm -> self-attention -> LayerNorm ->
// cross-attention -> LayerNorm -> GEGLU feed-forward, from the published SD-1.5 architecture), for activations
// kept CHANNEL-MAJOR [N][C][LP] (LP = tokens padded to a multiple of 256) through the whole block: every linear
// layer is then Y = W X on the split-precision GEMM of csrc/conv3x3.hip with no layout transposes, and these
// kernels produce its operands:
//   layernorm_split : per-token LayerNorm over the channel (strided) axis, written straight as fp16 hi/lo
//                     split planes [N][C/16][2][2][LP][8] (the GEMM's B operand);
//   geglu           : a * gelu(g) of the feed-forward's first projection + the absolute maximum that sizes the
//                     power-of-two scale of its split;
//   linear_small    : y = W act(x) + b for a handful of rows (the timestep embedding MLP and the per-ResNet-block
//                     time projections, x of shape [2, 1280]) -- one wavefront per output feature, weights streamed
//                     once, exact fp32.
#include "common.h"

namespace mvip {
namespace tok {

__device__ __forceinline__ void split8(const float (&v)[8], uint4 &hi, uint4 &lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h2 a, b;
        a.x = (_Float16)v[2 * i]; a.y = (_Float16)v[2 * i + 1];
        b.x = (_Float16)(v[2 * i] - (float)a.x); b.y = (_Float16)(v[2 * i + 1] - (float)a.y);
        h[i] = __builtin_bit_cast(unsigned, a); l[i] = __builtin_bit_cast(unsigned, b);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// grid (LP / 64, N), 256 threads: lane = token, wave = channel group (chunks wave, wave + 4, ... of 16 channels).
// Two-pass statistics in fp32 (mean, then centred second moment), the three passes re-read x from L2.
__global__ void __launch_bounds__(256)
layernorm_split_kernel(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                       int C, int L, int LP, float eps, float out_scale, uint4 *__restrict__ xs) {
    __shared__ float red[4][64];
    const int tok = threadIdx.x & 63;
    const int cg = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = blockIdx.y;
    const int p = blockIdx.x * 64 + tok;
    const int CK = C / 16;
    const bool live = p < L;
    const float *xp = x + (int64_t)n * C * LP + p;
    float s = 0.f;
    if (live)
        for (int ck = cg; ck < CK; ck += 4)
#pragma unroll
            for (int c = 0; c < 16; ++c) s += xp[(int64_t)(ck * 16 + c) * LP];
    red[cg][tok] = s;
    __syncthreads();
    const float mean = (red[0][tok] + red[1][tok] + red[2][tok] + red[3][tok]) / (float)C;
    __syncthreads();
    float q = 0.f;
    if (live)
        for (int ck = cg; ck < CK; ck += 4)
#pragma unroll
            for (int c = 0; c < 16; ++c) { const float d = xp[(int64_t)(ck * 16 + c) * LP] - mean; q += d * d; }
    red[cg][tok] = q;
    __syncthreads();
    const float var = (red[0][tok] + red[1][tok] + red[2][tok] + red[3][tok]) / (float)C;
    const float rstd = 1.0f / sqrtf(var + eps);
    for (int ck = cg; ck < CK; ck += 4) {
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int ch = ck * 16 + c;
            const float g = gamma ? gamma[ch] : 1.f, b = beta ? beta[ch] : 0.f;
            v[c] = live ? ((xp[(int64_t)ch * LP] - mean) * rstd * g + b) * out_scale : 0.f;
        }
        uint4 *dst = xs + ((int64_t)(n * CK + ck) * 4) * LP + p;
#pragma unroll
        for (int kg = 0; kg < 2; ++kg) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = v[kg * 8 + j];
            uint4 hi, lo;
            split8(t, hi, lo);
            dst[(int64_t)(kg * 2 + 0) * LP] = hi;
            dst[(int64_t)(kg * 2 + 1) * LP] = lo;
        }
    }
}

// y [N][2R][LP] -> out [N][R][LP] = y[:, r] * gelu(y[:, R + r]) (erf form, torch's default), zero for p >= L;
// the absolute maximum of the result is collected into bits (one atomic per workgroup).
__global__ void __launch_bounds__(256)
geglu_kernel(const float *__restrict__ y, int64_t R, int L, int LP, float *__restrict__ out, unsigned *__restrict__ bits,
             int64_t total4) {
    float m = 0.f;
    const int LP4 = LP >> 2;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (int64_t)gridDim.x * blockDim.x) {
        const int p4 = (int)(i % LP4);
        const int64_t row = i / LP4;                 // n * R + r
        const int64_t n = row / R, r = row - n * R;
        const float4 a = reinterpret_cast<const float4 *>(y + ((n * 2 * R + r) * LP))[p4];
        const float4 g = reinterpret_cast<const float4 *>(y + ((n * 2 * R + R + r) * LP))[p4];
        auto f = [&](float av, float gv, int k) {
            const float v = (p4 * 4 + k < L) ? av * (0.5f * gv * (1.0f + erff(gv * 0.70710678118654752f))) : 0.f;
            const float w = fabsf(v);
            m = (w == w && w < 3.0e38f) ? fmaxf(m, w) : m;
            return v;
        };
        float4 o;
        o.x = f(a.x, g.x, 0); o.y = f(a.y, g.y, 1); o.z = f(a.z, g.z, 2); o.w = f(a.w, g.w, 3);
        reinterpret_cast<float4 *>(out + row * LP)[p4] = o;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(bits, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

__global__ void scale_from_bits_kernel(float *__restrict__ scale2) {
    const float m = __uint_as_float(reinterpret_cast<const unsigned *>(scale2)[2]);
    float sc = 1.f;
    if (m > 0.f && m < 3.0e38f) {
        int e;
        frexpf(m, &e);
        int k = 10 - e;
        k = k > 60 ? 60 : (k < -60 ? -60 : k);
        sc = ldexpf(1.f, k);
    }
    scale2[0] = sc;
    scale2[1] = 1.f / sc;
}

// y[nb][m] = sum_k W[m][k] act(x[nb][k]) + b[m];  one wavefront per output feature m, NB <= 8 input rows.
template <int NB>
__global__ void __launch_bounds__(256)
linear_small_kernel(const float *__restrict__ x, const float *__restrict__ W, const float *__restrict__ b, int64_t M, int K,
                    int act_in, float *__restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    const float *w = W + m * K;
    float acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) acc[i] = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float wv = w[k];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            float xv = x[(int64_t)i * K + k];
            if (act_in == 1) xv = xv / (1.0f + expf(-xv));
            acc[i] = fmaf(wv, xv, acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        float v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) y[(int64_t)i * M + m] = v + (b ? b[m] : 0.f);
    }
}

}  // namespace tok
}  // namespace mvip

using namespace mvip;
using namespace mvip::tok;

extern "C" int mvip_layernorm_split_planes(const float *x, const float *gamma, const float *beta, int64_t N, int64_t C,
                                           int64_t L, int64_t LP, float eps, float out_scale, void *xs, void *stream) {
    if (N < 0 || C <= 0 || C % 64 != 0 || L <= 0 || LP < L || LP % 64 != 0 || N > 65535) return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    if (!x || !xs) return MVIP_EINVAL;
    hipLaunchKernelGGL(layernorm_split_kernel, dim3((unsigned)(LP / 64), (unsigned)N), dim3(256), 0, as_stream(stream), x,
                       gamma, beta, (int)C, (int)L, (int)LP, eps, out_scale, (uint4 *)xs);
    return check_launch();
}

extern "C" int mvip_geglu(const float *y, int64_t N, int64_t R, int64_t L, int64_t LP, float *out, float *scale2,
                          void *stream) {
    if (N < 0 || R <= 0 || L <= 0 || LP < L || LP % 4 != 0 || !scale2) return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
    zero_words(scale2, 4, st);
    if (N > 0) {
        if (!y || !out) return MVIP_EINVAL;
        const int64_t total4 = N * R * (LP / 4);
        int64_t blocks = (total4 + 255) / 256;
        blocks = blocks > 4096 ? 4096 : blocks;
        hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)blocks), dim3(256), 0, st, y, R, (int)L, (int)LP, out,
                           (unsigned *)(scale2 + 2), total4);
    }
    hipLaunchKernelGGL(scale_from_bits_kernel, dim3(1), dim3(1), 0, st, scale2);
    return check_launch();
}

extern "C" int mvip_linear_small(const float *x, const float *W, const float *b, int64_t NB, int64_t M, int64_t K,
                                 int act_in, float *y, void *stream) {
    if (NB <= 0 || NB > 8 || M <= 0 || K <= 0 || K > 0x7fffffff || !x || !W || !y) return MVIP_EINVAL;
    const dim3 grid((unsigned)((M + 3) / 4));
    hipStream_t st = as_stream(stream);
    switch (NB) {
#define MVIP_LS(n) case n: hipLaunchKernelGGL((linear_small_kernel<n>), grid, dim3(256), 0, st, x, W, b, M, (int)K, act_in, y); break;
        MVIP_LS(1) MVIP_LS(2) MVIP_LS(3) MVIP_LS(4) MVIP_LS(5) MVIP_LS(6) MVIP_LS(7) MVIP_LS(8)
#undef MVIP_LS
    }
    return check_launch();
}
