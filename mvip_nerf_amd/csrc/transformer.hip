// Token-side kernels of the SD UNet's transformer blocks (unet(...) inside the SDS step,
// DS_NeRF/guidance/sd_utils.py:390-403; BasicTransformerBlock = LayerNorm -> self-attention -> LayerNorm ->
// cross-attention -> LayerNorm -> GEGLU feed-forward, from the published SD-1.5 architecture), for activations
// kept CHANNEL-MAJOR [N][C][LP] (LP = tokens padded to a multiple of 256) through the whole block: every linear
// layer is then Y = W X on the split-precision GEMM of csrc/conv3x3.hip with no layout transposes, and these
// kernels produce its operands:
//   layernorm_split : per-token LayerNorm over the channel (strided) axis, written straight as fp16 hi/lo
//                     split planes [N][C/16][2][2][LP][8] (the GEMM's B operand); two launches, both parallel over
//                     (token, channel segment): fp64 partial moments, then normalise + split;
//   geglu           : a * gelu(g) of the feed-forward's first projection + the absolute maximum that sizes the
//                     power-of-two scale of its split;
//   linear_small    : y = W act(x) + b for a handful of rows (the timestep embedding MLP and the per-ResNet-block
//                     time projections, x of shape [2, 1280]) -- one wavefront per output feature, weights streamed
//                     once, exact fp32.
#include "common.h"
#include <stdlib.h>

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

// LayerNorm statistics, pass 1: grid (LP / 64, C / 64, N), 256 threads: lane = token, wave = one 16-channel chunk of
// the workgroup's 64-channel segment.  Per (sample, segment, token): sum and sum of squares in fp64 (exact enough
// that var = E[x^2] - mean^2 carries no cancellation error at fp32 output precision).
__global__ void __launch_bounds__(256)
ln_stats_kernel(const float *__restrict__ x, int C, int L, int LP, double *__restrict__ part) {
    __shared__ double red[2][4][64];
    const int tok = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int seg = blockIdx.y, n = blockIdx.z, S = C / 64;
    const int p = blockIdx.x * 64 + tok;
    double s = 0.0, q = 0.0;
    if (p < L) {
        const float *xp = x + ((int64_t)n * C + seg * 64 + w * 16) * LP + p;
        float v[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = xp[(int64_t)c * LP];
#pragma unroll
        for (int c = 0; c < 16; ++c) { s += (double)v[c]; q += (double)v[c] * (double)v[c]; }
    }
    red[0][w][tok] = s;
    red[1][w][tok] = q;
    __syncthreads();
    if (w < 2) {                         // wave 0 writes the sums, wave 1 the squares
        const double t = red[w][0][tok] + red[w][1][tok] + red[w][2][tok] + red[w][3][tok];
        part[(((int64_t)n * S + seg) * 2 + w) * LP + p] = t;
    }
}

// pass 2: grid (LP / 256, N * C / 16): thread = (token, 16-channel chunk) -> normalise, scale, split, write planes.
__global__ void __launch_bounds__(256)
ln_apply_kernel(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                const double *__restrict__ part, int C, int L, int LP, float eps, float out_scale,
                uint4 *__restrict__ xs, int prec, int S) {
    // S: segments of the partial statistics (C / 64 from ln_stats_kernel, the producing GEMM's row blocks otherwise)
    const int CK = C / 16;
    const int ck = blockIdx.y % CK, n = blockIdx.y / CK;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= LP) return;
    float v[16];
    if (p < L) {
        double s = 0.0, q = 0.0;
        for (int g = 0; g < S; ++g) {
            s += part[(((int64_t)n * S + g) * 2 + 0) * LP + p];
            q += part[(((int64_t)n * S + g) * 2 + 1) * LP + p];
        }
        const double mean = s / C;
        double var = q / C - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const float mf = (float)mean, rstd = (float)(1.0 / sqrt(var + (double)eps));
        const float *xp = x + ((int64_t)n * C + ck * 16) * LP + p;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const int ch = ck * 16 + c;
            const float g = gamma ? gamma[ch] : 1.f, b = beta ? beta[ch] : 0.f;
            v[c] = ((xp[(int64_t)c * LP] - mf) * rstd * g + b) * out_scale;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = 0.f;
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
        if (prec != 1) dst[(int64_t)(kg * 2 + 1) * LP] = lo;        // fp16 mode: hi planes only
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

// scale2 = {2^k, 2^-k} from the maximum collected in *bits; *bits is left ZERO for its next user (the scratch word is
// owned by the caller and travels zero between calls, which saves a zeroing launch per use)
__global__ void scale_from_bits_kernel(float *__restrict__ scale2, unsigned *__restrict__ bits) {
    const float m = __uint_as_float(*bits);
    *bits = 0u;
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

// The same product for SEVERAL layers that share the input (the 22 ResNet time projections of the UNet all read
// silu(temb)): W / b are the layers' rows one after the other, `off` the layers' first rows (nl + 1 entries, device), and
// layer l's result is written as its own contiguous [NB][C_l] block at y + off[l] * NB.  One launch instead of 22.
template <int NB>
__global__ void __launch_bounds__(256)
linear_small_grouped_kernel(const float *__restrict__ x, const float *__restrict__ W, const float *__restrict__ b, int64_t M, int K,
                            int act_in, const int *__restrict__ off, int nl, float *__restrict__ y) {
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
    int l = 0;
    while (l + 1 < nl && m >= off[l + 1]) ++l;                  // <= 32 layers: a linear walk over scalar loads
    const int64_t o0 = off[l], cl = off[l + 1] - o0;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        float v = acc[i];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) y[o0 * NB + (int64_t)i * cl + (m - o0)] = v + (b ? b[m] : 0.f);
    }
}

}  // namespace tok
}  // namespace mvip

using namespace mvip;
using namespace mvip::tok;

extern "C" int64_t mvip_layernorm_workspace_bytes(int64_t N, int64_t C, int64_t LP) {
    return (N <= 0 || C <= 0 || LP <= 0) ? 0 : N * (C / 64) * 2 * LP * (int64_t)sizeof(double);
}

extern "C" int mvip_layernorm_split_planes(const float *x, const float *gamma, const float *beta, int64_t N, int64_t C,
                                           int64_t L, int64_t LP, float eps, float out_scale, void *workspace, void *xs,
                                           int prec, void *stream) {
    if ((prec < 0 || prec > 2) || N < 0 || C <= 0 || C % 64 != 0 || L <= 0 || LP < L || LP % 256 != 0 || N * (C / 16) > 65535) return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    if (!x || !xs || !workspace) return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
    // (A fused statistics + apply kernel -- one workgroup per 64 tokens, all channels, two passes -- was built in round 4 and is
    // SLOWER: 48 launches 0.70 ms against 0.21 + 0.38 ms for this pair, whose apply pass spreads over 5x more workgroups; the step
    // 22.34 vs 22.09 ms on the same box.  Removed.)
    hipLaunchKernelGGL(ln_stats_kernel, dim3((unsigned)(LP / 64), (unsigned)(C / 64), (unsigned)N), dim3(256), 0, st, x,
                       (int)C, (int)L, (int)LP, (double *)workspace);
    hipLaunchKernelGGL(ln_apply_kernel, dim3((unsigned)(LP / 256), (unsigned)(N * (C / 16))), dim3(256), 0, st, x, gamma,
                       beta, (const double *)workspace, (int)C, (int)L, (int)LP, eps, out_scale, (uint4 *)xs, prec, (int)(C / 64));
    return check_launch();
}

// The apply half alone, on statistics somebody else left: `part` = [N][segments][2][LP] fp64 sums / sums of squares over
// `segments` disjoint channel ranges that cover C (mvip_gemm_f16x3_ws_ln's output): one launch instead of two, no extra
// pass over x.  Same arithmetic from the statistics on (fp64 mean / variance, then fp32).
extern "C" int mvip_layernorm_split_planes_stats(const float *x, const float *gamma, const float *beta, const void *part,
                                                 int64_t segments, int64_t N, int64_t C, int64_t L, int64_t LP, float eps,
                                                 float out_scale, void *xs, int prec, void *stream) {
    if ((prec < 0 || prec > 2) || N < 0 || C <= 0 || C % 16 != 0 || L <= 0 || LP < L || LP % 256 != 0 || N * (C / 16) > 65535 ||
        segments < 1 || segments > C)
        return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    if (!x || !xs || !part) return MVIP_EINVAL;
    hipLaunchKernelGGL(ln_apply_kernel, dim3((unsigned)(LP / 256), (unsigned)(N * (C / 16))), dim3(256), 0, as_stream(stream), x, gamma,
                       beta, (const double *)part, (int)C, (int)L, (int)LP, eps, out_scale, (uint4 *)xs, prec, (int)segments);
    return check_launch();
}

extern "C" int mvip_geglu(const float *y, int64_t N, int64_t R, int64_t L, int64_t LP, float *out, float *scale2,
                          void *zero_word, void *stream) {
    if (N < 0 || R <= 0 || L <= 0 || LP < L || LP % 4 != 0 || !scale2 || !zero_word) return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
    if (N > 0) {
        if (!y || !out) return MVIP_EINVAL;
        const int64_t total4 = N * R * (LP / 4);
        int64_t blocks = (total4 + 255) / 256;
        blocks = blocks > 4096 ? 4096 : blocks;
        hipLaunchKernelGGL(geglu_kernel, dim3((unsigned)blocks), dim3(256), 0, st, y, R, (int)L, (int)LP, out,
                           (unsigned *)zero_word, total4);
    }
    hipLaunchKernelGGL(scale_from_bits_kernel, dim3(1), dim3(1), 0, st, scale2, (unsigned *)zero_word);
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

extern "C" int mvip_linear_small_grouped(const float *x, const float *W, const float *b, int64_t NB, int64_t M, int64_t K,
                                         int act_in, const int *layer_off, int64_t n_layers, float *y, void *stream) {
    if (NB <= 0 || NB > 8 || M <= 0 || K <= 0 || K > 0x7fffffff || n_layers <= 0 || n_layers > 64 || !x || !W || !y || !layer_off)
        return MVIP_EINVAL;
    const dim3 grid((unsigned)((M + 3) / 4));
    hipStream_t st = as_stream(stream);
    switch (NB) {
#define MVIP_LS(n) case n: hipLaunchKernelGGL((linear_small_grouped_kernel<n>), grid, dim3(256), 0, st, x, W, b, M, (int)K, act_in, layer_off, (int)n_layers, y); break;
        MVIP_LS(1) MVIP_LS(2) MVIP_LS(3) MVIP_LS(4) MVIP_LS(5) MVIP_LS(6) MVIP_LS(7) MVIP_LS(8)
#undef MVIP_LS
    }
    return check_launch();
}
