// 3x3 / stride 1 / pad 1 convolution of the SDS networks as an implicit GEMM on the fp16 matrix cores in
// split precision ("f16x3": both operands split into fp16 hi + lo, the three leading products
// Wh.Xh + Wh.Xl + Wl.Xh accumulated in fp32 -- ~1e-6 relative, i.e. fp32-grade results at ~2.7x the
// exact-fp32 MFMA rate).  These are the ResNet-block convolutions inside vae.encode / unet
// (call sites DS_NeRF/guidance/sd_utils.py:148, :162, :189, :212; block structure from the published
// SD-1.5 architecture) -- 1.0 TFLOP per VAE encode, the largest single item of the SDS step.
//
// GEMM view: M = output channels, N = pixels, K = (tap, input channel).
//   * activations arrive in "split planes": xs[n][Cin/16][kg 2][hl 2][H][W][8 halves] -- for 16-channel
//     chunk ck, plane (kg, hl) holds channels ck*16 + kg*8 + 0..7 of every pixel as the hi (hl=0) or lo
//     (hl=1) fp16 term.  16 B per (pixel, plane) = one MFMA B-operand fragment, so a haloed pixel tile goes
//     HBM -> LDS by DMA (global_load_lds) with no register staging, and the shifted windows of the nine
//     taps are plain offset reads of the same LDS tile.  The producers write this layout directly
//     (GroupNorm+SiLU apply, or the plain converter for gradients).
//   * weights are packed once per layer in MFMA A-fragment order, pre-scaled by a power of two so that
//     the lo terms stay in fp16's normal range:
//     wp[Cout/32][Cin/16][ky][kx][hl][lane 64][8 halves]; the tail holds {scale, 1/scale}.
//   * one workgroup = MT*32 output channels x (8 x 32) pixels, 4 waves, each wave MT x 2 accumulator tiles;
//     MT = 4, 2 or 1 is chosen per launch so that the grid covers the 256 CUs.
//     K loop: 16 input channels x one kernel row per stage (3 taps, 72 MFMAs per wave at MT=4), weights
//     and the input tile double-buffered in LDS, one barrier per stage.
//   * epilogue: x 1/scale, + bias, + per-(sample, channel) addend (time embedding), + residual, NCHW fp32.
//   * NP (template; `prec` of the C ABI): products per contraction step.  3 = Wh.Xh + Wh.Xl + Wl.Xh (prec 0); 1 = Wh.Xh only
//     (prec 1, the reference's --fp16 mode); 2 = Wh.Xh + Wh.Xl (prec 2) for weights that are EXACTLY fp16 values -- every
//     weight the reference ever loads is one (DS_NeRF/guidance/sd_utils.py:69-74: `revision="fp16"`, cast up in the default
//     fp32 mode), so the packed image's lo fragments are all zero, the third product adds exact zeros, and leaving it out
//     changes no bit of the result while a third of the matrix work and half of the weight-operand bytes go away.  The
//     packers record whether any lo fragment is non-zero (tail word 12; mvip_packed_weights_two_product reads it).
#include "common.h"
#include "plane_sink.h"
#include <stdlib.h>

namespace mvip {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CV_TH = 8, CV_TW = 32;
constexpr int CV_HW = CV_TW + 2;                    // haloed tile width
constexpr int CV_PIX = (CV_TH + 2) * CV_HW;         // 340 haloed pixels of the 8 x 32 tile (4 planes x 340 slots = 21,760 B)

__device__ __forceinline__ void glds16b(const void *src_lane, void *dst_wave) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src_lane,
                                     (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
}

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

// ---- power-of-two scale from the absolute maximum ----------------------------------------------------
__global__ void __launch_bounds__(256) cv_absmax_kernel(const float *__restrict__ x, int64_t n, unsigned *__restrict__ out) {
    float m = 0.f;
    auto take = [&](float v) { v = fabsf(v); m = (v == v && v < 3.0e38f) ? fmaxf(m, v) : m; };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        const int64_t n4 = n >> 2;
        // four independent 16-byte loads per thread and trip: one in flight per thread (2 MB over the whole grid) held this
        // streaming read to ~1 TB/s
        int64_t i = tid;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const float4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
            take(a.x); take(a.y); take(a.z); take(a.w); take(b.x); take(b.y); take(b.z); take(b.w);
            take(c.x); take(c.y); take(c.z); take(c.w); take(d.x); take(d.y); take(d.z); take(d.w);
        }
        for (; i < n4; i += stride) { const float4 v = x4[i]; take(v.x); take(v.y); take(v.z); take(v.w); }
        for (int64_t k = (n4 << 2) + tid; k < n; k += stride) take(x[k]);
    } else {
        for (int64_t i = tid; i < n; i += stride) take(x[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];                 // one atomic per workgroup: same-address atomics serialise (~10 ns each)
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

// One-launch form: the maxima are collected in words[0] (zero on entry), every workgroup takes a ticket in words[1]
// (zero on entry) and the LAST one turns the maximum into scale2 = {s, 1/s} and leaves both words zero again.  The two
// words are caller-owned scratch that travels zero between calls on a stream (saves the zeroing and the scale launch).
__global__ void __launch_bounds__(256) cv_absmax_scale_kernel(const float *__restrict__ x, int64_t n, unsigned *__restrict__ words,
                                                             float *__restrict__ scale2) {
    float m = 0.f;
    auto take = [&](float v) { v = fabsf(v); m = (v == v && v < 3.0e38f) ? fmaxf(m, v) : m; };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {
        const float4 *x4 = reinterpret_cast<const float4 *>(x);
        const int64_t n4 = n >> 2;
        int64_t i = tid;                               // eight independent 16-byte loads per thread and trip: these launches are
        for (; i + 7 * stride < n4; i += 8 * stride) { // latency-bound (a trip is one memory round trip), see absmax_blocks
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x4[i + u * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) { take(v[u].x); take(v[u].y); take(v[u].z); take(v[u].w); }
        }
        for (; i < n4; i += stride) { const float4 v = x4[i]; take(v.x); take(v.y); take(v.z); take(v.w); }
        for (int64_t k = (n4 << 2) + tid; k < n; k += stride) take(x[k]);
    } else {
        for (int64_t i = tid; i < n; i += stride) take(x[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(words, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
        __threadfence();
        const unsigned ticket = atomicAdd(words + 1, 1u);
        if (ticket == gridDim.x - 1) {
            const float mx = __uint_as_float(atomicExch(words, 0u));         // every workgroup's maximum has landed
            atomicExch(words + 1, 0u);
            float sc = 1.f;
            if (mx > 0.f && mx < 3.0e38f) {
                int e;
                frexpf(mx, &e);
                int k = 10 - e;
                k = k > 60 ? 60 : (k < -60 ? -60 : k);
                sc = ldexpf(1.f, k);
            }
            scale2[0] = sc;
            scale2[1] = 1.f / sc;
        }
    }
}

// one workgroup per 64 K elements, at most one per CU: every workgroup ends with two atomics on the same pair of words,
// which serialise in L2 (512 workgroups: 1.54 ms over the 94 calls of an SDS step, 256: see profiles)
static inline unsigned absmax_blocks(int64_t n) {
    // Small tensors are latency-bound (a trip of a thread's loop is one memory round trip, a workgroup's two atomics serialise
    // in L2): 64 floats per thread on at most 64 workgroups; large ones are bandwidth-bound: 64 K floats per workgroup on at
    // most 256.  Per launch inside a graph replay (tools/absmax_sweep.py, profiles/r4_absmax_sweep.json): 0.16 M floats 6.2 ->
    // 3.7 us, 0.66 M 6.5 -> 4.1, 2.6 M 6.7 -> 5.9, 33 M 25.2 (unchanged).  MVIP_ABSMAX_FPT / MVIP_ABSMAX_MAXB override (tuning).
    const char *ef = getenv("MVIP_ABSMAX_FPT"), *eb = getenv("MVIP_ABSMAX_MAXB");
    if (ef || eb) {
        const int64_t fpt = ef ? atoi(ef) : 256, maxb = eb ? atoi(eb) : 256;
        const int64_t b = (n + 256 * fpt - 1) / (256 * fpt);
        return (unsigned)(b < 1 ? 1 : (b > maxb ? maxb : b));
    }
    int64_t b = (n + 256 * 64 - 1) / (256 * 64);
    if (b > 64) {
        const int64_t big = (n + 256 * 256 - 1) / (256 * 256);
        b = big > 64 ? big : 64;
    }
    return (unsigned)(b < 1 ? 1 : (b > 256 ? 256 : b));
}

// scale2 = {s, 1/s} with s a power of two such that absmax*s lies in [2^9, 2^10)
__global__ void cv_scale_kernel(const unsigned *__restrict__ bits, float *__restrict__ scale2) {
    const float m = __uint_as_float(*bits);
    float s = 1.f;
    if (m > 0.f && m < 3.0e38f) {
        int e;
        frexpf(m, &e);
        int k = 10 - e;
        if (k > 60) k = 60;
        if (k < -60) k = -60;
        s = ldexpf(1.f, k);
    }
    scale2[0] = s;
    scale2[1] = 1.f / s;
}

// ---- weight packing ----------------------------------------------------------------------------------
// one thread per 16-byte fragment piece.  transpose = 1 packs the data-gradient operator:
// W'[co'][ci'][ky][kx] = W[ci'][co'][2-ky][2-kx]  (Cout' = Cin, Cin' = Cout of the forward layer).
// *lo_flag |= 1 when a lo fragment holds a non-zero half (looked at before the atomic: once set, nobody writes again)
__device__ __forceinline__ void note_lo(const uint4 &lo, unsigned *lo_flag) {
    if (((lo.x | lo.y | lo.z | lo.w) & 0x7fff7fffu) && *reinterpret_cast<volatile unsigned *>(lo_flag) == 0u) atomicOr(lo_flag, 1u);
}

__global__ void cv_pack_kernel(const float *__restrict__ w, int Cout, int Cin, int transpose,
                               const float *__restrict__ scale2, uint4 *__restrict__ out, unsigned *__restrict__ lo_flag) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int CK = Cin / 16;
    const int64_t total = (int64_t)(Cout / 32) * CK * 9 * 2 * 64;
    if (idx >= total) return;
    int64_t r = idx;
    const int lane = (int)(r % 64); r /= 64;
    const int hl = (int)(r % 2); r /= 2;
    const int kx = (int)(r % 3); r /= 3;
    const int ky = (int)(r % 3); r /= 3;
    const int ck = (int)(r % CK); r /= CK;
    const int co = (int)r * 32 + (lane & 31);
    const int kg = lane >> 5;
    const float s = scale2[0];
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = ck * 16 + kg * 8 + j;
        const int64_t src = transpose ? (((int64_t)ci * Cout + co) * 3 + (2 - ky)) * 3 + (2 - kx)
                                      : (((int64_t)co * Cin + ci) * 3 + ky) * 3 + kx;
        v[j] = w[src] * s;
    }
    uint4 hi, lo;
    split8(v, hi, lo);
    out[idx] = hl ? lo : hi;
    if (hl) note_lo(lo, lo_flag);
}

// A operand of the plain GEMM (1x1 "convolution"): A[m][k] = src[m*sm + k*sk], packed [M/32][K/16][hl][lane][8]
__global__ void gm_pack_kernel(const float *__restrict__ src, int M, int K, int64_t sm, int64_t sk,
                               const float *__restrict__ scale2, uint4 *__restrict__ out, unsigned *__restrict__ lo_flag) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int CK = K / 16;
    const int64_t total = (int64_t)(M / 32) * CK * 2 * 64;
    if (idx >= total) return;
    int64_t r = idx;
    const int lane = (int)(r % 64); r /= 64;
    const int hl = (int)(r % 2); r /= 2;
    const int ck = (int)(r % CK); r /= CK;
    const int m = (int)r * 32 + (lane & 31);
    const int kg = lane >> 5;
    const float s = scale2[0];
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(int64_t)m * sm + (int64_t)(ck * 16 + kg * 8 + j) * sk] * s;
    uint4 hi, lo;
    split8(v, hi, lo);
    out[idx] = hl ? lo : hi;
    if (hl) note_lo(lo, lo_flag);
}

// ---- activation producers ------------------------------------------------------------------------------
// grid (ceil(HW/256), N*C/16): thread = pixel, 16 channel rows read coalesced, 4 plane pieces written.
// GN: z = act((x - mean)*rstd*gamma + beta) with per-group statistics; otherwise z = x * scale2[0].
template <bool GN, bool SILU>
__global__ void __launch_bounds__(256)
cv_to_split_kernel(const float *__restrict__ x, const float *__restrict__ gamma, const float *__restrict__ beta,
                   const float *__restrict__ mean, const float *__restrict__ rstd, const float *__restrict__ scale2,
                   int C, int64_t HW, int cpg, uint4 *__restrict__ xs, int64_t sn, int64_t sc, int64_t sp, int prec,
                   const double *__restrict__ part = nullptr, int chunks = 0, float eps = 0.f,
                   float *__restrict__ mean_out = nullptr, float *__restrict__ rstd_out = nullptr) {
    __shared__ float pa[16], pb[16], pm[16];
    const int CK = C / 16;
    const int ck = (int)(blockIdx.y % CK);
    const int64_t n = blockIdx.y / CK;
    if (GN) {
        const int G = C / cpg;
        if (part) {
            // Statistics straight from the moment partials of gn_moments_kernel (csrc/group_norm.hip): each of the <= 5
            // groups this block's 16 channels touch is summed by one wave in gn_finalize_kernel's order (lane-strided,
            // then the xor tree), so mean / rstd are bit-identical to that kernel's -- whose launch this saves when
            // nobody else needs them (the no-grad forwards: the whole UNet and the masked-image encode).
            __shared__ float gmean[8], grstd[8];
            const int g0 = (ck * 16) / cpg, g1 = (ck * 16 + 15) / cpg;
            const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
            for (int gi = wave; gi <= g1 - g0; gi += 4) {
                const int64_t pbase = (n * C + (int64_t)(g0 + gi) * cpg) * chunks;
                const int P = cpg * chunks;
                double sm = 0.0, q = 0.0;
                for (int i = lane; i < P; i += 64) { sm += part[(pbase + i) * 2]; q += part[(pbase + i) * 2 + 1]; }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { sm += __shfl_xor(sm, o, 64); q += __shfl_xor(q, o, 64); }
                if (lane == 0) {
                    const double m = (double)cpg * (double)HW;
                    const double mu = sm / m;
                    double var = q / m - mu * mu;
                    if (var < 0.0) var = 0.0;
                    gmean[gi] = (float)mu;
                    grstd[gi] = (float)(1.0 / sqrt(var + (double)eps));
                    // the statistics also leave for a backward that needs them (the first pixel block of every channel chunk
                    // writes its groups; chunks that share a group write the same bits): no gn_finalize launch
                    if (mean_out && blockIdx.x == 0) { mean_out[n * G + g0 + gi] = gmean[gi]; rstd_out[n * G + g0 + gi] = grstd[gi]; }
                }
            }
            __syncthreads();
            if (threadIdx.x < 16) {
                const int c = ck * 16 + threadIdx.x;
                const int gi = c / cpg - g0;
                pm[threadIdx.x] = gmean[gi];
                pa[threadIdx.x] = grstd[gi] * (gamma ? gamma[c] : 1.f);
                pb[threadIdx.x] = beta ? beta[c] : 0.f;
            }
        } else if (threadIdx.x < 16) {
            const int c = ck * 16 + threadIdx.x;
            const int g = c / cpg;
            pm[threadIdx.x] = mean[n * G + g];
            pa[threadIdx.x] = rstd[n * G + g] * (gamma ? gamma[c] : 1.f);
            pb[threadIdx.x] = beta ? beta[c] : 0.f;
        }
        __syncthreads();
    }
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float s = (!GN && scale2) ? scale2[0] : 1.f;
    const float *xr = x + n * sn + (int64_t)(ck * 16) * sc + p * sp;      // element (n, channel, pixel)
    float v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = xr[c * sc];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        if (GN) {
            const float z = (v[c] - pm[c]) * pa[c] + pb[c];
            // SiLU with v_rcp_f32 (1 ulp) instead of the ten-instruction IEEE divide: this writer is partly VALU-bound
            // (84 launches of the SDS step 1.13 -> 1.02 ms); the result is rounded to fp16 hi + lo (2^-22) right below
            v[c] = SILU ? z * __builtin_amdgcn_rcpf(1.0f + expf(-z)) : z;
        } else {
            v[c] *= s;
        }
    }
    uint4 *dst = xs + ((n * CK + ck) * 4) * HW + p;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = v[kg * 8 + j];
        uint4 hi, lo;
        split8(t, hi, lo);
        dst[(kg * 2 + 0) * HW] = hi;
        if (prec != 1) dst[(kg * 2 + 1) * HW] = lo;          // prec 1 (fp16 mode): consumers fetch the hi plane only
    }
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Nearest-neighbour 2x up-sampling folded into the plane writer (Upsample2D of the UNet, F.interpolate(scale_factor=2) in
// front of a 3x3 convolution): output pixel (oy, ox) of the [2H, 2W] image reads x[oy / 2][ox / 2]; the up-sampled fp32
// tensor never exists.  grid (ceil(4HW/256), N*C/16).
__global__ void __launch_bounds__(256)
cv_to_split_up2_kernel(const float *__restrict__ x, const float *__restrict__ scale2, int C, int H, int W,
                       uint4 *__restrict__ xs, int prec) {
    const int CK = C / 16;
    const int ck = (int)(blockIdx.y % CK);
    const int64_t n = blockIdx.y / CK;
    const int64_t OHW = (int64_t)4 * H * W;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= OHW) return;
    const int oy = (int)(p / (2 * W)), ox = (int)(p - (int64_t)oy * 2 * W);
    const float s = scale2 ? scale2[0] : 1.f;
    const float *xr = x + (n * C + (int64_t)ck * 16) * H * W + (int64_t)(oy >> 1) * W + (ox >> 1);
    float v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = xr[(int64_t)c * H * W] * s;
    uint4 *dst = xs + ((n * CK + ck) * 4) * OHW + p;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = v[kg * 8 + j];
        uint4 hi, lo;
        split8(t, hi, lo);
        dst[(kg * 2 + 0) * OHW] = hi;
        if (prec != 1) dst[(kg * 2 + 1) * OHW] = lo;
    }
}

// ---- general convolution as a GEMM: the strided and the narrow-channel layers -----------------------------------
// (the three stride-2 down-samplers of the UNet and of the VAE encoder, conv_in / conv_out with 3, 4, 8 or 9 channels,
// quant_conv: DS_NeRF/guidance/sd_utils.py:207, :240 -- none fits the 3x3 stride-1 kernel's 16 / 32-channel operand tiles)
//   X_col[k = ci*KH*KW + ky*KW + kx][p = oy*OW + ox] = x[ci][oy*stride + ky - pad_top][ox*stride + kx - pad_left]
// written DIRECTLY as fp16 hi/lo split planes [N][KP/16][2][2][PP][8] (KP, PP: K and P padded to the GEMM's multiples, zero
// filled), so the convolution is mvip_gemm_f16x3 with the natural weight [Cout][Cin*KH*KW] as the A operand; the data
// gradient is the GEMM with A^T into col[N][KP][PP] followed by the gather below.
// grid (ceil(PP/256), N*KP/16): thread = output pixel, 16 consecutive k.
// KH_T, KW_T > 0: the kernel size as a compile-time constant (3 x 3 and 1 x 1: every layer of the SDS networks) -- the 32
// divisions by `taps` and KW per thread become multiplications; with run-time divisors this writer was bound by them.
template <int KH_T, int KW_T>
__global__ void __launch_bounds__(256)
cv_im2col_split_kernel(const float *__restrict__ x, int Cin, int H, int W, int KH_, int KW_, int stride, int pad_top,
                       int pad_left, int OH, int OW, int KP, int64_t PP, const float *__restrict__ scale2,
                       uint4 *__restrict__ xs, int prec) {
    const int KH = KH_T > 0 ? KH_T : KH_, KW = KW_T > 0 ? KW_T : KW_;
    const int CKP = KP / 16;
    const int ck = (int)(blockIdx.y % CKP);
    const int64_t n = blockIdx.y / CKP;
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= PP) return;
    const float s = scale2 ? scale2[0] : 1.f;
    const int taps = KH * KW, K = Cin * taps;
    const bool inside = p < (int64_t)OH * OW;
    const unsigned pu = (unsigned)p;                                   // OH * OW <= 2^30 (checked by the caller)
    const int oy = inside ? (int)(pu / (unsigned)OW) : 0, ox = inside ? (int)(pu - (unsigned)oy * (unsigned)OW) : 0;
    const float *xn = x + n * Cin * H * W;
    // all 16 gathers are issued before any is used: a load under its own `if` is followed by s_waitcnt vmcnt(0), i.e. 16
    // exposed memory latencies per thread (invalid taps read element 0 and are zeroed afterwards)
    float v[16];
    int64_t src[16];
    bool ok[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int k = ck * 16 + c;
        const int kk = k < K ? k : 0;
        const int ci = kk / taps, t = kk - ci * taps;
        const int ky = t / KW, kx = t - ky * KW;
        const int iy = oy * stride + ky - pad_top, ix = ox * stride + kx - pad_left;
        ok[c] = inside && k < K && iy >= 0 && iy < H && ix >= 0 && ix < W;
        src[c] = ok[c] ? ((int64_t)ci * H + iy) * W + ix : 0;
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = xn[src[c]];
#pragma unroll
    for (int c = 0; c < 16; ++c) v[c] = ok[c] ? v[c] * s : 0.f;
    uint4 *dst = xs + ((n * CKP + ck) * 4) * PP + p;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
        float t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = v[kg * 8 + j];
        uint4 hi, lo;
        split8(t8, hi, lo);
        dst[(kg * 2 + 0) * PP] = hi;
        if (prec != 1) dst[(kg * 2 + 1) * PP] = lo;
    }
}

// dx[n][ci][iy][ix] = sum over the (ky, kx) whose output pixel exists of col[n][ci*KH*KW + ky*KW + kx][oy*OW + ox]:
// a gather (deterministic), one thread per input element; grid (ceil(H W / 256), N * Cin).  KH_T, KW_T, ST_T > 0: kernel
// size and stride as compile-time constants (the tap loops unroll, `% stride` and `/ stride` become shifts, and the flat
// 64-bit index with its four divisions per thread is gone: that arithmetic, not memory, bounded the run-time version).
template <int KH_T, int KW_T, int ST_T>
__global__ void __launch_bounds__(256)
cv_col2im_kernel(const float *__restrict__ col, int Cin, int H, int W, int KH_, int KW_, int stride_, int pad_top, int pad_left,
                 int OH, int OW, int KP, int64_t PP, float *__restrict__ dx) {
    const int KH = KH_T > 0 ? KH_T : KH_, KW = KW_T > 0 ? KW_T : KW_, stride = ST_T > 0 ? ST_T : stride_;
    const unsigned p = blockIdx.x * 256u + threadIdx.x;                  // H * W <= 2^30 (checked by the caller)
    if (p >= (unsigned)(H * W)) return;
    const int iy = (int)(p / (unsigned)W), ix = (int)(p - (unsigned)iy * (unsigned)W);
    const int ci = (int)(blockIdx.y % (unsigned)Cin);
    const int64_t n = blockIdx.y / (unsigned)Cin;
    const float *cn = col + n * KP * PP;
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < KH; ++ky) {
        const int ty = iy + pad_top - ky;
        if (ty < 0 || ty % stride != 0) continue;
        const int oy = ty / stride;
        if (oy >= OH) continue;
#pragma unroll
        for (int kx = 0; kx < KW; ++kx) {
            const int tx = ix + pad_left - kx;
            if (tx < 0 || tx % stride != 0) continue;
            const int ox = tx / stride;
            if (ox >= OW) continue;
            acc += cn[(int64_t)(ci * KH * KW + ky * KW + kx) * PP + (int64_t)oy * OW + ox];
        }
    }
    dx[((n * Cin + ci) * H + iy) * (int64_t)W + ix] = acc;
}

// ---- the convolution -------------------------------------------------------------------------------------
struct ConvArgs {
    const char *xs, *wp, *zero16;
    const float *bias, *chan_add, *residual, *x_scale2, *w_scale2;
    float *y;
    int N, CK, Cout, H, W, tilesX, tilesY, MB;
    // split-K (partial != nullptr): workgroup `split` of `splits` contracts channel chunks [split*cks, (split+1)*cks) and
    // writes its raw accumulators to partial[split][n][Cout][H][W]; cv_split_reduce_kernel sums them in order
    int splits, cks;
    float *partial;
    // GroupNorm moment partials of the OUTPUT from the unsplit epilogue (round 6; null = none): per (image, output channel,
    // pixel tile, wave) the fp32 sum and sum of squares of the wave's 64 finished values, tile_part[(((n * Cout + co) * tiles
    // + tile) * 4 + wave) * 2 + {0, 1}]; gn_moments_from_tiles_kernel adds them in fp64 in index order.  The GroupNorm
    // that reads y next then needs no pass over y for its statistics (it was gn_moments_kernel: one full read of y).
    float *tile_part;
    int prio;                    // 1: the co-resident workgroups alternate their wave priority stage by stage
#ifdef MVIP_EXPERIMENT_CONV
    int dbg;                     // timing experiments (MVIP_CONV_DBG): 1 = no epilogue, 2 = no MFMAs, 4 = no input DMA, 8 = no weight DMA, 16 = no barrier, 32 = linear B reads
    unsigned long long *probe;   // per workgroup (8 words): {shader cycles total, 100 MHz ticks total, prologue, DMA issue, compute, epilogue, barrier wait, start tick}
#endif
};

// MT <= 2: 80 KB of LDS and 256 registers, i.e. TWO workgroups per CU = two waves per SIMD, so one wave's LDS
// reads, waits and epilogue run under the other wave's MFMAs.
// TW = 32: pixel tile 8 rows x 32 columns, an MFMA column block (32 pixels) = one image row of the tile.
// TW = 16 (images narrower than 32 pixels: the UNet's 16x16 level): tile 16 x 16, a column block = two rows of 16.
// NW = 8 (TW = 32 only, opt-in): a workgroup of eight waves owns a 16 x 32 pixel tile, so the weights of a stage and
// the haloed input tile are shared by twice the MFMAs (10.9 instead of 16.8 B/clk/CU of LDS-DMA at full matrix rate for
// MT = 2).  Built to test whether operand delivery bounds this kernel as it bounds the plain GEMM and the attention
// kernel: it does not -- both shapes run the VAE / UNet convolutions in the same time (round 2, tools/conv_wide_ab.py).
// TW = 8 (8 x 8 images, the UNet's innermost level): a workgroup takes FOUR images, wave w image w (two column blocks of
// 4 rows x 8 pixels); the LDS tile holds the four haloed 10 x 10 images one after the other.
// Split-K: at the inner UNet levels the grid of (pixel tile, row block) pairs is far smaller than the chip (1280 channels
// at 16 x 16 x 2 images: 80 workgroups, at 8 x 8: 40) while one layer's weights are 59-118 MB, so those layers are bound by
// how many CUs pull weights at once; with a.partial set the channel range is divided over a.splits workgroups.
// F16 (the reference's --fp16 mode, DS_NeRF/guidance/sd_utils.py:66): ONE fp16 product per contraction step -- only the hi
// plane of the activations and the hi fragments of the weights are fetched (the lo halves of both operand images are
// neither read nor, by the producers, written), fp32 accumulation; a third of the matrix work and half the operand traffic.
// NP = 3 / 2 / 1 products per step (see the file header): 2 = the weights' lo fragments are zero and not fetched.
template <int MT, int TW = CV_TW, int NW = 4, int NP = 3>
__global__ void __launch_bounds__(NW * 64, (NW == 8 ? (MT <= 2 ? 2 : 1) : (MT <= 2 ? 2 : 1))) conv3x3_f16x3_kernel(const ConvArgs a) {
    constexpr bool F16 = NP == 1;
    constexpr int NT = NW * 64;
    constexpr int TH = NT / TW, HW = TW + 2, RPB = 32 / TW;                         // RPB: image rows per column block
    constexpr int PIX = TW == 8 ? 4 * 10 * HW : (TH + 2) * HW;
    constexpr int CV_IN_BYTES = 4 * PIX * 16;          // shadows the 8 x 32 constants: sized for this tile
    constexpr int CV_IN_ROUNDS = (4 * PIX + NT - 1) / NT;
    constexpr int WB = 3 * MT * 2 * 1024;              // weights of one stage (kernel row, 16 channels)
    __shared__ __attribute__((aligned(16))) char lds[3 * WB + 2 * CV_IN_BYTES];
    char *lds_w = lds, *lds_in = lds + 3 * WB;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, kg = lane >> 5;
#ifdef MVIP_EXPERIMENT_CONV
    unsigned long long pc0 = 0, pr0 = 0, p_sync = 0, p_comp = 0, p_pro = 0, p_mark = 0, p_bar = 0;
    if (a.probe) { pc0 = __builtin_amdgcn_s_memtime(); pr0 = __builtin_amdgcn_s_memrealtime(); }
#define CV_MARK(acc_) do { if (a.probe) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc_ += now_ - p_mark; p_mark = now_; } } while (0)
#else
#define CV_MARK(acc_) do { } while (0)
#endif
    // XCD-aware order: each of the 8 XCDs walks a contiguous range of (tile, channel-block) pairs, the
    // channel blocks of one pixel tile adjacent in time, so the tile's planes are served by that XCD's L2.
    int id = blockIdx.x;
    const int total = gridDim.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int split = id % a.splits;                    // the splits of one (tile, row block) adjacent: they share the input tile
    id /= a.splits;
    const int mb = id % a.MB;
    int tile = id / a.MB;
    const int tx = tile % a.tilesX; tile /= a.tilesX;
    const int ty = tile % a.tilesY;
    const int n = (TW == 8 ? 4 : 1) * (tile / a.tilesY);                 // first image of the workgroup
    const int y0 = ty * TH, x0 = tx * TW;
    const int H = a.H, W = a.W;
    const int64_t plane = (int64_t)H * W * 16;
    const int ck0 = split * a.cks;
    const int nck = (a.CK - ck0 < a.cks) ? a.CK - ck0 : a.cks;           // channel chunks of this workgroup (>= 1)

    int in_off[CV_IN_ROUNDS];
#pragma unroll
    for (int r = 0; r < CV_IN_ROUNDS; ++r) {
        const int s = r * NT + tid;
        in_off[r] = s < 4 * PIX ? -1 : -2;             // -1: halo outside the image (zero page), -2: no slot
        if (F16 && s < 4 * PIX && ((s / PIX) & 1)) in_off[r] = -2;        // lo planes (piece = kg*2 + hl) are not fetched
        if (s < 4 * PIX && in_off[r] != -2) {
            const int piece = s / PIX, p = s - piece * PIX;
            const int row = p / HW, col = p - row * HW;
            if constexpr (TW == 8) {
                const int img = row / 10, gy = row - img * 10 - 1, gx = col - 1;
                if (n + img < a.N && gy >= 0 && gy < 8 && gx >= 0 && gx < 8)
                    in_off[r] = (int)(img * a.CK * 4 * plane) + (int)(piece * plane) + (gy * 8 + gx) * 16;
            } else {
                const int gy = y0 - 1 + row, gx = x0 - 1 + col;
                if (gy >= 0 && gy < H && gx >= 0 && gx < W) in_off[r] = (int)(piece * plane) + (gy * W + gx) * 16;
            }
        }
    }
    const char *xs_n = a.xs + ((int64_t)n * a.CK + ck0) * 4 * plane;
    constexpr int WROW = 3 * 2 * 1024;                 // one stage of one 32-channel row block
    const char *wp_b = a.wp + ((int64_t)mb * MT * a.CK + ck0) * 3 * WROW;

    auto issue_input = [&](int ck, int buf) {
#ifdef MVIP_EXPERIMENT_CONV
        if (a.dbg & 4) return;
#endif
        const char *base = xs_n + (int64_t)ck * 4 * plane;
        char *dst = lds_in + buf * CV_IN_BYTES + wave * 1024;
#pragma unroll
        for (int r = 0; r < CV_IN_ROUNDS; ++r)
            if (in_off[r] != -2) glds16b(in_off[r] >= 0 ? base + in_off[r] : a.zero16, dst + r * (NT * 16));
    };
    auto issue_weights = [&](int t, int buf) {
#ifdef MVIP_EXPERIMENT_CONV
        if (a.dbg & 8) return;
#endif
        const char *src = wp_b + (int64_t)t * WROW + lane * 16;
        char *dst = lds_w + buf * WB;
#pragma unroll
        for (int b0 = 0; b0 < 6 * MT; b0 += NW) {       // LDS block b = m*6 + kx*2 + hl
            const int b = b0 + wave;
            if (b < 6 * MT && !(NP < 3 && (b & 1)))        // b odd = lo fragments (fetched by the three-product kernel only)
                glds16b(src + (int64_t)(b / 6) * a.CK * 3 * WROW + (b % 6) * 1024, dst + b * 1024);
        }
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

    // Software pipeline.  A "group" = one tap (kx) of one 32-channel output row block m: two A fragments (hi, lo),
    // the tap's four B fragments, six MFMAs on two independent accumulators.  Fragments are read from LDS TWO
    // groups ahead of their use, right after the first pair of MFMAs of a group and pinned there by scheduling
    // barriers: this kernel runs one wave per SIMD, so nothing hides an exposed LDS latency (the first version
    // read, waited and multiplied group by group and kept the matrix pipe 34 % busy).  Weights are staged TWO
    // stages ahead in three LDS buffers so that the next stage's fragments can be read before the barrier.
    constexpr int NG = 3 * MT;                          // groups per stage, kx-major
    const int nstage = nck * 3;
    const int jrow0 = 2 * wave;
    const int img_rows = TW == 8 ? 2 * wave : 0;        // TW = 8: every image in front of this wave's adds two halo rows
    // PD = groups between a fragment's LDS read and its use (NS sets of A fragments, set = group % NS, NG % NS == 0 so the
    // rotation survives stages).  Two groups; four (PD = 4, NS = 6 at MT = 2: ~770 cycles of cover, 203 registers)
    // measured the same on every VAE / UNet shape -- the LDS read latency is not what the waves wait for.
    constexpr int PD = 2;
    constexpr int NS = 3;
    h16x8 Ah[NS], Al[NS];
    h16x8 Bh[3][2], Bl[3][2];                           // set = kx: the tap PD groups ahead is always the one just finished
    auto load_a = [&](const char *wb, int g, int set) {
        const int kx = g / MT, m = g % MT;
        Ah[set] = *reinterpret_cast<const h16x8 *>(wb + ((m * 3 + kx) * 2 + 0) * 1024);
        if constexpr (NP == 3) Al[set] = *reinterpret_cast<const h16x8 *>(wb + ((m * 3 + kx) * 2 + 1) * 1024);
    };
    auto load_b = [&](const char *inb, int ky, int kx) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int p = ((jrow0 + j) * RPB + img_rows + l32 / TW + ky) * HW + l32 % TW + kx;
#ifdef MVIP_EXPERIMENT_CONV
            if (a.dbg & 32) p = (j * 3 + kx) * 64 + lane - (kg * 2) * PIX;          // timing only: wave-linear 1-KB fragment reads
#endif
            Bh[kx][j] = *reinterpret_cast<const h16x8 *>(inb + p * 16);
            if constexpr (!F16) Bl[kx][j] = *reinterpret_cast<const h16x8 *>(inb + PIX * 16 + p * 16);
        }
    };
    auto w_base = [&](int t) { return lds_w + (t % 3) * WB + lane * 16; };
    auto in_base = [&](int t) { return lds_in + ((t / 3) & 1) * CV_IN_BYTES + (kg * 2) * (PIX * 16); };

    issue_input(0, 0);
    issue_weights(0, 0);
    if (nstage > 1) issue_weights(1, 1);
    __syncthreads();                                    // stages 0 and 1 (and input chunk 0) have landed
    if (nstage > 2) issue_weights(2, 2);
    if (nck > 1) issue_input(1, 1);
#pragma unroll
    for (int g = 0; g < PD; ++g) {                      // PD < NG: the first PD groups are all of stage 0
        load_a(w_base(0), g, g % NS);
        if (g % MT == 0) load_b(in_base(0), 0, g / MT);
    }
#ifdef MVIP_EXPERIMENT_CONV
    if (a.probe) { p_mark = __builtin_amdgcn_s_memtime(); p_pro = p_mark - pc0; }
#endif
    // The two workgroups that share a CU sit in hardware wave slots 0 and 1 of each SIMD, and at equal priority the
    // arbiter serves the older slot first: the slot-0 workgroup of a one-round launch finished in 203 us, its partner in
    // 238 us (tools/conv_stragglers.py), the last 35 us with one wave per SIMD.  Alternating the priority stage by stage
    // (s_setprio; MVIP_CONV_PRIO=1) makes the pair finish together (226 / 232 us) and the launch 3 % shorter, which the
    // whole SDS step does not show (9.95 vs 10.0 ms of convolutions) -- off by default.
    const int prio_phase = a.prio ? (int)(__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 4) & 1u) : -1;
    for (int t = 0; t < nstage; ++t) {
        const int ck = t / 3, ky = t - ck * 3;
        if (prio_phase >= 0) { if ((t + prio_phase) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
        CV_MARK(p_comp);
        if (t > 0) {
#ifdef MVIP_EXPERIMENT_CONV
            if (!(a.dbg & 16))
#endif
            __syncthreads();                   // stage t+1 landed (vmcnt(0)); every wave is done with stage t-1
            CV_MARK(p_bar);
            if (t + 2 < nstage) issue_weights(t + 2, (t + 2) % 3);
            if (ky == 0 && ck + 1 < nck) issue_input(ck + 1, (ck + 1) & 1);
        }
        CV_MARK(p_sync);
        const char *wb = w_base(t), *inb = in_base(t);
        const char *wb_n = w_base(t + 1), *inb_n = in_base(t + 1);
        const int ky_n = (ky == 2) ? 0 : ky + 1;
        const bool more = t + 1 < nstage;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int kx = g / MT, m = g % MT, set = g % NS;
#ifdef MVIP_EXPERIMENT_CONV
            if (a.dbg & 2) {
                acc[m][0][0] += (float)Ah[set][0] + (float)Bh[kx][0][0] + (float)Bl[kx][1][0];
                acc[m][1][0] += (float)Al[set][0] + (float)Bh[kx][1][0] + (float)Bl[kx][0][0];
                const int g2 = g + PD;
                if (g2 < NG) { load_a(wb, g2, g2 % NS); if (g2 % MT == 0) load_b(inb, ky, g2 / MT); }
                else if (more) { load_a(wb_n, g2 - NG, g2 % NS); if ((g2 - NG) % MT == 0) load_b(inb_n, ky_n, (g2 - NG) / MT); }
                continue;
            }
#endif
            acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[set], Bh[kx][0], acc[m][0], 0, 0, 0);
            acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[set], Bh[kx][1], acc[m][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            {   // fragments of group g+PD (this stage or the next one)
                const int g2 = g + PD;
                if (g2 < NG) {
                    load_a(wb, g2, g2 % NS);
                    if (g2 % MT == 0) load_b(inb, ky, g2 / MT);
                } else if (more) {
                    load_a(wb_n, g2 - NG, g2 % NS);
                    if ((g2 - NG) % MT == 0) load_b(inb_n, ky_n, (g2 - NG) / MT);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (NP >= 2) {
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[set], Bl[kx][0], acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[set], Bl[kx][1], acc[m][1], 0, 0, 0);
            }
            if constexpr (NP == 3) {               // NP == 2: Al == 0 -- this pair would add exact zeros
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[set], Bh[kx][0], acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[set], Bh[kx][1], acc[m][1], 0, 0, 0);
            }
        }
    }

    CV_MARK(p_comp);
    const float inv = a.w_scale2[1] * (a.x_scale2 ? a.x_scale2[1] : 1.f);
    const int gx = x0 + l32 % TW;
    const int n_out = TW == 8 ? n + wave : n;
    if (TW == 8 && n_out >= a.N) return;
#ifdef MVIP_EXPERIMENT_CONV
    if (a.dbg & 1) {
        float t = 0.f;
        for (int m = 0; m < MT; ++m) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) t += acc[m][j][r];
        if (t == 123.456f) a.y[0] = t;
        if (a.probe && tid == 0) {
            const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
            unsigned long long *o = a.probe + 8 * (unsigned long long)blockIdx.x;
            o[0] = c1 - pc0; o[1] = r1 - pr0; o[2] = p_pro; o[3] = p_sync; o[4] = p_comp; o[5] = c1 - p_mark; o[6] = p_bar; o[7] = pr0;
        }
        return;
    }
#endif
    // The epilogue's loads are issued TOGETHER, ahead of the arithmetic: written element by element (`if (bias) v +=
    // bias[co]; if (chan_add) ...; if (residual) v += residual[o]; y[o] = v`) every load sits behind its own branch and
    // is followed by s_waitcnt vmcnt(0) -- up to 3 x 64 exposed memory latencies per wave, 122 k of the 211 k cycles of a
    // 128-channel 512 x 512 workgroup (tools/conv_probe.py).
    auto out_index = [&](int m, int j, int r, int &co) -> int64_t {
        const int gy = TW == 8 ? j * RPB + l32 / TW : y0 + (jrow0 + j) * RPB + l32 / TW;
        co = (mb * MT + m) * 32 + 8 * (r >> 2) + 4 * kg + (r & 3);
        return (((int64_t)n_out * a.Cout + co) * H + gy) * W + gx;
    };
    if (a.partial) {
        float *pp = a.partial + (int64_t)split * a.N * a.Cout * H * W;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { int co; pp[out_index(m, j, r, co)] = acc[m][j][r]; }
    } else {
        const bool hb = a.bias != nullptr, hc = a.chan_add != nullptr, hr = a.residual != nullptr;
        float bv[MT][16], cv[MT][16];
        f32x16 rv[MT][2];
        if (hr) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) { int co; rv[m][j][r] = a.residual[out_index(m, j, r, co)]; }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = (mb * MT + m) * 32 + 8 * (r >> 2) + 4 * kg + (r & 3);
                bv[m][r] = hb ? a.bias[co] : 0.f;
                cv[m][r] = hc ? a.chan_add[(int64_t)n_out * a.Cout + co] : 0.f;
            }
        bool with_moments = false;
        if constexpr (TW != 8 && NW == 4) with_moments = a.tile_part != nullptr;
        if (!with_moments) {
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int co;
                        const int64_t o = out_index(m, j, r, co);
                        float v = acc[m][j][r] * inv;
                        if (hb) v += bv[m][r];
                        if (hc) v += cv[m][r];
                        if (hr) v += rv[m][j][r];
                        a.y[o] = v;
                    }
        }
        if constexpr (TW != 8 && NW == 4) {
            if (with_moments) {
                // The same stores, channel by channel (both pixels of a register pair finished together), and the channel's
                // moment partial right behind them.  Register r of lane (l32, kg) holds output channel 8 (r >> 2) + 4 kg + (r & 3)
                // of row tile m at the two pixels (j = 0, 1) of column l32: the channel's sum over the wave's 64 pixels = the two
                // values added, then the total over the 32 lanes of the half-wave -- four row_shr steps (lane 15 of each 16-lane
                // row = the row's total) and one row_bcast:15 into rows 1 and 3 (lanes 31 / 63 = the half-waves' totals).  Five
                // DPP adds per statistic and channel on the VALU, under the co-resident workgroup's MFMAs; no LDS, no barrier.
                const int tiles = a.tilesX * a.tilesY, tile_lin = ty * a.tilesX + tx;
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        int co;
                        const int64_t o0 = out_index(m, 0, r, co), o1 = out_index(m, 1, r, co);
                        float v0 = acc[m][0][r] * inv, v1 = acc[m][1][r] * inv;
                        if (hb) { v0 += bv[m][r]; v1 += bv[m][r]; }
                        if (hc) { v0 += cv[m][r]; v1 += cv[m][r]; }
                        if (hr) { v0 += rv[m][0][r]; v1 += rv[m][1][r]; }
                        a.y[o0] = v0;
                        a.y[o1] = v1;
                        float sm = v0 + v1, q = v0 * v0 + v1 * v1;
                        sm += dpp_f32<0x111>(0.f, sm); q += dpp_f32<0x111>(0.f, q);
                        sm += dpp_f32<0x112>(0.f, sm); q += dpp_f32<0x112>(0.f, q);
                        sm += dpp_f32<0x114>(0.f, sm); q += dpp_f32<0x114>(0.f, q);
                        sm += dpp_f32<0x118>(0.f, sm); q += dpp_f32<0x118>(0.f, q);
                        sm += dpp_f32<0x142, 0xa>(0.f, sm); q += dpp_f32<0x142, 0xa>(0.f, q);
                        if (l32 == 31) {
                            float2 *dst = reinterpret_cast<float2 *>(a.tile_part) + (((int64_t)n_out * a.Cout + co) * tiles + tile_lin) * 4 + wave;
                            *dst = make_float2(sm, q);
                        }
                    }
            }
        }
    }
#ifdef MVIP_EXPERIMENT_CONV
    if (a.probe && tid == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        unsigned long long *o = a.probe + 8 * (unsigned long long)blockIdx.x;
        o[0] = c1 - pc0; o[1] = r1 - pr0; o[2] = p_pro; o[3] = p_sync; o[4] = p_comp; o[5] = c1 - p_mark; o[6] = p_bar; o[7] = pr0;
        if (a.dbg & 64) o[5] = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) << 32);   // HW_ID, XCC_ID
    }
#endif
}

// y = (sum_s partial[s]) / (s_w s_x) + bias + chan_add + residual, the splits added in index order (deterministic)
__global__ void cv_split_reduce_kernel(const float *__restrict__ partial, int splits, int64_t total, int Cout, int64_t HW,
                                       const float *__restrict__ w_scale2, const float *__restrict__ x_scale2,
                                       const float *__restrict__ bias, const float *__restrict__ chan_add,
                                       const float *__restrict__ residual, float *__restrict__ y) {
    // grid (ceil(HW / 1024), N * Cout): the row (n, co) is the block's y index -- a flat 64-bit index cost two 64-bit
    // divisions per thread, as much time as the memory traffic of these small tensors
    const int64_t p4 = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;        // HW % 4 == 0: four pixels of one channel
    if (p4 >= HW) return;
    const float inv = w_scale2[1] * (x_scale2 ? x_scale2[1] : 1.f);
    const unsigned rows = (unsigned)(total / HW);      // wave-uniform; N * Cout < 2^31
  for (unsigned nc = blockIdx.y; nc < rows; nc += gridDim.y) {                      // n * Cout + co (one trip unless rows > 65535)
    const int64_t i4 = (int64_t)nc * HW + p4;
    f32x4 sum = *reinterpret_cast<const f32x4 *>(partial + i4);
    int s = 1;
    for (; s + 3 < splits; s += 4) {                   // four loads in flight, added in index order
        const f32x4 p0 = *reinterpret_cast<const f32x4 *>(partial + (int64_t)s * total + i4);
        const f32x4 p1 = *reinterpret_cast<const f32x4 *>(partial + (int64_t)(s + 1) * total + i4);
        const f32x4 p2 = *reinterpret_cast<const f32x4 *>(partial + (int64_t)(s + 2) * total + i4);
        const f32x4 p3 = *reinterpret_cast<const f32x4 *>(partial + (int64_t)(s + 3) * total + i4);
        sum += p0; sum += p1; sum += p2; sum += p3;
    }
    for (; s < splits; ++s) sum += *reinterpret_cast<const f32x4 *>(partial + (int64_t)s * total + i4);
    f32x4 v = sum * inv;                               // same order of roundings as the unsplit epilogue
    if (bias) v += bias[nc % (unsigned)Cout];
    if (chan_add) v += chan_add[nc];
    if (residual) v += *reinterpret_cast<const f32x4 *>(residual + i4);
    *reinterpret_cast<f32x4 *>(y + i4) = v;
  }
}

// The same reduction that also leaves the GroupNorm moments of its OUTPUT rows, in the layout gn_moments_kernel
// (csrc/group_norm.hip) writes for HW <= 4096 (one {sum, sum of squares} fp64 pair per (sample, channel) row): the next
// layer's GroupNorm then needs no pass over y for its statistics.  A row (HW = 64, 256, 1024 or 4096 values) is owned by
// LPR = min(HW / 4, 256) consecutive threads (rows of 1024+ values: one workgroup, looping), so the sums are formed in a
// fixed order: per thread in index order, a DPP scan over the row's lanes, then the four waves in order.
template <int LPR, int IT = 1>
__global__ void __launch_bounds__(256)
cv_split_reduce_moments_kernel(const float *__restrict__ partial, int splits, int64_t total, int Cout, int64_t HW,
                               const float *__restrict__ w_scale2, const float *__restrict__ x_scale2,
                               const float *__restrict__ bias, const float *__restrict__ chan_add,
                               const float *__restrict__ residual, float *__restrict__ y, double *__restrict__ moments) {
    constexpr int ROWS = 256 / LPR;                           // rows per workgroup
    const int64_t row = (int64_t)blockIdx.x * ROWS + threadIdx.x / LPR;             // n * Cout + co
    const int lr = threadIdx.x % LPR;
    const float inv = w_scale2[1] * (x_scale2 ? x_scale2[1] : 1.f);
    const float bv = bias ? bias[(unsigned)row % (unsigned)Cout] : 0.f, cv = chan_add ? chan_add[row] : 0.f;      // N * Cout < 2^31
    double sm = 0.0, q = 0.0;
    const int64_t i0 = row * HW + (int64_t)lr * 4;        // IT pieces of LPR * 4 values per row, all loads of a split in flight
    f32x4 sum[IT], rv[IT];
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        sum[it] = *reinterpret_cast<const f32x4 *>(partial + i0 + (int64_t)it * LPR * 4);
        if (residual) rv[it] = *reinterpret_cast<const f32x4 *>(residual + i0 + (int64_t)it * LPR * 4);
    }
    for (int sp = 1; sp < splits; ++sp) {
        f32x4 t[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) t[it] = *reinterpret_cast<const f32x4 *>(partial + (int64_t)sp * total + i0 + (int64_t)it * LPR * 4);
#pragma unroll
        for (int it = 0; it < IT; ++it) sum[it] += t[it];
    }
#pragma unroll
    for (int it = 0; it < IT; ++it) {
        f32x4 v = sum[it] * inv;                           // same order of roundings as cv_split_reduce_kernel
        if (bias) v += bv;
        if (chan_add) v += cv;
        if (residual) v += rv[it];
        *reinterpret_cast<f32x4 *>(y + i0 + (int64_t)it * LPR * 4) = v;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const double d = (double)v[k]; sm += d; q += d * d; }
    }
    // totals over the row's lanes on DPP row operations (common.h; a dozen VALU steps instead of 24 dependent ds_bpermute round
    // trips): the four row_shr steps leave a 16-lane row's total in its last lane, the two broadcasts the wave's in lane 63
    sm += dpp_f64<0x111>(0.0, sm); q += dpp_f64<0x111>(0.0, q);
    sm += dpp_f64<0x112>(0.0, sm); q += dpp_f64<0x112>(0.0, q);
    sm += dpp_f64<0x114>(0.0, sm); q += dpp_f64<0x114>(0.0, q);
    sm += dpp_f64<0x118>(0.0, sm); q += dpp_f64<0x118>(0.0, q);
    if constexpr (LPR >= 64) {
        sm += dpp_f64<0x142, 0xa>(0.0, sm); q += dpp_f64<0x142, 0xa>(0.0, q);
        sm += dpp_f64<0x143, 0xc>(0.0, sm); q += dpp_f64<0x143, 0xc>(0.0, q);
    }
    if constexpr (LPR == 256) {
        __shared__ double red[2][4];
        if ((threadIdx.x & 63) == 63) { red[0][threadIdx.x >> 6] = sm; red[1][threadIdx.x >> 6] = q; }
        __syncthreads();
        sm = red[0][0] + red[0][1] + red[0][2] + red[0][3];
        q = red[1][0] + red[1][1] + red[1][2] + red[1][3];
    }
    if (lr == LPR - 1) { moments[row * 2] = sm; moments[row * 2 + 1] = q; }
}

// GroupNorm moments of a convolution output from the tile partials its unsplit epilogue left (ConvArgs::tile_part): the
// (image, channel) row's `parts` (sum, sum of squares) pairs added in fp64 and written in the layout gn_moments_kernel leaves
// for this row length (`chunks` pairs per row: the total in the first, zeros in the rest), which is what
// cv_to_split_kernel<GN> and gn_finalize_kernel read.
__global__ void __launch_bounds__(256)
gn_moments_from_tiles_kernel(const float2 *__restrict__ part, int parts, int64_t rows, int chunks, double *__restrict__ out) {
    // one WORKGROUP per row (the first version -- one wave per row, one load in flight per lane -- took 17.6 us per launch at
    // 4,096 partials per row: as long as the pass over y it replaces): thread t adds partials t, t + 256, ... four at a time,
    // then the xor tree inside each wave and the four waves in index order -- a fixed order, bit-reproducible
    __shared__ double red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row = blockIdx.x;
    const float2 *p = part + row * parts;
    double sm = 0.0, q = 0.0;
    int i = threadIdx.x;
    for (; i + 768 < parts; i += 1024) {
        const float2 t0 = p[i], t1 = p[i + 256], t2 = p[i + 512], t3 = p[i + 768];
        sm += ((double)t0.x + (double)t1.x) + ((double)t2.x + (double)t3.x);
        q += ((double)t0.y + (double)t1.y) + ((double)t2.y + (double)t3.y);
    }
    for (; i < parts; i += 256) { const float2 t = p[i]; sm += (double)t.x; q += (double)t.y; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { sm += __shfl_xor(sm, o, 64); q += __shfl_xor(q, o, 64); }
    if (lane == 0) { red[0][wave] = sm; red[1][wave] = q; }
    __syncthreads();
    if (threadIdx.x < chunks) {
        const double ts = ((red[0][0] + red[0][1]) + red[0][2]) + red[0][3], tq = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        out[(row * chunks + threadIdx.x) * 2] = threadIdx.x == 0 ? ts : 0.0;
        out[(row * chunks + threadIdx.x) * 2 + 1] = threadIdx.x == 0 ? tq : 0.0;
    }
}

// The same reduction whose result leaves as split planes [N][M/16][2][2][P][8 halves] * out_scale (an operand sink behind a
// split-K launch): thread = (sample, 8-row block, column); the 8 rows of a fragment are 8 coalesced row reads per split.
__global__ void __launch_bounds__(256)
cv_split_reduce_planes_kernel(const float *__restrict__ partial, int splits, int N, int M, int64_t P,
                              const float *__restrict__ w_scale2, const float *__restrict__ x_scale2,
                              const float *__restrict__ bias, const float *__restrict__ residual, float out_scale,
                              uint4 *__restrict__ planes, int prec) {
    const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int blk = blockIdx.y % (M / 8), n = blockIdx.y / (M / 8);
    if (p >= P) return;
    const float inv = w_scale2[1] * (x_scale2 ? x_scale2[1] : 1.f);
    const int64_t total = (int64_t)N * M * P;
    const int64_t base = ((int64_t)n * M + blk * 8) * P + p;
    float sum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) sum[j] = partial[base + j * P];
    for (int s = 1; s < splits; ++s) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = partial[(int64_t)s * total + base + j * P];
#pragma unroll
        for (int j = 0; j < 8; ++j) sum[j] += t[j];
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float t = sum[j] * inv;
        if (bias) t += bias[blk * 8 + j];
        if (residual) t += residual[base + j * P];
        v[j] = t * out_scale;
    }
    uint4 hi, lo;
    sink_split8(v, hi, lo);
    uint4 *dst = planes + (((int64_t)n * (M / 16) + (blk >> 1)) * 4 + (blk & 1) * 2) * P + p;
    dst[0] = hi;
    if (prec != 1) dst[P] = lo;
}

// ---- plain GEMM (1x1 convolution, attention products) on the same operand formats -----------------------
//   Y[n][m][p] = sum_k A[m][k] X[n][k][p] / (s_a s_x) + bias[m] + chan_add[n][m] + residual[n][m][p]
// A packed by gm_pack_kernel, X in split planes [n][K/16][2][2][P][8].  Workgroup = 32*MT rows x 256
// columns, stage = 32 k (two 16-channel chunks: 48 MFMAs per wave at MT=4), double-buffered DMA.
constexpr int GM_PIX = 256;
constexpr int GM_KC = 2;
constexpr int64_t GM_P_MAX = (int64_t)1 << 26;      // columns per sample: the streamed kernel's per-lane 32-bit plane offsets

struct GemmArgs {
    const char *xs, *wp;
    const float *bias, *chan_add, *residual, *x_scale2, *w_scale2;
    float *y;
    int N, CK, M, MB, tiles;
    int64_t P;
    // GEGLU epilogue (MT = 2 only): the rows are interleaved in 32-row tiles (value tile, gate tile); the kernel
    // writes value * gelu(gate) as [N][M/2][P], zero for columns >= geglu_L, and collects the absolute maximum
    int geglu_L;                 // 0: plain epilogue
    unsigned *absmax_bits;
    // split-K (gemm_f16x3_kernel only; partial != nullptr): workgroup `split` contracts stages [split*sks, (split+1)*sks)
    // and writes raw accumulators to partial[split][n][M][P]; cv_split_reduce_kernel adds them in order
    int splits, sks;
    float *partial;
    // Operand sinks (nsec > 0; gemm5_f16x3_kernel only): the M rows are cut into <= 3 consecutive sections (multiples of
    // 64 rows), each written in the operand format of the contraction that consumes it, scaled by a power of two fixed
    // before the launch: kind 1 = split planes [N][rows/16][2][2][P][8] (also the GEGLU product when geglu_L > 0, rows =
    // M/2), kind 2 = attention V fragments [N][heads][DT][P/16][2][64][8] -- that section is computed TRANSPOSED (the MFMA
    // operands swapped: lane = output row, registers = tokens), which is the fragment order of csrc/attention.hip.
    // y is not written for these rows.
    struct Sec { char *ptr; float scale; int row_end, kind; } sec[3];
    int nsec, v_dt;              // v_dt: 32-row tiles per head in a kind-2 section
    int prec;                    // 1: single fp16 product (hi planes / hi fragments only; sinks write no lo halves)
    // LayerNorm statistics of the OUTPUT (plain fp32 epilogue, unsplit launches only): the rows are the channels a later
    // LayerNorm reduces over, so each workgroup leaves, per column (token), the fp64 sum and sum of squares of its 32 MT
    // finished rows in the layout ln_stats_kernel (csrc/transformer.hip) writes for 64-channel segments:
    // ln_part[((n * MB + mb) * 2 + {0, 1}) * P + p].  The statistics launch of the next LayerNorm disappears.
    double *ln_part;
#ifdef MVIP_EXPERIMENT_GEMM
    int dbg;                     // timing experiments: 1 = no epilogue stores, 2 = no MFMAs, 4 = no LDS reads either
#endif
};

// epilogue of the SWAP instantiation: the accumulators are transposed (lane = row of the weight tile, registers = the 32
// tokens of column block j) and leave as attention V fragments [N][heads][v_dt][P/16][2][64][8 halves]; the launch's M
// rows are the heads * v_dt * 32 rows of ONE section (a.sec[0])
template <int MT>
__device__ __forceinline__ void gemm_epilogue_vfrag(const GemmArgs &a, f32x16 (&acc)[MT][2], int n, int mb, int64_t p0, int wave,
                                                    int lane) {
    const int l32 = lane & 31;
    const float sc = a.sec[0].scale;
    const float mul = a.w_scale2[1] * (a.x_scale2 ? a.x_scale2[1] : 1.f) * sc;
    const bool hb = a.bias != nullptr;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int lt = mb * MT + m;                            // (head, dt) block of this row tile: lt = head * v_dt + dt
        const float bs = hb ? a.bias[lt * 32 + l32] * sc : 0.f;
        char *vt = a.sec[0].ptr + ((int64_t)n * (a.M / 32) + lt) * (a.P / 16) * 2048;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[m][j][r] * mul + bs;
            sink_store_vfrag(vt, (int)((p0 + (2 * wave + j) * 32) / 16), lane, v, a.prec != 1);
        }
    }
}

// epilogue shared by the 32/64-row GEMM kernels: acc[m][j] = 32 x 32 tile (row tile mb*MT + m, columns p0 + (2 wave + j)*32 ..)
// LN: the instantiation can leave LayerNorm statistics (GemmArgs::ln_part; the B-in-registers kernel only -- in the
// LDS-staged kernel the extra live registers spilled)
template <int MT, bool LN = false>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs &a, f32x16 (&acc)[MT][2], int n, int mb, int64_t p0, int split,
                                              int wave, int lane) {
    const int l32 = lane & 31, kg = lane >> 5;
    const float inv = a.w_scale2[1] * (a.x_scale2 ? a.x_scale2[1] : 1.f);
#ifdef MVIP_EXPERIMENT_GEMM
    if (a.dbg & 1) {
        float t = 0.f;
        for (int m = 0; m < MT; ++m) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) t += acc[m][j][r];
        if (t == 123.456f) a.y[0] = t;
        return;
    }
#endif
    // all loads of the epilogue are issued together, ahead of the arithmetic (see the convolution's epilogue)
    const bool hb = a.bias != nullptr, hc = a.chan_add != nullptr, hr = a.residual != nullptr;
    if (a.nsec > 0 && a.geglu_L == 0 && !a.partial) {       // (split-K: raw partial sums first, the sink is the reduce launch)
        // ---- operand sinks: this workgroup's MT row tiles lie in ONE section (sections are multiples of 64 rows) ----
        const int row0 = mb * MT * 32;
        GemmArgs::Sec sc = a.sec[0];                 // static indices only: a dynamically indexed kernel argument goes to scratch
        int sec_row0 = 0;
        if (a.nsec > 1 && row0 >= a.sec[0].row_end) { sc = a.sec[1]; sec_row0 = a.sec[0].row_end; }
        if (a.nsec > 2 && row0 >= a.sec[1].row_end) { sc = a.sec[2]; sec_row0 = a.sec[1].row_end; }
        {
            const int n_blk8 = (sc.row_end - sec_row0) / 8;
            char *pn = sc.ptr + (int64_t)n * (n_blk8 / 2) * 4 * a.P * 16;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                float bv[16];
                f32x16 rv[2];
                if (hr) {                                       // the residual stream joins before the split (loads issued together)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r)
                            rv[j][r] = a.residual[((int64_t)n * a.M + row0 + m * 32 + 8 * (r >> 2) + 4 * kg + (r & 3)) * a.P + p0 +
                                                  (2 * wave + j) * 32 + l32];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) bv[r] = hb ? a.bias[row0 + m * 32 + 8 * (r >> 2) + 4 * kg + (r & 3)] : 0.f;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    float v[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float t = acc[m][j][r] * inv;           // the fp32 epilogue's order of roundings, then the scale
                        if (hb) t += bv[r];
                        if (hr) t += rv[j][r];
                        v[r] = t * sc.scale;
                    }
                    sink_store_planes(pn, a.P, (row0 - sec_row0) / 8 + m * 4, n_blk8, p0 + (2 * wave + j) * 32 + l32, kg, v, true, a.prec != 1);
                }
            }
        }
        return;
    }
    if (MT == 2 && a.geglu_L > 0) {
        // feed-forward first projection: out = value * gelu(gate) (erf form), the two halves sit in this workgroup's
        // two row tiles; the [N][8C][P] intermediate never exists
        float ba[16], bg[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int r32 = 8 * (r >> 2) + 4 * kg + (r & 3);
            ba[r] = hb ? a.bias[(mb * 2 + 0) * 32 + r32] : 0.f;
            bg[r] = hb ? a.bias[(mb * 2 + 1) * 32 + r32] : 0.f;
        }
        if (a.nsec > 0) {
            // the product goes out as the second projection's operand planes, scaled by a power of two fixed before the
            // launch (|value * gelu(gate)| <= |value| |gate|): no absolute-maximum collection, no fp32 intermediate
            const int n_blk8 = (a.M / 2) / 8;
            char *pn = a.sec[0].ptr + (int64_t)n * (n_blk8 / 2) * 4 * a.P * 16;
            const float os = a.sec[0].scale;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t px = p0 + (2 * wave + j) * 32 + l32;
                float v[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float av = acc[0][j][r] * inv, gv = acc[MT - 1][j][r] * inv;
                    if (hb) { av += ba[r]; gv += bg[r]; }
                    const float t = av * (0.5f * gv * (1.0f + erff(gv * 0.70710678118654752f)));
                    v[r] = px >= a.geglu_L ? 0.f : t * os;
                }
                sink_store_planes(pn, a.P, mb * 4, n_blk8, px, kg, v, true, a.prec != 1);
            }
            return;
        }
        float mx = 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t px = p0 + (2 * wave + j) * 32 + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int r32 = 8 * (r >> 2) + 4 * kg + (r & 3);
                float av = acc[0][j][r] * inv, gv = acc[MT - 1][j][r] * inv;
                if (hb) { av += ba[r]; gv += bg[r]; }
                float v = av * (0.5f * gv * (1.0f + erff(gv * 0.70710678118654752f)));
                if (px >= a.geglu_L) v = 0.f;
                const float w = fabsf(v);
                mx = (w == w && w < 3.0e38f) ? fmaxf(mx, w) : mx;
                a.y[((int64_t)n * (a.M / 2) + mb * 32 + r32) * a.P + px] = v;
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if (lane == 0) atomicMax(a.absmax_bits, __float_as_uint(mx));
        return;
    }
    auto out_index = [&](int m, int j, int r, int &row) -> int64_t {
        row = (mb * MT + m) * 32 + 8 * (r >> 2) + 4 * kg + (r & 3);
        return ((int64_t)n * a.M + row) * a.P + p0 + (2 * wave + j) * 32 + l32;
    };
    if (a.partial) {
        float *pp = a.partial + (int64_t)split * a.N * a.M * a.P;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { int row; pp[out_index(m, j, r, row)] = acc[m][j][r]; }
        return;
    }
    float bv[MT][16], cv[MT][16];
    f32x16 rv[MT][2];
    if (hr) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { int row; rv[m][j][r] = a.residual[out_index(m, j, r, row)]; }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (mb * MT + m) * 32 + 8 * (r >> 2) + 4 * kg + (r & 3);
            bv[m][r] = hb ? a.bias[row] : 0.f;
            cv[m][r] = hc ? a.chan_add[(int64_t)n * a.M + row] : 0.f;
        }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row;
                const int64_t o = out_index(m, j, r, row);
                float v = acc[m][j][r] * inv;
                if (hb) v += bv[m][r];
                if (hc) v += cv[m][r];
                if (hr) v += rv[m][j][r];
                a.y[o] = v;
                if constexpr (LN) acc[m][j][r] = v;             // kept for the statistics below
            }
    if constexpr (LN) if (a.ln_part) {
        // a lane holds 16 MT rows of each of its two columns; the other half-wave (kg) holds the other 16 MT rows of the
        // same columns: in-register fp64 sums + one cross-half add, no LDS
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            double sm = 0.0, q = 0.0;
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const double t = (double)acc[m][j][r]; sm += t; q += t * t; }
            sm += __shfl_xor(sm, 32, 64);
            q += __shfl_xor(q, 32, 64);
            if (kg == 0) {
                const int64_t p = p0 + (2 * wave + j) * 32 + l32;
                double *dst = a.ln_part + (((int64_t)n * a.MB + mb) * 2) * a.P + p;
                dst[0] = sm;
                dst[a.P] = q;
            }
        }
    }
}

template <int MT>
__global__ void __launch_bounds__(256, (MT <= 2 ? 2 : 1)) gemm_f16x3_kernel(const GemmArgs a) {
    constexpr int WB = GM_KC * MT * 2 * 1024;
    constexpr int IB = GM_KC * 4 * GM_PIX * 16;
    __shared__ __attribute__((aligned(16))) char lds[2 * WB + 2 * IB];
    char *lds_w = lds, *lds_in = lds + 2 * WB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, kg = lane >> 5;
    int id = blockIdx.x;
    const int total = gridDim.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int split = id % a.splits;
    id /= a.splits;
    const int mb = id % a.MB;
    const int tile = id / a.MB;
    const int tp = tile % a.tiles;
    const int n = tile / a.tiles;
    const int64_t p0 = (int64_t)tp * GM_PIX;
    const int64_t plane = a.P * 16;
    const int s0 = split * a.sks;                       // first stage (of GM_KC chunks) of this workgroup
    const char *xs_n = a.xs + ((int64_t)n * a.CK + s0 * GM_KC) * 4 * plane + (p0 + tid) * 16;

    auto issue_input = [&](int s, int buf) {
#pragma unroll
        for (int kc = 0; kc < GM_KC; ++kc)
#pragma unroll
            for (int piece = 0; piece < 4; ++piece)
                glds16b(xs_n + ((int64_t)(s * GM_KC + kc) * 4 + piece) * plane,
                        lds_in + buf * IB + ((kc * 4 + piece) * GM_PIX + wave * 64) * 16);
    };
    auto issue_weights = [&](int s, int buf) {
#pragma unroll
        for (int b0 = 0; b0 < GM_KC * MT * 2; b0 += 4) {       // LDS block b = (kc*MT + m)*2 + hl
            const int b = b0 + wave;
            if (b < GM_KC * MT * 2) {
                const int hl = b & 1, m = (b >> 1) % MT, kc = (b >> 1) / MT;
                glds16b(a.wp + ((((int64_t)(mb * MT + m) * a.CK + (s0 + s) * GM_KC + kc) * 2 + hl) * 1024) + lane * 16,
                        lds_w + buf * WB + b * 1024);
            }
        }
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

    const int nall = a.CK / GM_KC;
    const int nstage = (nall - s0 < a.sks) ? nall - s0 : a.sks;
    issue_input(0, 0);
    issue_weights(0, 0);
    for (int s = 0; s < nstage; ++s) {
        __syncthreads();
        if (s + 1 < nstage) { issue_weights(s + 1, (s + 1) & 1); issue_input(s + 1, (s + 1) & 1); }
        const char *inb = lds_in + (s & 1) * IB;
        const char *wb = lds_w + (s & 1) * WB + lane * 16;
#pragma unroll
        for (int kc = 0; kc < GM_KC; ++kc) {
            h16x8 bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = (2 * wave + j) * 32 + l32;
                bh[j] = *reinterpret_cast<const h16x8 *>(inb + ((kc * 4 + kg * 2 + 0) * GM_PIX + p) * 16);
                bl[j] = *reinterpret_cast<const h16x8 *>(inb + ((kc * 4 + kg * 2 + 1) * GM_PIX + p) * 16);
            }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const h16x8 ah = *reinterpret_cast<const h16x8 *>(wb + ((kc * MT + m) * 2 + 0) * 1024);
                const h16x8 al = *reinterpret_cast<const h16x8 *>(wb + ((kc * MT + m) * 2 + 1) * 1024);
#ifdef MVIP_EXPERIMENT_GEMM
                if (a.dbg & 2) { acc[m][0][0] += (float)ah[0] + (float)bh[0][0] + (float)bl[1][0] + (float)al[0] + (float)bh[1][0] + (float)bl[0][0]; continue; }
#endif
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[m][j], 0, 0, 0);
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], acc[m][j], 0, 0, 0);
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], acc[m][j], 0, 0, 0);
                }
            }
        }
    }
    gemm_epilogue<MT>(a, acc, n, mb, p0, split, wave, lane);
}

// ---- the same GEMM with the B operand streamed straight into registers ----------------------------------------------
// The kernel above is bound by operand delivery: 36 KB of DMA per 12 MFMAs and workgroup (94 B/clk/CU at the matrix rate)
// with ONE stage in flight, where L2 -> CU delivery needs ~64 KB in flight per CU to reach its ~57 B/clk
// (tools/micro/l2_load_bw.hip) -- PMC: matrix pipe 12 % busy, 54 % of the wave time in s_waitcnt.  Here
//  * the B operand (activations: no reuse between the four waves, each owns 64 columns) never touches LDS: a lane's
//    fragment is 16 contiguous bytes of a split plane, so the MFMA operand registers are loaded directly
//    (global_load_dwordx4, 512 B contiguous per half-wave), SIX 16-k chunks (24 loads, 24 KB per wave) ahead of their use;
//  * the A operand (weights, shared by the four waves) streams through a three-slot LDS ring of 64-k chunks by LDS-DMA,
//    two chunks ahead; the only synchronisation is one s_barrier per chunk WITHOUT a vmcnt(0) drain: loads return in
//    order, and the wave has by then waited for B fragments that were issued after the chunk's DMA (DB <= 7), so its
//    share of that DMA has landed.
// Same grid, operand formats, split-K and epilogue as gemm_f16x3_kernel.
template <int V> struct cic { static constexpr int value = V; };
template <int N, class F, int I = 0>
__device__ __forceinline__ void cv_static_for(F &&f) {
    if constexpr (I < N) { f(cic<I>{}); cv_static_for<N, F, I + 1>(static_cast<F &&>(f)); }
}
constexpr int G5_DB = 6;        // B prefetch depth in 16-k chunks
constexpr int G5_CA = 4;        // 16-k chunks per A ring slot
template <int MT, bool SWAP = false, int NP = 3>
__global__ void __launch_bounds__(256, MT == 1 ? 3 : 2) gemm5_f16x3_kernel(const GemmArgs a) {
    constexpr bool F16 = NP == 1;
    constexpr bool AHI = NP < 3;                            // the weights' hi fragments only (NP 2: the lo ones are zero)
    constexpr int NHL = F16 ? 1 : 2;                        // halves of the B (activation) operand fetched per fragment
    constexpr int NHA = AHI ? 1 : 2;                        // halves of the A (weight) operand fetched per fragment
    constexpr int SLOT = G5_CA * MT * 2 * 1024;            // bytes of a ring slot: [u][m][hl][lane][16 B]
    // three separate arrays, not one: the compiler's wait-count pass then knows that an LDS-DMA into one slot cannot
    // alias the fragment reads of another and puts no vmcnt(0) in front of them
    __shared__ __attribute__((aligned(16))) char ring0[SLOT];
    __shared__ __attribute__((aligned(16))) char ring1[SLOT];
    __shared__ __attribute__((aligned(16))) char ring2[SLOT];
    auto ring = [&](auto slot_) -> char * {
        constexpr int slot = decltype(slot_)::value;
        if constexpr (slot == 0) return ring0;
        else if constexpr (slot == 1) return ring1;
        else return ring2;
    };
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l32 = lane & 31, kg = lane >> 5;
    int id = blockIdx.x;
    const int total = gridDim.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int split = id % a.splits;
    id /= a.splits;
    const int mb = id % a.MB;
    const int tile = id / a.MB;
    const int tp = tile % a.tiles;
    const int n = tile / a.tiles;
    const int64_t p0 = (int64_t)tp * GM_PIX;
    const int64_t plane = a.P * 16;
    const int ck0 = split * a.sks * GM_KC;                 // first 16-k chunk of this workgroup
    const int nall = a.CK - ck0;
    const int nck = nall < a.sks * GM_KC ? nall : a.sks * GM_KC;
    const int nchunk = (nck + G5_CA - 1) / G5_CA;

    // B: this lane's fragment of column block j, hi / lo, for chunk ck.  The address is split into a WAVE-UNIFORM 64-bit
    // part (sample, chunk, plane, column tile, wave: scalar registers, scalar arithmetic) and ONE loop-invariant 32-bit
    // per-lane offset (kg and the lane's column), i.e. the saddr form of global_load_dwordx4 -- the compiler otherwise keeps
    // a 64-bit VGPR pointer per load of the unrolled trip (48 of them) and the two-product instantiation spilled.
    const char *bbase = a.xs + (((int64_t)n * a.CK + ck0) * 4) * plane + (p0 + (int64_t)wave * 64) * 16;
    const unsigned bvoff = (unsigned)(kg * 2) * (unsigned)plane + (unsigned)l32 * 16u;     // < 2^32: P <= 2^26 columns
    h16x8 Bq[G5_DB][2][NHL];
    auto load_b = [&](int ck, auto slot_) {
        constexpr int slot = decltype(slot_)::value;
        const char *cb = bbase + (int64_t)ck * 4 * plane;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int hl = 0; hl < NHL; ++hl)
                Bq[slot][j][hl] = *reinterpret_cast<const h16x8 *>(cb + (int64_t)hl * plane + j * 512 + bvoff);
    };
    // A: ring slot `slot` <- 16-k chunks c*CA .. c*CA+CA-1 of the MT row blocks; piece q = (u*MT + m)*2 + hl.  The slot
    // is a compile-time constant so that the DMA provably does not alias the fragment reads of the other slots.
    auto issue_a = [&](int c, auto slot_, bool always = false) {
        constexpr int slot = decltype(slot_)::value;
        if (always || c < nchunk) {
#pragma unroll
            for (int q0 = 0; q0 < G5_CA * MT * NHA; q0 += 4) {
                const int qq = q0 + wave;                    // AHI: the hi pieces only, spread over the four waves
                const int q = AHI ? qq * 2 : qq;
                const int hl = q & 1, m = (q >> 1) % MT, u = (q >> 1) / MT;
                int ck = c * G5_CA + u;
                if (ck >= nck) ck = nck - 1;                 // partial last chunk: a valid address, never multiplied
                if (!AHI || qq < G5_CA * MT)
                    glds16b(a.wp + ((((int64_t)(mb * MT + m) * a.CK + ck0 + ck) * 2 + hl) * 1024) + lane * 16,
                            ring(slot_) + q * 1024);
            }
        }
    };

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

    issue_a(0, cic<0>{});
    issue_a(1, cic<1>{});
    cv_static_for<G5_DB>([&](auto d) { if (d.value < nck) load_b(d.value, d); });
    // ring slots 0 and 1 are in LDS once this wave's loads issued before B(0) have returned and every wave says so
    if (nck >= G5_DB) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NHL * (G5_DB - 1)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_barrier" ::: "memory");

    // 12 chunks of 16 k = 3 ring slots = 2 rounds of the B registers per trip, everything indexed statically.  The main
    // loop has NO branches (every trip issues all of its loads and DMAs): the compiler's wait-count pass then keeps exact
    // counts (s_waitcnt vmcnt(20): the 20 younger B loads stay in flight); with a branch in the body it merges the
    // pending-load states into vmcnt(0) and drains the queue.  The last trips run the guarded copy of the same steps.
    // SWAP (operand sink of kind 2: attention V fragments; its own launch): the MFMA operands trade places, so the
    // accumulators hold the TRANSPOSED tile -- lane = row of the weight tile, registers = the 32 tokens of column block j
    // -- which is the key order of the attention kernel's V fragments; same three products, same roundings per element.
    auto step = [&](int base, auto t_, auto guarded_) {
        constexpr int t = decltype(t_)::value;
        constexpr bool guarded = decltype(guarded_)::value != 0;
        constexpr int slot = t / G5_CA, u = t % G5_CA, bs = t % G5_DB;
        const int ck = base + t;
        if (guarded && ck >= nck) return;
        if constexpr (u == 0) issue_a(ck / G5_CA + 2, cic<(slot + 2) % 3>{}, !guarded);   // main loop: chunk +2 exists
        const char *ab = ring(cic<slot>{}) + lane * 16;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const h16x8 ah = *reinterpret_cast<const h16x8 *>(ab + ((u * MT + m) * 2 + 0) * 1024);
            h16x8 al;
            if constexpr (NP == 3) al = *reinterpret_cast<const h16x8 *>(ab + ((u * MT + m) * 2 + 1) * 1024);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // products in the order hi.hi, hi.lo (NP >= 2), lo.hi (NP == 3: with NP == 2 the weights' lo half is zero
                // and this product would add exact zeros); SWAP trades the MFMA operands (transposed accumulator)
                if constexpr (SWAP) {
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bq[bs][j][0], ah, acc[m][j], 0, 0, 0);
                    if constexpr (NP >= 2) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bq[bs][j][NHL - 1], ah, acc[m][j], 0, 0, 0);
                    if constexpr (NP == 3) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Bq[bs][j][0], al, acc[m][j], 0, 0, 0);
                } else {
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bq[bs][j][0], acc[m][j], 0, 0, 0);
                    if constexpr (NP >= 2) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, Bq[bs][j][NHL - 1], acc[m][j], 0, 0, 0);
                    if constexpr (NP == 3) acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, Bq[bs][j][0], acc[m][j], 0, 0, 0);
                }
            }
        }
        // the scheduler otherwise gathers the loads of several steps into one cluster late in the chunk, which shortens
        // the prefetch distance and turns the counted waits into vmcnt(0)
        __builtin_amdgcn_sched_barrier(0);
        if (!guarded || ck + G5_DB < nck) load_b(ck + G5_DB, cic<bs>{});
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (u == G5_CA - 1) asm volatile("s_barrier" ::: "memory");   // slot free for chunk +3, chunk +1 landed
    };
    int base = 0;
    for (; base + 12 + G5_DB <= nck; base += 12) cv_static_for<12>([&](auto t_) { step(base, t_, cic<0>{}); });
    for (; base < nck; base += 12) cv_static_for<12>([&](auto t_) { step(base, t_, cic<1>{}); });
    if constexpr (SWAP) gemm_epilogue_vfrag<MT>(a, acc, n, mb, p0, wave, lane);
    else
    gemm_epilogue<MT, true>(a, acc, n, mb, p0, split, wave, lane);
}

// scale2 = {2^k, 2^-k} from the maximum collected in *bits, which is left zero (caller-owned scratch word)
__global__ void gm_scale_from_bits_kernel(float *__restrict__ scale2, unsigned *__restrict__ bits) {
    const float m = __uint_as_float(*bits);
    *bits = 0u;
    float s = 1.f;
    if (m > 0.f && m < 3.0e38f) {
        int e;
        frexpf(m, &e);
        int k = 10 - e;
        k = k > 60 ? 60 : (k < -60 ? -60 : k);
        s = ldexpf(1.f, k);
    }
    scale2[0] = s;
    scale2[1] = 1.f / s;
}

// ---- the same GEMM with a square-ish workgroup tile ---------------------------------------------------------
// At split-precision MFMA rates the kernel above is bound by L2 -> LDS operand traffic, not by the matrix pipe: a
// workgroup tile of Tm x Tp needs (Tm + Tp) / (Tm Tp) x 2731 B/clk/CU of operands to keep the pipe busy, i.e. 96
// B/clk for 32 x 256 and 53 B/clk for 64 x 256 against ~55 B/clk/CU of L2 bandwidth.  Here a workgroup owns
// (WGM*64) x (WGP*64) outputs (128 x 256: 32 B/clk), every wave a 64 x 64 block (2 x 2 MFMA tiles, 12 MFMAs per
// eight fragment reads), operands of one 32-deep k-stage land by DMA NST-1 stages ahead of their use and the wave
// waits with a COUNTED vmcnt for exactly the stage it is about to read.
template <int WGM, int WGP, int NST>
__global__ void __launch_bounds__(WGM * WGP * 64) gemm2_f16x3_kernel(const GemmArgs a) {
    constexpr int NW = WGM * WGP;
    constexpr int TP = WGP * 64;
    constexpr int AB = WGM * 2 * GM_KC * 2 * 1024;          // A stage: [m-tile][kc][hl][lane][16 B]
    constexpr int BB = GM_KC * 4 * TP * 16;                 // B stage: [kc][kg*2+hl][column][16 B]
    constexpr int NAP = AB / 1024, NBP = BB / 1024;
    constexpr int PPW = (NAP + NBP) / NW;                   // DMA pieces per wave per stage
    static_assert((NAP + NBP) % NW == 0, "pieces must divide evenly over the waves (counted vmcnt)");
    constexpr int AHEAD = NST - 1;
    constexpr int WAITN = PPW * (AHEAD - 1);                // pieces that may still be in flight when a stage is read
    static_assert(WAITN < 64, "vmcnt field");
    __shared__ __attribute__((aligned(16))) char lds[NST * (AB + BB)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, kg = lane >> 5;
    const int wm = wave % WGM, wp = wave / WGM;
    int id = blockIdx.x;
    const int total = gridDim.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int mb = id % a.MB;
    const int tile = id / a.MB;
    const int tp = tile % a.tiles;
    const int n = tile / a.tiles;
    const int64_t p0 = (int64_t)tp * TP;
    const int64_t plane = a.P * 16;
    const char *xs_n = a.xs + (int64_t)n * a.CK * 4 * plane + (p0 + lane) * 16;
    const char *wp_b = a.wp + (int64_t)mb * (WGM * 2) * a.CK * 2048 + lane * 16;

    auto issue = [&](int s, int buf) {
        char *ad = lds + buf * (AB + BB), *bd = ad + AB;
#pragma unroll
        for (int q0 = 0; q0 < NAP + NBP; q0 += NW) {
            const int q = q0 + wave;
            if (q < NAP) {                                  // q = (mt * GM_KC + kc) * 2 + hl
                const int hl = q & 1, kc = (q >> 1) % GM_KC, mt = (q >> 1) / GM_KC;
                glds16b(wp_b + (((int64_t)mt * a.CK + s * GM_KC + kc) * 2 + hl) * 1024, ad + q * 1024);
            } else {                                        // r = (kc * 4 + piece) * WGP + column block
                const int r = q - NAP;
                const int cb = r % WGP, pl = r / WGP;       // pl = kc * 4 + (kg * 2 + hl)
                glds16b(xs_n + ((int64_t)(s * GM_KC * 4 + pl)) * plane + cb * 1024, bd + (pl * TP + cb * 64) * 16);
            }
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;

    const int nstage = a.CK / GM_KC;
#pragma unroll
    for (int s = 0; s < AHEAD; ++s)
        if (s < nstage) issue(s, s);
    for (int s = 0; s < nstage; ++s) {
        // stage s has landed once at most the pieces of the AHEAD-1 younger stages are outstanding
        if (s + AHEAD - 1 < nstage) __builtin_amdgcn_s_waitcnt(0x0f70 | (WAITN & 15) | ((WAITN >> 4) << 14));
        else __builtin_amdgcn_s_waitcnt(0x0f70);
        __syncthreads();                                   // everyone's pieces; everyone is done with stage s-1
        if (s + AHEAD < nstage) issue(s + AHEAD, (s + AHEAD) % NST);
        const char *ab = lds + (s % NST) * (AB + BB) + lane * 16, *bb = lds + (s % NST) * (AB + BB) + AB;
#pragma unroll
        for (int kc = 0; kc < GM_KC; ++kc) {
            h16x8 bh[2], bl[2], ah[2], al[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int col = (wp * 2 + j) * 32 + l32;
                bh[j] = *reinterpret_cast<const h16x8 *>(bb + ((kc * 4 + kg * 2 + 0) * TP + col) * 16);
                bl[j] = *reinterpret_cast<const h16x8 *>(bb + ((kc * 4 + kg * 2 + 1) * TP + col) * 16);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ah[m] = *reinterpret_cast<const h16x8 *>(ab + (((wm * 2 + m) * GM_KC + kc) * 2 + 0) * 1024);
                al[m] = *reinterpret_cast<const h16x8 *>(ab + (((wm * 2 + m) * GM_KC + kc) * 2 + 1) * 1024);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bh[j], acc[m][j], 0, 0, 0);
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[m], bl[j], acc[m][j], 0, 0, 0);
                    acc[m][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[m], bh[j], acc[m][j], 0, 0, 0);
                }
        }
    }
    const float inv = a.w_scale2[1] * (a.x_scale2 ? a.x_scale2[1] : 1.f);
    // loads of the epilogue issued together, ahead of the arithmetic (see the convolution's epilogue)
    const bool hb = a.bias != nullptr, hc = a.chan_add != nullptr, hr = a.residual != nullptr;
    auto out_index = [&](int m, int j, int r, int &row) -> int64_t {
        row = ((mb * WGM + wm) * 2 + m) * 32 + 8 * (r >> 2) + 4 * kg + (r & 3);
        return ((int64_t)n * a.M + row) * a.P + p0 + (wp * 2 + j) * 32 + l32;
    };
    float bv[2][16], cv[2][16];
    f32x16 rv[2][2];
    if (hr) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) { int row; rv[m][j][r] = a.residual[out_index(m, j, r, row)]; }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = ((mb * WGM + wm) * 2 + m) * 32 + 8 * (r >> 2) + 4 * kg + (r & 3);
            bv[m][r] = hb ? a.bias[row] : 0.f;
            cv[m][r] = hc ? a.chan_add[(int64_t)n * a.M + row] : 0.f;
        }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                int row;
                const int64_t o = out_index(m, j, r, row);
                float v = acc[m][j][r] * inv;
                if (hb) v += bv[m][r];
                if (hc) v += cv[m][r];
                if (hr) v += rv[m][j][r];
                a.y[o] = v;
            }
}

// largest MT in {4, 2, 1} dividing Cout/32 whose grid still has >= 256 workgroups (else the smallest)
static inline int cv_mt(int64_t Cout, int64_t tiles) {
    if (Cout % 32 != 0) return 0;
    const int64_t rows = Cout / 32;
    for (int mt = 2; mt > 1; mt >>= 1)                 // MT = 4 (one workgroup per CU) measured slower than two MT = 2
        if (rows % mt == 0 && tiles * (rows / mt) >= 512) return mt;
    return 1;
}

}  // namespace mvip

using namespace mvip;

extern "C" int mvip_conv3x3_supported(int64_t Cout, int64_t Cin, int64_t H, int64_t W) {
    return Cout > 0 && Cout % 32 == 0 && Cin > 0 && Cin % 16 == 0 && H > 0 && W > 0 &&
           ((H % CV_TH == 0 && W % CV_TW == 0) || (H % 16 == 0 && W % 16 == 0) || (H == 8 && W == 8)) &&
           H * W <= (1 << 24);
}

extern "C" int64_t mvip_conv3x3_packed_bytes(int64_t Cout, int64_t Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return Cout * Cin * 9 * 4 + 256;       // tail: {scale, 1/scale, absmax bits}, then a 16-byte zero page
}

// tail layout (byte offsets from Cout*Cin*36): 0 scale, 4 1/scale, 8 absmax bits, 12 "a lo fragment is non-zero", 128..143 zeros
extern "C" int mvip_conv3x3_pack(const float *weight, int64_t Cout, int64_t Cin, int transpose, void *packed,
                                 void *stream) {
    const int64_t co = transpose ? Cin : Cout, ci = transpose ? Cout : Cin;      // operator dims
    if (!weight || !packed || co <= 0 || co % 32 != 0 || ci <= 0 || ci % 16 != 0) return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
    char *tail = (char *)packed + co * ci * 36;
    zero_words(tail, 64, st);
    hipLaunchKernelGGL(cv_absmax_kernel, dim3(absmax_blocks(Cout * Cin * 9)), dim3(256), 0, st, weight, Cout * Cin * 9,
                       (unsigned *)(tail + 8));
    hipLaunchKernelGGL(cv_scale_kernel, dim3(1), dim3(1), 0, st, (const unsigned *)(tail + 8), (float *)tail);
    const int64_t total = (co / 32) * (ci / 16) * 9 * 2 * 64;
    hipLaunchKernelGGL(cv_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, weight, (int)co, (int)ci,
                       transpose, (const float *)tail, (uint4 *)packed, (unsigned *)(tail + 12));
    return check_launch();
}

// 1 in *two_product_host when every lo fragment of a packed weight image is zero, i.e. the scaled weights are exact fp16
// values and `prec = 2` (two products) gives the three-product result bit for bit.  `fragment_bytes` = the image's size
// without its 256-byte tail (Cout * Cin * 36 for mvip_conv3x3_pack, M * K * 4 for mvip_gemm_pack_a).  Reads ONE word back
// to the host after synchronising `stream`: a pack-time query, not for the step.
extern "C" int mvip_packed_weights_two_product(const void *packed, int64_t fragment_bytes, int *two_product_host, void *stream) {
    if (!packed || !two_product_host || fragment_bytes <= 0) return MVIP_EINVAL;
    unsigned flag = 1u;
    hipStream_t st = as_stream(stream);
    hipError_t e = hipMemcpyAsync(&flag, (const char *)packed + fragment_bytes + 12, sizeof(flag), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { set_last_error(e); return MVIP_ELAUNCH; }
    *two_product_host = flag == 0u ? 1 : 0;
    return MVIP_OK;
}

extern "C" int mvip_absmax_scale(const float *x, int64_t n, float *scale2, void *zero_words2, void *stream) {
    if (n < 0 || !scale2 || (n > 0 && !x)) return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
    if (zero_words2 && n > 0) {                 // one launch: caller-owned scratch words that are zero between calls
        hipLaunchKernelGGL(cv_absmax_scale_kernel, dim3(absmax_blocks(n)), dim3(256), 0, st, x, n, (unsigned *)zero_words2,
                           scale2);
        return check_launch();
    }
    zero_words(scale2, 4, st);
    if (n > 0) hipLaunchKernelGGL(cv_absmax_kernel, dim3(absmax_blocks(n)), dim3(256), 0, st, x, n, (unsigned *)(scale2 + 2));
    hipLaunchKernelGGL(cv_scale_kernel, dim3(1), dim3(1), 0, st, (const unsigned *)(scale2 + 2), scale2);
    return check_launch();
}

extern "C" int mvip_split_planes(const float *x, int64_t N, int64_t C, int64_t HW, const float *scale2, void *xs, int prec,
                                 void *stream) {
    if (N < 0 || C <= 0 || C % 16 != 0 || HW < 0 || (prec < 0 || prec > 2)) return MVIP_EINVAL;
    if (N == 0 || HW == 0) return MVIP_OK;
    if (!x || !xs || N * (C / 16) > 65535) return MVIP_EINVAL;
    const dim3 grid((unsigned)((HW + 255) / 256), (unsigned)(N * (C / 16)));
    hipLaunchKernelGGL((cv_to_split_kernel<false, false>), grid, dim3(256), 0, as_stream(stream), x, nullptr, nullptr,
                       nullptr, nullptr, scale2, (int)C, HW, 1, (uint4 *)xs, C * HW, HW, (int64_t)1, prec);
    return check_launch();
}

extern "C" int mvip_split_planes_upsample2(const float *x, int64_t N, int64_t C, int64_t H, int64_t W, const float *scale2,
                                           void *xs, int prec, void *stream) {
    if (N < 0 || C <= 0 || C % 16 != 0 || H <= 0 || W <= 0 || H * W > (1 << 26) || (prec < 0 || prec > 2)) return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    if (!x || !xs || N * (C / 16) > 65535) return MVIP_EINVAL;
    const dim3 grid((unsigned)((4 * H * W + 255) / 256), (unsigned)(N * (C / 16)));
    hipLaunchKernelGGL(cv_to_split_up2_kernel, grid, dim3(256), 0, as_stream(stream), x, scale2, (int)C, (int)H, (int)W,
                       (uint4 *)xs, prec);
    return check_launch();
}

extern "C" int mvip_im2col_split_planes(const float *x, int64_t N, int64_t Cin, int64_t H, int64_t W, int KH, int KW,
                                        int stride, int pad_top, int pad_left, int64_t OH, int64_t OW, int64_t KP,
                                        int64_t PP, const float *scale2, void *xs, int prec, void *stream) {
    if ((prec < 0 || prec > 2) || N < 0 || Cin <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad_top < 0 || pad_left < 0 ||
        OH <= 0 || OW <= 0 || KP <= 0 || KP % 16 != 0 || KP < Cin * KH * KW || PP < OH * OW || H * W > (1 << 30) ||
        OH * OW > (1 << 30))
        return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    if (!x || !xs || N * (KP / 16) > 65535) return MVIP_EINVAL;
    const dim3 grid((unsigned)((PP + 255) / 256), (unsigned)(N * (KP / 16)));
#define MVIP_I2C(A, B) hipLaunchKernelGGL((cv_im2col_split_kernel<A, B>), grid, dim3(256), 0, as_stream(stream), x, (int)Cin, (int)H, \
                                          (int)W, KH, KW, stride, pad_top, pad_left, (int)OH, (int)OW, (int)KP, PP, scale2,      \
                                          (uint4 *)xs, prec)
    if (KH == 3 && KW == 3) MVIP_I2C(3, 3); else if (KH == 1 && KW == 1) MVIP_I2C(1, 1); else MVIP_I2C(0, 0);
#undef MVIP_I2C
    return check_launch();
}

extern "C" int mvip_col2im(const float *col, int64_t N, int64_t Cin, int64_t H, int64_t W, int KH, int KW, int stride,
                           int pad_top, int pad_left, int64_t OH, int64_t OW, int64_t KP, int64_t PP, float *dx,
                           void *stream) {
    if (N < 0 || Cin <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad_top < 0 || pad_left < 0 ||
        OH <= 0 || OW <= 0 || KP < Cin * KH * KW || PP < OH * OW || H * W > (1 << 30) || OH * OW > (1 << 30))
        return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    if (!col || !dx) return MVIP_EINVAL;
    if (N * Cin > 65535) return MVIP_EINVAL;
    const dim3 grid((unsigned)((H * W + 255) / 256), (unsigned)(N * Cin));
#define MVIP_C2I(A, B, C) hipLaunchKernelGGL((cv_col2im_kernel<A, B, C>), grid, dim3(256), 0, as_stream(stream), col, (int)Cin, (int)H, \
                                             (int)W, KH, KW, stride, pad_top, pad_left, (int)OH, (int)OW, (int)KP, PP, dx)
    if (KH == 3 && KW == 3 && stride == 2) MVIP_C2I(3, 3, 2);
    else if (KH == 3 && KW == 3 && stride == 1) MVIP_C2I(3, 3, 1);
    else if (KH == 1 && KW == 1 && stride == 1) MVIP_C2I(1, 1, 1);
    else MVIP_C2I(0, 0, 0);
#undef MVIP_C2I
    return check_launch();
}

extern "C" int mvip_split_planes_strided(const float *x, int64_t N, int64_t C, int64_t HW, int64_t sn, int64_t sc,
                                         int64_t sp, const float *scale2, void *xs, int prec, void *stream) {
    if (N < 0 || C <= 0 || C % 16 != 0 || HW < 0 || (prec < 0 || prec > 2)) return MVIP_EINVAL;
    if (N == 0 || HW == 0) return MVIP_OK;
    if (!x || !xs || N * (C / 16) > 65535) return MVIP_EINVAL;
    const dim3 grid((unsigned)((HW + 255) / 256), (unsigned)(N * (C / 16)));
    hipLaunchKernelGGL((cv_to_split_kernel<false, false>), grid, dim3(256), 0, as_stream(stream), x, nullptr, nullptr,
                       nullptr, nullptr, scale2, (int)C, HW, 1, (uint4 *)xs, sn, sc, sp, prec);
    return check_launch();
}

extern "C" int mvip_groupnorm_split_planes(const float *x, const float *gamma, const float *beta, const float *mean,
                                           const float *rstd, int64_t N, int64_t C, int64_t HW, int G, int silu,
                                           void *xs, int prec, void *stream) {
    if (N < 0 || C <= 0 || C % 16 != 0 || HW < 0 || G <= 0 || C % G != 0 || (prec < 0 || prec > 2)) return MVIP_EINVAL;
    if (N == 0 || HW == 0) return MVIP_OK;
    if (!x || !xs || !mean || !rstd || N * (C / 16) > 65535) return MVIP_EINVAL;
    const dim3 grid((unsigned)((HW + 255) / 256), (unsigned)(N * (C / 16)));
    if (silu)
        hipLaunchKernelGGL((cv_to_split_kernel<true, true>), grid, dim3(256), 0, as_stream(stream), x, gamma, beta, mean,
                           rstd, nullptr, (int)C, HW, (int)(C / G), (uint4 *)xs, C * HW, HW, (int64_t)1, prec);
    else
        hipLaunchKernelGGL((cv_to_split_kernel<true, false>), grid, dim3(256), 0, as_stream(stream), x, gamma, beta, mean,
                           rstd, nullptr, (int)C, HW, (int)(C / G), (uint4 *)xs, C * HW, HW, (int64_t)1, prec);
    return check_launch();
}

extern "C" int mvip_groupnorm_split_planes_moments(const float *x, const float *gamma, const float *beta,
                                                   const void *moments, float eps, int64_t N, int64_t C, int64_t HW, int G,
                                                   int silu, void *xs, int prec, void *stream) {
    if (N < 0 || C <= 0 || C % 16 != 0 || HW < 0 || G <= 0 || C % G != 0 || (prec < 0 || prec > 2)) return MVIP_EINVAL;
    if (N == 0 || HW == 0) return MVIP_OK;
    if (!x || !xs || !moments || N * (C / 16) > 65535 || C / G < 4) return MVIP_EINVAL;          // <= 5 groups per 16 channels
    const int chunks = (int)(mvip_groupnorm_workspace_bytes(1, 1, HW) / 16);
    const dim3 grid((unsigned)((HW + 255) / 256), (unsigned)(N * (C / 16)));
    if (silu)
        hipLaunchKernelGGL((cv_to_split_kernel<true, true>), grid, dim3(256), 0, as_stream(stream), x, gamma, beta, nullptr,
                           nullptr, nullptr, (int)C, HW, (int)(C / G), (uint4 *)xs, C * HW, HW, (int64_t)1, prec,
                           (const double *)moments, chunks, eps);
    else
        hipLaunchKernelGGL((cv_to_split_kernel<true, false>), grid, dim3(256), 0, as_stream(stream), x, gamma, beta, nullptr,
                           nullptr, nullptr, (int)C, HW, (int)(C / G), (uint4 *)xs, C * HW, HW, (int64_t)1, prec,
                           (const double *)moments, chunks, eps);
    return check_launch();
}

// mvip_groupnorm_split_planes_moments that ALSO writes mean / rstd [N, G] (bit-identical to mvip_groupnorm_stats' own): the
// forward of a layer whose backward needs the statistics gets them from the plane writer, one launch less per GroupNorm.
extern "C" int mvip_groupnorm_split_planes_moments_out(const float *x, const float *gamma, const float *beta,
                                                       const void *moments, float eps, int64_t N, int64_t C, int64_t HW, int G,
                                                       int silu, void *xs, float *mean_out, float *rstd_out, int prec,
                                                       void *stream) {
    if (N < 0 || C <= 0 || C % 16 != 0 || HW < 0 || G <= 0 || C % G != 0 || (prec < 0 || prec > 2)) return MVIP_EINVAL;
    if (N == 0 || HW == 0) return MVIP_OK;
    if (!x || !xs || !moments || !mean_out || !rstd_out || N * (C / 16) > 65535 || C / G < 4) return MVIP_EINVAL;
    const int chunks = (int)(mvip_groupnorm_workspace_bytes(1, 1, HW) / 16);
    const dim3 grid((unsigned)((HW + 255) / 256), (unsigned)(N * (C / 16)));
    if (silu)
        hipLaunchKernelGGL((cv_to_split_kernel<true, true>), grid, dim3(256), 0, as_stream(stream), x, gamma, beta, nullptr,
                           nullptr, nullptr, (int)C, HW, (int)(C / G), (uint4 *)xs, C * HW, HW, (int64_t)1, prec,
                           (const double *)moments, chunks, eps, mean_out, rstd_out);
    else
        hipLaunchKernelGGL((cv_to_split_kernel<true, false>), grid, dim3(256), 0, as_stream(stream), x, gamma, beta, nullptr,
                           nullptr, nullptr, (int)C, HW, (int)(C / G), (uint4 *)xs, C * HW, HW, (int64_t)1, prec,
                           (const double *)moments, chunks, eps, mean_out, rstd_out);
    return check_launch();
}

// Number of channel splits for a launch whose unsplit grid has `blocks` workgroups: fill ~2 workgroups per CU, keep at
// least 4 channel chunks (12 stages) per workgroup so the pipeline prologue stays small.
static inline int cv_splits(int64_t blocks, int64_t CK) {
    static const int forced = [] { const char *e = getenv("MVIP_CONV_SPLITS"); return e ? atoi(e) : 0; }();   // tuning
    // the chip holds 512 of these workgroups at a time: as many splits as keep the launch within ONE round (the probe of
    // tools/conv_probe.py on 640 channels at 32 x 32: 560 workgroups = 1.09 rounds took 72 us, the median workgroup 42)
    int64_t s = forced > 0 ? forced : (blocks >= 256 ? 1 : 512 / blocks);
    if (s > CK / 4) s = CK / 4;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    const int64_t cks = (CK + s - 1) / s;
    return (int)((CK + cks - 1) / cks);                // no empty split
}
static inline void cv_geometry(int64_t N, int64_t Cout, int64_t H, int64_t W, int &tw, int &MT, int64_t &blocks) {
    tw = (W % CV_TW == 0 && H % CV_TH == 0) ? CV_TW : (H == 8 && W == 8 ? 8 : 16);
    const int th = 256 / tw;
    MT = tw != CV_TW ? 1 : cv_mt(Cout, N * (W / tw) * (H / th));
    // 64-row workgroups also on small grids (the fragment reads and the input DMA of a stage serve twice the MFMAs); the
    // channel splits fill the chip instead of the row blocks: 12.0 -> 11.3 ms of convolutions per SDS step, +0.3 ms of
    // split reduction.  MVIP_CONV_MT2=0 restores the 32-row choice (A-B switch).
    static const int mt2_env = [] { const char *e = getenv("MVIP_CONV_MT2"); return e ? atoi(e) : 1; }();
    if (mt2_env && tw == CV_TW && MT == 1 && Cout % 64 == 0) MT = 2;
    blocks = (tw == 8 ? (N + 3) / 4 : N * (W / tw) * (H / th)) * (Cout / (32 * MT));
}

extern "C" int64_t mvip_conv3x3_workspace_bytes(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W) {
    if (N <= 0 || !mvip_conv3x3_supported(Cout, Cin, H, W)) return 0;
    int tw, MT;
    int64_t blocks;
    cv_geometry(N, Cout, H, W, tw, MT, blocks);
    const int s = cv_splits(blocks, Cin / 16);
    return s > 1 ? (int64_t)s * N * Cout * H * W * 4 : 0;
}

static int conv3x3_launch(const void *xs, const void *packed, const float *bias, const float *chan_add,
                          const float *residual, const float *x_scale2, int64_t N, int64_t Cin, int64_t Cout,
                          int64_t H, int64_t W, float *y, void *workspace, int prec, void *stream,
                          double *row_moments = nullptr, float *tile_part = nullptr) {
    if (N < 0 || !mvip_conv3x3_supported(Cout, Cin, H, W) || prec < 0 || prec > 2) return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    if (!xs || !packed || !y) return MVIP_EINVAL;
    int tw, MT;
    int64_t blocks;
    cv_geometry(N, Cout, H, W, tw, MT, blocks);
    const int th = 256 / tw;
    ConvArgs a;
    a.xs = (const char *)xs; a.wp = (const char *)packed;
    const char *tail = (const char *)packed + Cout * Cin * 36;
    a.w_scale2 = (const float *)tail; a.zero16 = tail + 128;
    a.bias = bias; a.chan_add = chan_add; a.residual = residual; a.x_scale2 = x_scale2; a.y = y;
    a.N = (int)N; a.CK = (int)(Cin / 16); a.Cout = (int)Cout; a.H = (int)H; a.W = (int)W;
    a.tilesX = (int)(W / tw); a.tilesY = (int)(H / th); a.MB = (int)(Cout / (32 * MT));
    if (tw == 8) { a.tilesX = 1; a.tilesY = 1; }
    a.splits = 1; a.cks = a.CK; a.partial = nullptr; a.tile_part = nullptr;
    { static const int pr = [] { const char *e = getenv("MVIP_CONV_PRIO"); return e ? atoi(e) : 0; }(); a.prio = pr; }
#ifdef MVIP_EXPERIMENT_CONV
    { const char *e = getenv("MVIP_CONV_DBG"); a.dbg = e ? atoi(e) : 0; }
    { const char *e = getenv("MVIP_CONV_PROBE"); a.probe = e ? (unsigned long long *)strtoull(e, nullptr, 0) : nullptr; }
#endif
    hipStream_t st = as_stream(stream);
    // eight-wave workgroups on 16 x 32 pixel tiles (MVIP_CONV_WIDE=1; tuning switch, default off: measured equal to the
    // four-wave tile on every VAE / UNet shape and on the whole step, 7.56 vs 7.65 ms -- tools/conv_wide_ab.py)
    static const int wide_mode = [] { const char *e = getenv("MVIP_CONV_WIDE"); return e ? atoi(e) : 0; }();
    if (tw == CV_TW && wide_mode && H % 16 == 0 && prec != 1) {
        const int64_t tiles16 = N * (W / CV_TW) * (H / 16);
        int mtw = (Cout % 64 == 0 && tiles16 * (Cout / 64) >= 256) ? 2 : (tiles16 * (Cout / 32) >= 256 ? 1 : 0);
        // (MVIP_CONV_WIDE=2 -- 128 rows x 512 pixels per eight-wave workgroup, 12 fragment reads per 24 MFMAs instead of 16 --
        //  was measured no faster in round 3 and its <4, 32, 8, 3> instantiation spilled 164 registers: removed in round 5.)
        if (mtw) {
            a.tilesY = (int)(H / 16); a.MB = (int)(Cout / (32 * mtw));
            const int64_t wb = tiles16 * a.MB;
            if (wb > 0x7fffffffLL) return MVIP_EINVAL;
            if (mtw == 2 && prec == 2)
                hipLaunchKernelGGL((conv3x3_f16x3_kernel<2, CV_TW, 8, 2>), dim3((unsigned)wb), dim3(512), 0, st, a);
            else if (mtw == 2)
                hipLaunchKernelGGL((conv3x3_f16x3_kernel<2, CV_TW, 8>), dim3((unsigned)wb), dim3(512), 0, st, a);
            else if (prec == 2)
                hipLaunchKernelGGL((conv3x3_f16x3_kernel<1, CV_TW, 8, 2>), dim3((unsigned)wb), dim3(512), 0, st, a);
            else
                hipLaunchKernelGGL((conv3x3_f16x3_kernel<1, CV_TW, 8>), dim3((unsigned)wb), dim3(512), 0, st, a);
            return check_launch();
        }
    }
    if (workspace) {
        a.splits = cv_splits(blocks, a.CK);
        if (a.splits > 1) { a.cks = (a.CK + a.splits - 1) / a.splits; a.partial = (float *)workspace; }
    }
    blocks *= a.splits;
    if (blocks > 0x7fffffffLL) return MVIP_EINVAL;
    if (tile_part) {                                           // callers ask mvip_conv3x3_tile_moments_scratch_bytes first
        if (a.partial || tw == 8 || !row_moments) return MVIP_EINVAL;
        a.tile_part = tile_part;
    }
#define MVIP_CV_LAUNCH(F16_)   /* F16_ = NP: products per step */                                                                                             \
    do {                                                                                                                      \
        if (tw == 8) hipLaunchKernelGGL((conv3x3_f16x3_kernel<1, 8, 4, F16_>), dim3((unsigned)blocks), dim3(256), 0, st, a);         \
        else if (tw == 16) hipLaunchKernelGGL((conv3x3_f16x3_kernel<1, 16, 4, F16_>), dim3((unsigned)blocks), dim3(256), 0, st, a);  \
        else if (MT == 4) hipLaunchKernelGGL((conv3x3_f16x3_kernel<4, CV_TW, 4, F16_>), dim3((unsigned)blocks), dim3(256), 0, st, a); \
        else if (MT == 2) hipLaunchKernelGGL((conv3x3_f16x3_kernel<2, CV_TW, 4, F16_>), dim3((unsigned)blocks), dim3(256), 0, st, a); \
        else hipLaunchKernelGGL((conv3x3_f16x3_kernel<1, CV_TW, 4, F16_>), dim3((unsigned)blocks), dim3(256), 0, st, a);             \
    } while (0)
    if (prec == 1) MVIP_CV_LAUNCH(1); else if (prec == 2) MVIP_CV_LAUNCH(2); else MVIP_CV_LAUNCH(3);
#undef MVIP_CV_LAUNCH
    if (tile_part) {
        const int64_t HW = H * W, rows = N * Cout;
        const int chunks = (int)(mvip_groupnorm_workspace_bytes(1, 1, HW) / 16);
        hipLaunchKernelGGL(gn_moments_from_tiles_kernel, dim3((unsigned)rows), dim3(256), 0, st, (const float2 *)tile_part,
                           a.tilesX * a.tilesY * 4, rows, chunks, row_moments);
        return check_launch();
    }
    if (row_moments && !a.partial) return MVIP_EINVAL;         // callers ask mvip_conv3x3_row_moments_doubles first
    if (a.partial) {
        const int64_t total = N * Cout * H * W, HW = H * W;
        if (row_moments) {
#define MVIP_RM(LPR_, IT_) hipLaunchKernelGGL((cv_split_reduce_moments_kernel<LPR_, IT_>), dim3((unsigned)(N * Cout / (256 / LPR_))), \
                                              dim3(256), 0, st, a.partial, a.splits, total, (int)Cout, HW, a.w_scale2, x_scale2, \
                                              bias, chan_add, residual, y, row_moments)
            if (HW == 64) MVIP_RM(16, 1); else if (HW == 256) MVIP_RM(64, 1); else if (HW == 1024) MVIP_RM(256, 1); else MVIP_RM(256, 4);
#undef MVIP_RM
        } else {
            hipLaunchKernelGGL(cv_split_reduce_kernel, dim3((unsigned)((HW / 4 + 255) / 256), (unsigned)(N * Cout > 65535 ? 65535 : N * Cout)), dim3(256), 0, st,
                               a.partial, a.splits, total, (int)Cout, HW, a.w_scale2, x_scale2, bias, chan_add, residual, y);
        }
    }
    return check_launch();
}

// Doubles of `row_moments` ([N][Cout][2]) when this shape's launch is channel-split AND its rows fit the one-pair-per-row
// layout of mvip_groupnorm_stats (H * W = 64, 256, 1024 or 4096; N * Cout a multiple of 16), else 0.
extern "C" int64_t mvip_conv3x3_row_moments_doubles(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W) {
    if (mvip_conv3x3_workspace_bytes(N, Cin, Cout, H, W) == 0) return 0;
    const int64_t HW = H * W;
    if (!(HW == 64 || HW == 256 || HW == 1024 || HW == 4096) || (N * Cout) % 16 != 0) return 0;
    if (mvip_groupnorm_workspace_bytes(1, 1, HW) != 16) return 0;              // one moment pair per row in that layout
    return N * Cout * 2;
}

extern "C" int mvip_conv3x3_f16x3_ws_moments(const void *xs, const void *packed, const float *bias, const float *chan_add,
                                             const float *residual, const float *x_scale2, int64_t N, int64_t Cin,
                                             int64_t Cout, int64_t H, int64_t W, float *y, void *workspace, void *row_moments,
                                             int prec, void *stream) {
    if (!workspace || !row_moments || mvip_conv3x3_row_moments_doubles(N, Cin, Cout, H, W) == 0) return MVIP_EINVAL;
    return conv3x3_launch(xs, packed, bias, chan_add, residual, x_scale2, N, Cin, Cout, H, W, y, workspace, prec, stream,
                          (double *)row_moments);
}

// Bytes of `tile_scratch` for mvip_conv3x3_f16x3_tile_moments, or 0 when this shape's launch cannot leave moment partials from
// its epilogue: channel-split launches (they have mvip_conv3x3_f16x3_ws_moments), the 8 x 8 level, the eight-wave tiles.
extern "C" int64_t mvip_conv3x3_tile_moments_scratch_bytes(int64_t N, int64_t Cin, int64_t Cout, int64_t H, int64_t W) {
    if (N <= 0 || !mvip_conv3x3_supported(Cout, Cin, H, W)) return 0;
    if (mvip_conv3x3_workspace_bytes(N, Cin, Cout, H, W) != 0) return 0;
    static const int wide_mode = [] { const char *e = getenv("MVIP_CONV_WIDE"); return e ? atoi(e) : 0; }();
    if (wide_mode) return 0;
    int tw, MT;
    int64_t blocks;
    cv_geometry(N, Cout, H, W, tw, MT, blocks);
    if (tw == 8) return 0;
    const int th = 256 / tw;
    return N * Cout * (W / tw) * (H / th) * 4 * 2 * (int64_t)sizeof(float);
}

// The unsplit convolution that ALSO leaves the GroupNorm moments of y (the statistics pass of the GroupNorm that reads y next:
// norm2 after conv1, the next block's norm1 after conv2 -- the resnet chains of the VAE encoder and the UNet,
// DS_NeRF/guidance/sd_utils.py:330-352, :390-403): moments = [N][Cout][chunks][2] fp64 in the layout of mvip_groupnorm_stats'
// workspace (mvip_groupnorm_workspace_bytes(N, Cout, H * W) bytes), what mvip_groupnorm_split_planes_moments reads.  Two
// launches (the convolution; a reduction of N * Cout rows of tile partials) instead of convolution + a full read of y.
extern "C" int mvip_conv3x3_f16x3_tile_moments(const void *xs, const void *packed, const float *bias, const float *chan_add,
                                               const float *residual, const float *x_scale2, int64_t N, int64_t Cin,
                                               int64_t Cout, int64_t H, int64_t W, float *y, void *tile_scratch, void *moments,
                                               int prec, void *stream) {
    if (!tile_scratch || !moments || mvip_conv3x3_tile_moments_scratch_bytes(N, Cin, Cout, H, W) == 0) return MVIP_EINVAL;
    return conv3x3_launch(xs, packed, bias, chan_add, residual, x_scale2, N, Cin, Cout, H, W, y, nullptr, prec, stream,
                          (double *)moments, (float *)tile_scratch);
}

extern "C" int mvip_conv3x3_f16x3(const void *xs, const void *packed, const float *bias, const float *chan_add,
                                  const float *residual, const float *x_scale2, int64_t N, int64_t Cin, int64_t Cout,
                                  int64_t H, int64_t W, float *y, int prec, void *stream) {
    return conv3x3_launch(xs, packed, bias, chan_add, residual, x_scale2, N, Cin, Cout, H, W, y, nullptr, prec, stream);
}

// The same with a caller-owned workspace of mvip_conv3x3_workspace_bytes(...) bytes (may be null when that is 0): layers
// whose grid would leave most of the chip idle are split over the input channels and summed by a second launch.
extern "C" int mvip_conv3x3_f16x3_ws(const void *xs, const void *packed, const float *bias, const float *chan_add,
                                     const float *residual, const float *x_scale2, int64_t N, int64_t Cin, int64_t Cout,
                                     int64_t H, int64_t W, float *y, void *workspace, int prec, void *stream) {
    if (!workspace && mvip_conv3x3_workspace_bytes(N, Cin, Cout, H, W) > 0) return MVIP_EINVAL;
    return conv3x3_launch(xs, packed, bias, chan_add, residual, x_scale2, N, Cin, Cout, H, W, y, workspace, prec, stream);
}

/* ---- plain GEMM ---------------------------------------------------------------------------------------- */
extern "C" int64_t mvip_gemm_packed_bytes(int64_t M, int64_t K) {
    if (M <= 0 || K <= 0) return 0;
    return M * K * 4 + 256;
}

extern "C" int mvip_gemm_pack_a(const float *src, int64_t M, int64_t K, int64_t sm, int64_t sk, void *packed,
                                void *stream) {
    if (!src || !packed || M <= 0 || M % 32 != 0 || K <= 0 || K % 32 != 0) return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
    char *tail = (char *)packed + M * K * 4;
    zero_words(tail, 64, st);
    // maximum + scale in one launch: its two scratch words (tail + 16, + 20) are zero on entry and left zero
    hipLaunchKernelGGL(cv_absmax_scale_kernel, dim3(absmax_blocks(M * K)), dim3(256), 0, st, src, M * K, (unsigned *)(tail + 16),
                       (float *)tail);
    const int64_t total = (M / 32) * (K / 16) * 2 * 64;
    hipLaunchKernelGGL(gm_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, (int)M, (int)K, sm,
                       sk, (const float *)tail, (uint4 *)packed, (unsigned *)(tail + 12));
    return check_launch();
}

// splits for the 32/64-row kernel: only grids under 3/4 of the chip, at least 16 stages (512 k) per workgroup -- with
// shorter contractions the second launch costs more than the idle CUs did (measured, tools/gemm_bench.py: K = 640 at
// 160 workgroups 15.7 -> 21.2 us with splits, K = 5120 at 80 workgroups 97.6 -> 46.4 us)
static inline int gm_splits(int64_t blocks, int64_t nstage) {
    static const int forced = [] { const char *e = getenv("MVIP_GEMM_SPLITS"); return e ? atoi(e) : 0; }();   // tuning
    int64_t s = forced > 0 ? forced : (blocks >= 192 ? 1 : (384 + blocks - 1) / blocks);
    if (s > nstage / 16) s = nstage / 16;
    if (s > 32) s = 32;
    if (s < 1) s = 1;
    const int64_t sks = (nstage + s - 1) / s;
    return (int)((nstage + sks - 1) / sks);
}
// row tiles per workgroup of the 32/64-row kernels: cv_mt's choice, or (MVIP_GEMM_MT2=1, experiment) 64 rows whenever M allows
static inline int gm_mt(int64_t M, int64_t tiles) {
    static const int mt2_env = [] { const char *e = getenv("MVIP_GEMM_MT2"); return e ? atoi(e) : 0; }();
    const int mt = cv_mt(M, tiles);
    return (mt2_env && mt == 1 && M % 64 == 0) ? 2 : mt;
}
static inline int gm_auto_cfg(int64_t N, int64_t M, int64_t P) {
    // Measured on the UNet's linear layers (tools/gemm_bench.py, profiles/r2_gemm_tiles.json): with K = 320..1280
    // (10..40 stages) every tile shape lands within ~10 % of the 32/64-row kernel -- these launches are bound by
    // the per-workgroup prologue / epilogue and by grid quantisation, not by operand traffic -- and the square
    // tile only wins once M is large enough for several full waves of workgroups (the 1280-wide GEGLU projection).
    return (M % 128 == 0 && M >= 8192 && (M / 128) * N * (P / 256) >= 128) ? 2 : 1;
}

extern "C" int64_t mvip_gemm_workspace_bytes(int64_t N, int64_t K, int64_t M, int64_t P) {
    if (N <= 0 || M <= 0 || M % 32 != 0 || K <= 0 || K % 32 != 0 || P <= 0 || P % GM_PIX != 0 || P > GM_P_MAX) return 0;
    if (gm_auto_cfg(N, M, P) != 1) return 0;
    const int64_t tiles = P / GM_PIX;
    const int MT = gm_mt(M, N * tiles);
    const int s = gm_splits(N * tiles * (M / (32 * MT)), K / 32);
    return s > 1 ? (int64_t)s * N * M * P * 4 : 0;
}

static int gemm_launch(const void *xs, const void *packed, const float *bias, const float *chan_add,
                       const float *residual, const float *x_scale2, int64_t N, int64_t K, int64_t M, int64_t P,
                       float *y, int cfg, void *workspace, int prec, void *stream, double *ln_part = nullptr) {
    if (N < 0 || M <= 0 || M % 32 != 0 || K <= 0 || K % 32 != 0 || P <= 0 || P % GM_PIX != 0 || P > GM_P_MAX || prec < 0 || prec > 2)
        return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    if (!xs || !packed || !y) return MVIP_EINVAL;
    GemmArgs a;
    a.xs = (const char *)xs; a.wp = (const char *)packed;
    a.w_scale2 = (const float *)((const char *)packed + M * K * 4);
    a.bias = bias; a.chan_add = chan_add; a.residual = residual; a.x_scale2 = x_scale2; a.y = y;
    a.N = (int)N; a.CK = (int)(K / 16); a.M = (int)M; a.P = P;
    a.geglu_L = 0; a.absmax_bits = nullptr; a.nsec = 0; a.v_dt = 1; a.prec = prec;
    a.splits = 1; a.ln_part = nullptr; a.sks = (int)(K / 32); a.partial = nullptr;
#ifdef MVIP_EXPERIMENT_GEMM
    a.dbg = cfg >> 8; cfg &= 255;
#endif
    hipStream_t st = as_stream(stream);
    const bool auto_cfg = cfg == 0;
    if (cfg == 0) cfg = prec ? 1 : gm_auto_cfg(N, M, P);     // the one- / two-product instantiations exist for the 32/64-row kernel
    // prec 1 (fp16 mode): the producers wrote NO lo planes, so a three-product kernel would read uninitialised halves --
    // refuse every configuration without a single-product instantiation (prec 2 may fall back: the result is identical)
    if (prec == 1 && (cfg == 2 || cfg == 3 || cfg == 4)) return MVIP_EUNSUP;
    if (cfg < 0 || cfg > 5) return MVIP_EINVAL;
    if ((cfg == 2 || cfg == 3) && M % 128 != 0) return MVIP_EINVAL;
    if (cfg == 4 && M % 64 != 0) return MVIP_EINVAL;
    if (ln_part && cfg != 1 && cfg != 5) return MVIP_EUNSUP;   // the 32/64-row kernels' epilogue only
    if (cfg == 2) {
        a.tiles = (int)(P / 256); a.MB = (int)(M / 128);
        hipLaunchKernelGGL((gemm2_f16x3_kernel<2, 4, 3>), dim3((unsigned)(N * a.tiles * a.MB)), dim3(512), 0, st, a);
    } else if (cfg == 3) {
        a.tiles = (int)(P / 128); a.MB = (int)(M / 128);
        hipLaunchKernelGGL((gemm2_f16x3_kernel<2, 2, 2>), dim3((unsigned)(N * a.tiles * a.MB)), dim3(256), 0, st, a);
    } else if (cfg == 4) {
        a.tiles = (int)(P / 128); a.MB = (int)(M / 64);
        hipLaunchKernelGGL((gemm2_f16x3_kernel<1, 2, 3>), dim3((unsigned)(N * a.tiles * a.MB)), dim3(128), 0, st, a);
    } else {
        a.tiles = (int)(P / GM_PIX);
        const int MT = gm_mt(M, N * a.tiles);
        a.MB = (int)(M / (32 * MT));
        int64_t blocks = N * a.tiles * a.MB;
        if (workspace && auto_cfg) {
            a.splits = gm_splits(blocks, K / 32);
            if (a.splits > 1) { a.sks = (int)((K / 32 + a.splits - 1) / a.splits); a.partial = (float *)workspace; }
        }
        blocks *= a.splits;
        if (blocks > 0x7fffffffLL) return MVIP_EINVAL;
        // cfg 5 (and the automatic choice unless MVIP_GEMM_STREAM=0): the B-in-registers kernel
        static const bool stream_env = [] { const char *e = getenv("MVIP_GEMM_STREAM"); return e ? atoi(e) != 0 : true; }();
        const bool stream = cfg == 5 || (auto_cfg && stream_env);
        if (ln_part && (a.splits > 1 || !stream || MT > 2)) return MVIP_EUNSUP;       // callers ask mvip_gemm_ln_segments first
        a.ln_part = ln_part;
        if (prec == 1 && !(MT <= 2 && stream)) return MVIP_EUNSUP;             // (see above: no lo planes in fp16 mode)
        if (prec == 1) {
            if (MT == 2) hipLaunchKernelGGL((gemm5_f16x3_kernel<2, false, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((gemm5_f16x3_kernel<1, false, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        } else if (prec == 2 && MT <= 2 && stream) {
            if (MT == 2) hipLaunchKernelGGL((gemm5_f16x3_kernel<2, false, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((gemm5_f16x3_kernel<1, false, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        } else if (MT == 4)
            hipLaunchKernelGGL((gemm_f16x3_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else if (MT == 2 && stream)
            hipLaunchKernelGGL((gemm5_f16x3_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else if (MT == 2)
            hipLaunchKernelGGL((gemm_f16x3_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else if (stream)
            hipLaunchKernelGGL((gemm5_f16x3_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else
            hipLaunchKernelGGL((gemm_f16x3_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        if (a.partial) {
            const int64_t total = N * M * P;
            hipLaunchKernelGGL(cv_split_reduce_kernel, dim3((unsigned)((P / 4 + 255) / 256), (unsigned)(N * M > 65535 ? 65535 : N * M)), dim3(256), 0, st,
                               a.partial, a.splits, total, (int)M, P, a.w_scale2, x_scale2, bias, chan_add, residual, y);
        }
    }
    return check_launch();
}

// cfg: 0 = choose by shape, 1 = 32/64-row kernel (operands through LDS), 2 = 128 x 256 tile, 3 = 128 x 128, 4 = 64 x 128,
// 5 = 32/64-row kernel with the B operand streamed into registers (timing switch)
extern "C" int mvip_gemm_f16x3_cfg(const void *xs, const void *packed, const float *bias, const float *chan_add,
                                   const float *residual, const float *x_scale2, int64_t N, int64_t K, int64_t M, int64_t P,
                                   float *y, int cfg, int prec, void *stream) {
    return gemm_launch(xs, packed, bias, chan_add, residual, x_scale2, N, K, M, P, y, cfg, nullptr, prec, stream);
}

extern "C" int mvip_gemm_f16x3(const void *xs, const void *packed, const float *bias, const float *chan_add,
                               const float *residual, const float *x_scale2, int64_t N, int64_t K, int64_t M, int64_t P,
                               float *y, int prec, void *stream) {
    return gemm_launch(xs, packed, bias, chan_add, residual, x_scale2, N, K, M, P, y, 0, nullptr, prec, stream);
}

// mvip_gemm_f16x3 with a caller-owned workspace of mvip_gemm_workspace_bytes(N, K, M, P) bytes (null when that is 0):
// launches with fewer than one workgroup per CU and a long contraction (the 1280-channel transformer blocks at 16 x 16:
// 80 workgroups x 160 stages) are split over K and summed by a second launch, in index order.
extern "C" int mvip_gemm_f16x3_ws(const void *xs, const void *packed, const float *bias, const float *chan_add,
                                  const float *residual, const float *x_scale2, int64_t N, int64_t K, int64_t M, int64_t P,
                                  float *y, void *workspace, int prec, void *stream) {
    if (!workspace && mvip_gemm_workspace_bytes(N, K, M, P) > 0) return MVIP_EINVAL;
    return gemm_launch(xs, packed, bias, chan_add, residual, x_scale2, N, K, M, P, y, 0, workspace, prec, stream);
}

// Segments (= workgroup row blocks of 32 or 64 rows) of the LayerNorm partial statistics mvip_gemm_f16x3_ws_ln leaves for
// this shape, or 0 when the launch cannot leave them (split-K launches sum raw partial products in a second launch; the
// square-tile kernel has an epilogue of its own): the caller then runs mvip_layernorm_split_planes' own statistics pass.
extern "C" int64_t mvip_gemm_ln_segments(int64_t N, int64_t K, int64_t M, int64_t P, int prec) {
    if (N <= 0 || M <= 0 || M % 32 != 0 || K <= 0 || K % 32 != 0 || P <= 0 || P % GM_PIX != 0 || P > GM_P_MAX || prec < 0 || prec > 2) return 0;
    if ((prec ? 1 : gm_auto_cfg(N, M, P)) != 1) return 0;
    const int64_t tiles = P / GM_PIX;
    const int MT = gm_mt(M, N * tiles);
    if (gm_splits(N * tiles * (M / (32 * MT)), K / 32) > 1) return 0;
    static const bool stream_env = [] { const char *e = getenv("MVIP_GEMM_STREAM"); return e ? atoi(e) != 0 : true; }();
    if (MT > 2 || !stream_env) return 0;               // the statistics live in the B-in-registers kernel's epilogue
    return M / (32 * MT);
}

// mvip_gemm_f16x3_ws that ALSO leaves the LayerNorm statistics of its output over the M rows (DS_NeRF/guidance/sd_utils.py:390-403:
// the unet(...) call -- BasicTransformerBlock's norm1 / norm2 / norm3 read the residual stream a projection just wrote): per
// column p and segment g (mvip_gemm_ln_segments of them) the fp64 sum and sum of squares of the finished rows of the
// segment, ln_part[((n * S + g) * 2 + {0, 1}) * P + p] -- what mvip_layernorm_split_planes_stats consumes.  N * S * 2 * P doubles.
extern "C" int mvip_gemm_f16x3_ws_ln(const void *xs, const void *packed, const float *bias, const float *chan_add,
                                     const float *residual, const float *x_scale2, int64_t N, int64_t K, int64_t M, int64_t P,
                                     float *y, void *workspace, void *ln_part, int prec, void *stream) {
    if (!ln_part || mvip_gemm_ln_segments(N, K, M, P, prec) == 0) return MVIP_EINVAL;
    return gemm_launch(xs, packed, bias, chan_add, residual, x_scale2, N, K, M, P, y, 0, workspace, prec, stream, (double *)ln_part);
}

// First projection of the transformer feed-forward with the GEGLU fused into the epilogue:
//   out[n][r][p] = (W_v x + b_v)[r] * gelu((W_g x + b_g)[r])   for p < L, zero beyond,
// `packed` / `bias` hold the 2R rows interleaved in 32-row tiles (value rows 32 t .. 32 t + 31, then the gate rows of
// the same t); scale2 receives the power-of-two scale of |out|max; zero_word as in mvip_absmax_scale_sections.
extern "C" int mvip_gemm_geglu_f16x3(const void *xs, const void *packed, const float *bias, const float *x_scale2,
                                     int64_t N, int64_t K, int64_t M2, int64_t P, int64_t L, float *out, float *scale2,
                                     void *zero_word, int prec, void *stream) {
    if (prec < 0 || prec > 2 || N < 0 || M2 <= 0 || M2 % 64 != 0 || K <= 0 || K % 32 != 0 || P <= 0 || P % GM_PIX != 0 || P > GM_P_MAX || L <= 0 || L > P ||
        !scale2 || !zero_word)
        return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
    if (N > 0) {
        if (!xs || !packed || !out) return MVIP_EINVAL;
        GemmArgs a;
        a.xs = (const char *)xs; a.wp = (const char *)packed;
        a.w_scale2 = (const float *)((const char *)packed + M2 * K * 4);
        a.bias = bias; a.chan_add = nullptr; a.residual = nullptr; a.x_scale2 = x_scale2; a.y = out;
        a.N = (int)N; a.CK = (int)(K / 16); a.M = (int)M2; a.P = P; a.tiles = (int)(P / GM_PIX); a.MB = (int)(M2 / 64);
        a.geglu_L = (int)L; a.absmax_bits = (unsigned *)zero_word; a.nsec = 0; a.v_dt = 1; a.prec = prec;
        a.splits = 1; a.ln_part = nullptr; a.sks = (int)(K / 32); a.partial = nullptr;
        const int64_t blocks = N * a.tiles * a.MB;
        if (blocks > 0x7fffffffLL) return MVIP_EINVAL;
        static const bool stream_env = [] { const char *e = getenv("MVIP_GEMM_STREAM"); return e ? atoi(e) != 0 : true; }();
        if (prec == 1) hipLaunchKernelGGL((gemm5_f16x3_kernel<2, false, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else if (prec == 2) hipLaunchKernelGGL((gemm5_f16x3_kernel<2, false, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else if (stream_env) hipLaunchKernelGGL((gemm5_f16x3_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm_f16x3_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    }
    hipLaunchKernelGGL(gm_scale_from_bits_kernel, dim3(1), dim3(1), 0, st, scale2, (unsigned *)zero_word);
    return check_launch();
}

// ---- GEMMs whose epilogue writes the NEXT contraction's operands (csrc/plane_sink.h) ------------------------------
// The M rows are cut into `nsec` <= 3 consecutive sections of sec_rows[i] rows (multiples of 64).  Section i goes to
// sec_ptr[i] in the format sec_kind[i] -- 1: split planes [N][sec_rows/16][2][2][P][8 halves] (the B operand of a later
// mvip_gemm_f16x3 / the Q or K operand of the attention kernel), 2: attention V fragments
// [N][heads][v_dt][P/16][2][64][8 halves] with sec_rows = heads * v_dt * 32 (rows of head h at h * v_dt * 32 ..) -- as
// (W x + bias) * sec_scale[i], sec_scale a power of two the caller fixed BEFORE the launch from a bound of the result.
extern "C" int mvip_gemm_f16x3_sinks(const void *xs, const void *packed, const float *bias, const float *x_scale2, int64_t N,
                                     int64_t K, int64_t M, int64_t P, int nsec, const int64_t *sec_rows, const int *sec_kind,
                                     void *const *sec_ptr, const float *sec_scale, int v_dt, int prec, void *stream) {
    if (prec < 0 || prec > 2 || N < 0 || M <= 0 || M % 64 != 0 || K <= 0 || K % 32 != 0 || P <= 0 || P % GM_PIX != 0 || P > GM_P_MAX || nsec < 1 || nsec > 3 ||
        !sec_rows || !sec_kind || !sec_ptr || !sec_scale)
        return MVIP_EINVAL;
    int64_t end = 0, plane_rows = 0;
    int n_plane = 0;
    for (int i = 0; i < nsec; ++i) {
        if (sec_rows[i] <= 0 || sec_rows[i] % 64 != 0 || (sec_kind[i] != 1 && sec_kind[i] != 2) || !(sec_scale[i] > 0.f))
            return MVIP_EINVAL;
        // the plane sections come first; a V-fragment section (computed by the transposed instantiation, its own launch)
        // is the last one
        if (sec_kind[i] == 2 && (i != nsec - 1 || v_dt < 1 || sec_rows[i] % (32 * (int64_t)v_dt) != 0)) return MVIP_EINVAL;
        if (sec_kind[i] == 1) { plane_rows += sec_rows[i]; ++n_plane; }
        end += sec_rows[i];
    }
    if (end != M) return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    if (!xs || !packed) return MVIP_EINVAL;
    for (int i = 0; i < nsec; ++i) if (!sec_ptr[i]) return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
    GemmArgs a;
    a.xs = (const char *)xs;
    a.w_scale2 = (const float *)((const char *)packed + M * K * 4);
    a.chan_add = nullptr; a.residual = nullptr; a.x_scale2 = x_scale2; a.y = nullptr;
    a.N = (int)N; a.CK = (int)(K / 16); a.P = P;
    a.geglu_L = 0; a.absmax_bits = nullptr; a.v_dt = v_dt < 1 ? 1 : v_dt; a.prec = prec;
    a.splits = 1; a.ln_part = nullptr; a.sks = (int)(K / 32); a.partial = nullptr;
#ifdef MVIP_EXPERIMENT_GEMM
    a.dbg = 0;
#endif
    a.tiles = (int)(P / GM_PIX);
    auto launch = [&](int64_t row0, int64_t rows, bool swap) -> int {
        a.wp = (const char *)packed + row0 * K * 4;          // row tiles are contiguous in the packed image
        a.bias = bias ? bias + row0 : nullptr;
        a.M = (int)rows;
        const int MT = cv_mt(rows, N * a.tiles) >= 2 ? 2 : 1;
        a.MB = (int)(rows / (32 * MT));
        const int64_t blocks = N * a.tiles * a.MB;
        if (blocks > 0x7fffffffLL) return MVIP_EINVAL;
#define MVIP_G5(MT_, SW_)                                                                                                    \
        do {                                                                                                                    \
            if (prec == 1) hipLaunchKernelGGL((gemm5_f16x3_kernel<MT_, SW_, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);  \
            else if (prec == 2) hipLaunchKernelGGL((gemm5_f16x3_kernel<MT_, SW_, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a); \
            else hipLaunchKernelGGL((gemm5_f16x3_kernel<MT_, SW_, 3>), dim3((unsigned)blocks), dim3(256), 0, st, a);            \
        } while (0)
        if (swap) { if (MT == 2) MVIP_G5(2, true); else MVIP_G5(1, true); }
        else { if (MT == 2) MVIP_G5(2, false); else MVIP_G5(1, false); }
#undef MVIP_G5
        return MVIP_OK;
    };
    for (int i = 0; i < 3; ++i) { a.sec[i].ptr = nullptr; a.sec[i].scale = 1.f; a.sec[i].row_end = 0; a.sec[i].kind = 0; }
    if (n_plane > 0) {
        int64_t e = 0;
        for (int i = 0; i < n_plane; ++i) {
            e += sec_rows[i];
            a.sec[i].ptr = (char *)sec_ptr[i]; a.sec[i].scale = sec_scale[i]; a.sec[i].row_end = (int)e; a.sec[i].kind = 1;
        }
        a.nsec = n_plane;
        const int rc = launch(0, plane_rows, false);
        if (rc != MVIP_OK) return rc;
    }
    if (n_plane < nsec) {
        for (int i = 0; i < 3; ++i) { a.sec[i].ptr = nullptr; a.sec[i].scale = 1.f; a.sec[i].row_end = 0; a.sec[i].kind = 0; }
        a.sec[0].ptr = (char *)sec_ptr[nsec - 1]; a.sec[0].scale = sec_scale[nsec - 1]; a.sec[0].row_end = (int)sec_rows[nsec - 1];
        a.sec[0].kind = 2;
        a.nsec = 1;
        const int rc = launch(plane_rows, sec_rows[nsec - 1], true);
        if (rc != MVIP_OK) return rc;
    }
    return check_launch();
}

// mvip_gemm_geglu_f16x3 whose product leaves as the second projection's operand planes [N][(M2/2)/16][2][2][P][8 halves],
// times the power of two `out_scale` fixed before the launch (|value * gelu(gate)| <= |value| |gate|): no absolute-maximum
// collection, no scale launch, no fp32 intermediate.
extern "C" int mvip_gemm_geglu_f16x3_sink(const void *xs, const void *packed, const float *bias, const float *x_scale2,
                                          int64_t N, int64_t K, int64_t M2, int64_t P, int64_t L, void *out_planes,
                                          float out_scale, int prec, void *stream) {
    if (prec < 0 || prec > 2 || N < 0 || M2 <= 0 || M2 % 64 != 0 || K <= 0 || K % 32 != 0 || P <= 0 || P % GM_PIX != 0 || P > GM_P_MAX || L <= 0 || L > P ||
        !(out_scale > 0.f) || (M2 / 2) % 16 != 0)
        return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    if (!xs || !packed || !out_planes) return MVIP_EINVAL;
    GemmArgs a;
    a.xs = (const char *)xs; a.wp = (const char *)packed;
    a.w_scale2 = (const float *)((const char *)packed + M2 * K * 4);
    a.bias = bias; a.chan_add = nullptr; a.residual = nullptr; a.x_scale2 = x_scale2; a.y = nullptr;
    a.N = (int)N; a.CK = (int)(K / 16); a.M = (int)M2; a.P = P; a.tiles = (int)(P / GM_PIX); a.MB = (int)(M2 / 64);
    a.geglu_L = (int)L; a.absmax_bits = nullptr; a.nsec = 1; a.v_dt = 1; a.prec = prec;
    for (int i = 0; i < 3; ++i) { a.sec[i].ptr = nullptr; a.sec[i].scale = 1.f; a.sec[i].row_end = 0; a.sec[i].kind = 0; }
    a.sec[0].ptr = (char *)out_planes; a.sec[0].scale = out_scale; a.sec[0].row_end = (int)(M2 / 2); a.sec[0].kind = 1;
    a.splits = 1; a.ln_part = nullptr; a.sks = (int)(K / 32); a.partial = nullptr;
#ifdef MVIP_EXPERIMENT_GEMM
    a.dbg = 0;
#endif
    const int64_t blocks = N * a.tiles * a.MB;
    if (blocks > 0x7fffffffLL) return MVIP_EINVAL;
    if (prec == 1) hipLaunchKernelGGL((gemm5_f16x3_kernel<2, false, 1>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
    else if (prec == 2) hipLaunchKernelGGL((gemm5_f16x3_kernel<2, false, 2>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
    else hipLaunchKernelGGL((gemm5_f16x3_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), a);
    return check_launch();
}

// Y = (W X + bias + residual) * out_scale as ONE section of split planes [N][M/16][2][2][P][8 halves] -- the operand of
// the next GEMM (the second feed-forward projection handing the residual stream to proj_out) -- with the split-K of
// mvip_gemm_f16x3_ws: workspace of mvip_gemm_workspace_bytes(N, K, M, P) bytes (null when that is 0).
extern "C" int mvip_gemm_f16x3_planes_ws(const void *xs, const void *packed, const float *bias, const float *residual,
                                         const float *x_scale2, int64_t N, int64_t K, int64_t M, int64_t P, void *out_planes,
                                         float out_scale, void *workspace, int prec, void *stream) {
    if (prec < 0 || prec > 2 || N < 0 || M <= 0 || M % 32 != 0 || K <= 0 || K % 32 != 0 || P <= 0 || P % GM_PIX != 0 || P > GM_P_MAX || !(out_scale > 0.f))
        return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    if (!xs || !packed || !out_planes) return MVIP_EINVAL;
    if (!workspace && mvip_gemm_workspace_bytes(N, K, M, P) > 0) return MVIP_EINVAL;
    GemmArgs a;
    a.xs = (const char *)xs; a.wp = (const char *)packed;
    a.w_scale2 = (const float *)((const char *)packed + M * K * 4);
    a.bias = bias; a.chan_add = nullptr; a.residual = residual; a.x_scale2 = x_scale2; a.y = nullptr;
    a.N = (int)N; a.CK = (int)(K / 16); a.M = (int)M; a.P = P;
    a.geglu_L = 0; a.absmax_bits = nullptr; a.nsec = 1; a.v_dt = 1; a.prec = prec;
    for (int i = 0; i < 3; ++i) { a.sec[i].ptr = nullptr; a.sec[i].scale = 1.f; a.sec[i].row_end = 0; a.sec[i].kind = 0; }
    a.sec[0].ptr = (char *)out_planes; a.sec[0].scale = out_scale; a.sec[0].row_end = (int)M; a.sec[0].kind = 1;
    a.splits = 1; a.ln_part = nullptr; a.sks = (int)(K / 32); a.partial = nullptr;
#ifdef MVIP_EXPERIMENT_GEMM
    a.dbg = 0;
#endif
    a.tiles = (int)(P / GM_PIX);
    int MT = gm_mt(M, N * a.tiles);
    if (MT > 2) MT = 2;
    a.MB = (int)(M / (32 * MT));
    int64_t blocks = N * a.tiles * a.MB;
    if (workspace && gm_auto_cfg(N, M, P) == 1 && MT == gm_mt(M, N * a.tiles)) {
        a.splits = gm_splits(blocks, K / 32);
        if (a.splits > 1) { a.sks = (int)((K / 32 + a.splits - 1) / a.splits); a.partial = (float *)workspace; }
    }
    blocks *= a.splits;
    if (blocks > 0x7fffffffLL || N * (M / 8) > 65535) return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
    if (prec == 1) {
        if (MT == 2) hipLaunchKernelGGL((gemm5_f16x3_kernel<2, false, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm5_f16x3_kernel<1, false, 1>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    } else if (prec == 2) {
        if (MT == 2) hipLaunchKernelGGL((gemm5_f16x3_kernel<2, false, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm5_f16x3_kernel<1, false, 2>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    } else {
        if (MT == 2) hipLaunchKernelGGL((gemm5_f16x3_kernel<2, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((gemm5_f16x3_kernel<1, false>), dim3((unsigned)blocks), dim3(256), 0, st, a);
    }
    if (a.partial)
        hipLaunchKernelGGL(cv_split_reduce_planes_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)(N * (M / 8))), dim3(256), 0,
                           st, a.partial, a.splits, (int)N, (int)M, P, a.w_scale2, x_scale2, bias, residual, out_scale,
                           (uint4 *)out_planes, prec);
    return check_launch();
}
