// GroupNorm (+ fused SiLU) over NCHW for the score-distillation networks: the `norm -> silu -> conv`
// prologue of every ResNet block of the UNet and the VAE encoder the SDS step runs
// (DS_NeRF/guidance/sd_utils.py:207 vae.encode, :240 unet; block structure from the published SD-1.5
// architecture).
//
// Why hand-written: the stock moments kernel launches ONE workgroup per (sample, group) -- 32 or 64
// workgroups on a 256-CU part -- and the normalise / affine / SiLU steps are three more full passes.
// Here the reduction is split over (sample, channel, chunk) workgroups (>= 2048 for the 512^2 VAE
// levels) and the apply pass fuses affine + SiLU: 2 launches, x read twice, y written once.
// HBM-bound: 12 B per element forward (fp32), 20 B per element backward.
//
// Statistics are accumulated in fp64 (sum, sum of squares; var = E[x^2] - mean^2 evaluated in fp64),
// partials are combined in a fixed order (no atomics): results are bit-reproducible run to run.
#include "common.h"

namespace mvip {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_CHUNKS = 16;      // chunks per channel row
constexpr int GN_MIN_CHUNK = 4096;     // elements

struct GnPlan {
    int chunks;
    int64_t chunk_elems;
};

static inline GnPlan gn_plan(int64_t HW) {
    GnPlan p;
    int64_t ce = GN_MIN_CHUNK;
    if ((HW + ce - 1) / ce > GN_MAX_CHUNKS) {
        ce = (HW + GN_MAX_CHUNKS - 1) / GN_MAX_CHUNKS;
        ce = (ce + GN_MIN_CHUNK - 1) / GN_MIN_CHUNK * GN_MIN_CHUNK;
    }
    p.chunk_elems = ce;
    p.chunks = (int)((HW + ce - 1) / ce);
    if (p.chunks < 1) p.chunks = 1;
    return p;
}

// ---- 16-byte vector access for fp32 / fp16 rows ---------------------------------------------------
template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int N = 4; };
template <> struct VecOf<_Float16> { static constexpr int N = 8; };

template <typename T, int V>
__device__ __forceinline__ void load_vals(const T *__restrict__ p, float (&v)[V]) {
    if constexpr (V == 1) {
        v[0] = (float)p[0];
    } else {
        const uint4 raw = *reinterpret_cast<const uint4 *>(p);
        if constexpr (sizeof(T) == 4) {
            v[0] = __uint_as_float(raw.x); v[1] = __uint_as_float(raw.y);
            v[2] = __uint_as_float(raw.z); v[3] = __uint_as_float(raw.w);
        } else {
            typedef _Float16 h2 __attribute__((ext_vector_type(2)));
            const unsigned w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                h2 h = __builtin_bit_cast(h2, w[i]);
                v[2 * i] = (float)h.x; v[2 * i + 1] = (float)h.y;
            }
        }
    }
}

template <typename T, int V>
__device__ __forceinline__ void store_vals(T *__restrict__ p, const float (&v)[V]) {
    if constexpr (V == 1) {
        p[0] = (T)v[0];
    } else if constexpr (sizeof(T) == 4) {
        uint4 raw;
        raw.x = __float_as_uint(v[0]); raw.y = __float_as_uint(v[1]);
        raw.z = __float_as_uint(v[2]); raw.w = __float_as_uint(v[3]);
        *reinterpret_cast<uint4 *>(p) = raw;
    } else {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            h2 h; h.x = (_Float16)v[2 * i]; h.y = (_Float16)v[2 * i + 1];
            w[i] = __builtin_bit_cast(unsigned, h);
        }
        *reinterpret_cast<uint4 *>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}

// block-wide sum of two doubles; result valid in every thread
__device__ __forceinline__ void block_sum2(double &a, double &b) {
    __shared__ double red[2][GN_THREADS / 64];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
    const int w = threadIdx.x >> 6;
    __syncthreads();                      // protect `red` against a previous use
    if ((threadIdx.x & 63) == 0) { red[0][w] = a; red[1][w] = b; }
    __syncthreads();
    a = red[0][0] + red[0][1] + red[0][2] + red[0][3];
    b = red[1][0] + red[1][1] + red[1][2] + red[1][3];
}

__device__ __forceinline__ float silu_f(float z) { return z / (1.0f + expf(-z)); }
__device__ __forceinline__ float silu_grad_f(float z) {
    const float s = 1.0f / (1.0f + expf(-z));
    return s * (1.0f + z * (1.0f - s));
}

// ---- forward ------------------------------------------------------------------------------------
// grid (chunks, N*C).  part[(row*chunks + j)*2 + {0,1}] = {sum x, sum x^2} over the chunk, fp64.
template <typename T, int V>
__global__ void __launch_bounds__(GN_THREADS)
gn_moments_kernel(const T *__restrict__ x, int64_t HW, int64_t chunk_elems, int chunks, double *__restrict__ part) {
    const int64_t row = blockIdx.y;
    const int j = blockIdx.x;
    const int64_t lo = (int64_t)j * chunk_elems;
    const int64_t hi = (lo + chunk_elems < HW) ? lo + chunk_elems : HW;
    const T *xr = x + row * HW;
    double s = 0.0, q = 0.0;
    for (int64_t i = lo + (int64_t)threadIdx.x * V; i < hi; i += (int64_t)GN_THREADS * V) {
        float v[V];
        load_vals<T, V>(xr + i, v);
#pragma unroll
        for (int k = 0; k < V; ++k) { const double d = (double)v[k]; s += d; q += d * d; }
    }
    block_sum2(s, q);
    if (threadIdx.x == 0) { part[(row * chunks + j) * 2] = s; part[(row * chunks + j) * 2 + 1] = q; }
}

// Sum the P = cpg*chunks partial pairs of one (sample, group); every thread gets the totals.
__device__ __forceinline__ void group_totals(const double *__restrict__ part, int64_t pbase, int P, double &a, double &b) {
    a = 0.0; b = 0.0;
    for (int i = threadIdx.x; i < P; i += GN_THREADS) { a += part[(pbase + i) * 2]; b += part[(pbase + i) * 2 + 1]; }
    block_sum2(a, b);
}

// grid N*G, one wave: mean / rstd of each (sample, group) from the partials (used when the normalised
// tensor is produced by another kernel, e.g. the split-plane writer of conv3x3.hip)
__global__ void __launch_bounds__(64)
gn_finalize_kernel(const double *__restrict__ part, int C, int64_t HW, int cpg, int chunks, float eps,
                   float *__restrict__ mean, float *__restrict__ rstd) {
    const int G = C / cpg;
    const int64_t n = blockIdx.x / G;
    const int g = blockIdx.x % G;
    const int64_t pbase = (n * C + (int64_t)g * cpg) * chunks;
    const int P = cpg * chunks;
    double s = 0.0, q = 0.0;
    for (int i = threadIdx.x; i < P; i += 64) { s += part[(pbase + i) * 2]; q += part[(pbase + i) * 2 + 1]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (threadIdx.x == 0) {
        const double m = (double)cpg * (double)HW;
        const double mu = s / m;
        double var = q / m - mu * mu;
        if (var < 0.0) var = 0.0;
        mean[blockIdx.x] = (float)mu;
        rstd[blockIdx.x] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// grid (chunks, N*C): y = act((x - mean)*a + beta[c]), a = rstd*gamma[c]  (no a*x - a*mean cancellation)
template <typename T, int V, bool SILU>
__global__ void __launch_bounds__(GN_THREADS)
gn_apply_kernel(const T *__restrict__ x, const T *__restrict__ gamma, const T *__restrict__ beta,
                const double *__restrict__ part, int C, int64_t HW, int cpg, int chunks, int64_t chunk_elems,
                float eps, T *__restrict__ y, float *__restrict__ mean, float *__restrict__ rstd) {
    const int64_t row = blockIdx.y;
    const int j = blockIdx.x;
    const int c = (int)(row % C);
    const int64_t n = row / C;
    const int g = c / cpg;
    const int G = C / cpg;
    double s, q;
    group_totals(part, (n * C + (int64_t)g * cpg) * chunks, cpg * chunks, s, q);
    const double m = (double)cpg * (double)HW;
    const double mu = s / m;
    double var = q / m - mu * mu;
    if (var < 0.0) var = 0.0;
    const float rs = (float)(1.0 / sqrt(var + (double)eps));
    const float muf = (float)mu;
    if (j == 0 && c == g * cpg && threadIdx.x == 0) { mean[n * G + g] = muf; rstd[n * G + g] = rs; }
    const float a = rs * (gamma ? (float)gamma[c] : 1.0f);
    const float b = beta ? (float)beta[c] : 0.0f;
    const int64_t lo = (int64_t)j * chunk_elems;
    const int64_t hi = (lo + chunk_elems < HW) ? lo + chunk_elems : HW;
    const T *xr = x + row * HW;
    T *yr = y + row * HW;
    for (int64_t i = lo + (int64_t)threadIdx.x * V; i < hi; i += (int64_t)GN_THREADS * V) {
        float v[V];
        load_vals<T, V>(xr + i, v);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float z = (v[k] - muf) * a + b;
            v[k] = SILU ? silu_f(z) : z;
        }
        store_vals<T, V>(yr + i, v);
    }
}

// ---- backward (gradient w.r.t. x only; the SDS networks are frozen) ----------------------------------
// dz = dy * act'(z), dh = dz * gamma[c], xh = (x - mean) * rstd
// dx = rstd * (dh - mean_g(dh) - xh * mean_g(dh * xh))
template <typename T, int V, bool SILU>
__global__ void __launch_bounds__(GN_THREADS)
gn_bwd_partials_kernel(const T *__restrict__ x, const T *__restrict__ dy, const T *__restrict__ gamma,
                       const T *__restrict__ beta, const float *__restrict__ mean, const float *__restrict__ rstd,
                       int C, int64_t HW, int cpg, int chunks, int64_t chunk_elems, double *__restrict__ part) {
    const int64_t row = blockIdx.y;
    const int j = blockIdx.x;
    const int c = (int)(row % C);
    const int64_t n = row / C;
    const int G = C / cpg;
    const float mu = mean[n * G + c / cpg], rs = rstd[n * G + c / cpg];
    const float gm = gamma ? (float)gamma[c] : 1.0f;
    const float a = rs * gm;
    const float b = beta ? (float)beta[c] : 0.0f;
    const int64_t lo = (int64_t)j * chunk_elems;
    const int64_t hi = (lo + chunk_elems < HW) ? lo + chunk_elems : HW;
    const T *xr = x + row * HW, *dr = dy + row * HW;
    double s1 = 0.0, s2 = 0.0;
    for (int64_t i = lo + (int64_t)threadIdx.x * V; i < hi; i += (int64_t)GN_THREADS * V) {
        float v[V], d[V];
        load_vals<T, V>(xr + i, v);
        load_vals<T, V>(dr + i, d);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float dz = d[k];
            if (SILU) dz *= silu_grad_f((v[k] - mu) * a + b);
            const float dh = dz * gm;
            const float xh = (v[k] - mu) * rs;
            s1 += (double)dh; s2 += (double)dh * (double)xh;
        }
    }
    block_sum2(s1, s2);
    if (threadIdx.x == 0) { part[(row * chunks + j) * 2] = s1; part[(row * chunks + j) * 2 + 1] = s2; }
}

template <typename T, int V, bool SILU>
__global__ void __launch_bounds__(GN_THREADS)
gn_bwd_apply_kernel(const T *__restrict__ x, const T *__restrict__ dy, const T *__restrict__ gamma,
                    const T *__restrict__ beta, const float *__restrict__ mean, const float *__restrict__ rstd,
                    const double *__restrict__ part, int C, int64_t HW, int cpg, int chunks, int64_t chunk_elems,
                    T *__restrict__ dx, const T *__restrict__ dx_add, float *__restrict__ pmax) {
    const int64_t row = blockIdx.y;
    const int j = blockIdx.x;
    const int c = (int)(row % C);
    const int64_t n = row / C;
    const int g = c / cpg;
    const int G = C / cpg;
    double s1, s2;
    group_totals(part, (n * C + (int64_t)g * cpg) * chunks, cpg * chunks, s1, s2);
    const double m = (double)cpg * (double)HW;
    const float m1 = (float)(s1 / m), m2 = (float)(s2 / m);
    const float mu = mean[n * G + g], rs = rstd[n * G + g];
    const float gm = gamma ? (float)gamma[c] : 1.0f;
    const float a = rs * gm;
    const float b = beta ? (float)beta[c] : 0.0f;
    const int64_t lo = (int64_t)j * chunk_elems;
    const int64_t hi = (lo + chunk_elems < HW) ? lo + chunk_elems : HW;
    const T *xr = x + row * HW, *dr = dy + row * HW;
    T *or_ = dx + row * HW;
    const T *ar = dx_add ? dx_add + row * HW : nullptr;
    float mx = 0.f;
    for (int64_t i = lo + (int64_t)threadIdx.x * V; i < hi; i += (int64_t)GN_THREADS * V) {
        float v[V], d[V], e[V];
        load_vals<T, V>(xr + i, v);
        load_vals<T, V>(dr + i, d);
        if (ar) load_vals<T, V>(ar + i, e);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float dz = d[k];
            if (SILU) dz *= silu_grad_f((v[k] - mu) * a + b);
            const float dh = dz * gm;
            const float xh = (v[k] - mu) * rs;
            d[k] = rs * (dh - m1 - xh * m2);
            if (ar) d[k] += e[k];
            const float ab = fabsf(d[k]);
            mx = (ab == ab && ab < 3.0e38f) ? fmaxf(mx, ab) : mx;           // the filter of cv_absmax_scale_kernel
        }
        store_vals<T, V>(or_ + i, d);
    }
    if (pmax) {                  // this workgroup's maximum of |dx|: one plain store, no atomics, no fence (see the header note)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        __shared__ float wm[GN_THREADS / 64];
        if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = mx;
        __syncthreads();
        if (threadIdx.x == 0) pmax[row * chunks + j] = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    }
}

// scale2 = {s, 1/s} from per-workgroup maxima (one workgroup; the power-of-two rule of cv_scale_kernel in conv3x3.hip)
__global__ void __launch_bounds__(256) gn_pmax_scale_kernel(const float *__restrict__ pmax, int64_t count, float *__restrict__ scale2) {
    float m = 0.f;
    for (int64_t i = threadIdx.x; i < count; i += 256) m = fmaxf(m, pmax[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        float sc = 1.f;
        if (m > 0.f && m < 3.0e38f) {
            int e;
            frexpf(m, &e);
            int k = 10 - e;
            if (k > 60) k = 60;
            if (k < -60) k = -60;
            sc = ldexpf(1.f, k);
        }
        scale2[0] = sc;
        scale2[1] = 1.f / sc;
    }
}

template <typename T, int V>
static int gn_forward_t(const void *x, const void *gamma, const void *beta, int64_t N, int C, int64_t HW, int G,
                        float eps, int silu, void *y, float *mean, float *rstd, double *ws, hipStream_t st) {
    const GnPlan p = gn_plan(HW);
    const dim3 grid((unsigned)p.chunks, (unsigned)(N * C));
    const int cpg = C / G;
    hipLaunchKernelGGL((gn_moments_kernel<T, V>), grid, dim3(GN_THREADS), 0, st, (const T *)x, HW, p.chunk_elems,
                       p.chunks, ws);
    if (silu)
        hipLaunchKernelGGL((gn_apply_kernel<T, V, true>), grid, dim3(GN_THREADS), 0, st, (const T *)x,
                           (const T *)gamma, (const T *)beta, ws, C, HW, cpg, p.chunks, p.chunk_elems, eps, (T *)y,
                           mean, rstd);
    else
        hipLaunchKernelGGL((gn_apply_kernel<T, V, false>), grid, dim3(GN_THREADS), 0, st, (const T *)x,
                           (const T *)gamma, (const T *)beta, ws, C, HW, cpg, p.chunks, p.chunk_elems, eps, (T *)y,
                           mean, rstd);
    return check_launch();
}

template <typename T, int V>
static int gn_backward_t(const void *x, const void *dy, const void *gamma, const void *beta, const float *mean,
                         const float *rstd, int64_t N, int C, int64_t HW, int G, int silu, void *dx, double *ws,
                         const void *dx_add, float *pmax, hipStream_t st) {
    const GnPlan p = gn_plan(HW);
    const dim3 grid((unsigned)p.chunks, (unsigned)(N * C));
    const int cpg = C / G;
    if (silu) {
        hipLaunchKernelGGL((gn_bwd_partials_kernel<T, V, true>), grid, dim3(GN_THREADS), 0, st, (const T *)x,
                           (const T *)dy, (const T *)gamma, (const T *)beta, mean, rstd, C, HW, cpg, p.chunks,
                           p.chunk_elems, ws);
        hipLaunchKernelGGL((gn_bwd_apply_kernel<T, V, true>), grid, dim3(GN_THREADS), 0, st, (const T *)x,
                           (const T *)dy, (const T *)gamma, (const T *)beta, mean, rstd, ws, C, HW, cpg, p.chunks,
                           p.chunk_elems, (T *)dx, (const T *)dx_add, pmax);
    } else {
        hipLaunchKernelGGL((gn_bwd_partials_kernel<T, V, false>), grid, dim3(GN_THREADS), 0, st, (const T *)x,
                           (const T *)dy, (const T *)gamma, (const T *)beta, mean, rstd, C, HW, cpg, p.chunks,
                           p.chunk_elems, ws);
        hipLaunchKernelGGL((gn_bwd_apply_kernel<T, V, false>), grid, dim3(GN_THREADS), 0, st, (const T *)x,
                           (const T *)dy, (const T *)gamma, (const T *)beta, mean, rstd, ws, C, HW, cpg, p.chunks,
                           p.chunk_elems, (T *)dx, (const T *)dx_add, pmax);
    }
    return check_launch();
}

static inline bool gn_vec_ok(const void *a, const void *b, const void *c, int64_t HW, int vec) {
    const uintptr_t m = (uintptr_t)a | (uintptr_t)b | (uintptr_t)c;
    return (HW % vec) == 0 && (m & 15) == 0;
}

static inline int gn_check(int64_t N, int64_t C, int64_t HW, int G, int dtype) {
    if (N < 0 || C <= 0 || HW < 0 || G <= 0 || (C % G) != 0) return MVIP_EINVAL;
    if (dtype != 0 && dtype != 1) return MVIP_EINVAL;
    if (N * C > 0x7fffffffLL / 2) return MVIP_EINVAL;      // grid.y
    return MVIP_OK;
}

}  // namespace mvip

using namespace mvip;

extern "C" int64_t mvip_groupnorm_workspace_bytes(int64_t N, int64_t C, int64_t HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    return N * C * (int64_t)gn_plan(HW).chunks * 2 * (int64_t)sizeof(double);
}

extern "C" int mvip_groupnorm_stats(const void *x, int64_t N, int64_t C, int64_t HW, int G, float eps, int dtype,
                                    float *mean, float *rstd, void *workspace, void *stream) {
    int rc = gn_check(N, C, HW, G, dtype);
    if (rc != MVIP_OK) return rc;
    if (N == 0 || HW == 0) return MVIP_OK;
    if (!x || !workspace || (!mean) != (!rstd)) return MVIP_EINVAL;      // mean == rstd == NULL: moment partials only
    hipStream_t st = as_stream(stream);
    double *ws = (double *)workspace;
    const GnPlan p = gn_plan(HW);
    const dim3 grid((unsigned)p.chunks, (unsigned)(N * C));
    if (dtype == 0) {
        if (gn_vec_ok(x, nullptr, nullptr, HW, 4))
            hipLaunchKernelGGL((gn_moments_kernel<float, 4>), grid, dim3(GN_THREADS), 0, st, (const float *)x, HW,
                               p.chunk_elems, p.chunks, ws);
        else
            hipLaunchKernelGGL((gn_moments_kernel<float, 1>), grid, dim3(GN_THREADS), 0, st, (const float *)x, HW,
                               p.chunk_elems, p.chunks, ws);
    } else {
        if (gn_vec_ok(x, nullptr, nullptr, HW, 8))
            hipLaunchKernelGGL((gn_moments_kernel<_Float16, 8>), grid, dim3(GN_THREADS), 0, st, (const _Float16 *)x, HW,
                               p.chunk_elems, p.chunks, ws);
        else
            hipLaunchKernelGGL((gn_moments_kernel<_Float16, 1>), grid, dim3(GN_THREADS), 0, st, (const _Float16 *)x, HW,
                               p.chunk_elems, p.chunks, ws);
    }
    if (mean)
        hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)(N * G)), dim3(64), 0, st, ws, (int)C, HW, (int)(C / G),
                           p.chunks, eps, mean, rstd);
    return check_launch();
}

extern "C" int mvip_groupnorm_forward(const void *x, const void *gamma, const void *beta, int64_t N, int64_t C,
                                      int64_t HW, int G, float eps, int silu, int dtype, void *y, float *mean,
                                      float *rstd, void *workspace, void *stream) {
    int rc = gn_check(N, C, HW, G, dtype);
    if (rc != MVIP_OK) return rc;
    if (N == 0 || HW == 0) return MVIP_OK;
    if (!x || !y || !mean || !rstd || !workspace) return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
    double *ws = (double *)workspace;
    if (dtype == 0)
        return gn_vec_ok(x, y, nullptr, HW, 4)
                   ? gn_forward_t<float, 4>(x, gamma, beta, N, (int)C, HW, G, eps, silu, y, mean, rstd, ws, st)
                   : gn_forward_t<float, 1>(x, gamma, beta, N, (int)C, HW, G, eps, silu, y, mean, rstd, ws, st);
    return gn_vec_ok(x, y, nullptr, HW, 8)
               ? gn_forward_t<_Float16, 8>(x, gamma, beta, N, (int)C, HW, G, eps, silu, y, mean, rstd, ws, st)
               : gn_forward_t<_Float16, 1>(x, gamma, beta, N, (int)C, HW, G, eps, silu, y, mean, rstd, ws, st);
}

static int gn_backward_any(const void *x, const void *dy, const void *gamma, const void *beta, const float *mean,
                           const float *rstd, int64_t N, int64_t C, int64_t HW, int G, int silu, int dtype, void *dx,
                           void *workspace, const void *dx_add, float *pmax, void *stream) {
    int rc = gn_check(N, C, HW, G, dtype);
    if (rc != MVIP_OK) return rc;
    if (N == 0 || HW == 0) return MVIP_OK;
    if (!x || !dy || !dx || !mean || !rstd || !workspace) return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
    double *ws = (double *)workspace;
    const bool al = dx_add == nullptr || ((uintptr_t)dx_add & 15) == 0;
    if (dtype == 0)
        return gn_vec_ok(x, dy, dx, HW, 4) && al
                   ? gn_backward_t<float, 4>(x, dy, gamma, beta, mean, rstd, N, (int)C, HW, G, silu, dx, ws, dx_add, pmax, st)
                   : gn_backward_t<float, 1>(x, dy, gamma, beta, mean, rstd, N, (int)C, HW, G, silu, dx, ws, dx_add, pmax, st);
    return gn_vec_ok(x, dy, dx, HW, 8) && al
               ? gn_backward_t<_Float16, 8>(x, dy, gamma, beta, mean, rstd, N, (int)C, HW, G, silu, dx, ws, dx_add, pmax, st)
               : gn_backward_t<_Float16, 1>(x, dy, gamma, beta, mean, rstd, N, (int)C, HW, G, silu, dx, ws, dx_add, pmax, st);
}

extern "C" int mvip_groupnorm_backward(const void *x, const void *dy, const void *gamma, const void *beta,
                                       const float *mean, const float *rstd, int64_t N, int64_t C, int64_t HW, int G,
                                       int silu, int dtype, void *dx, void *workspace, void *stream) {
    return gn_backward_any(x, dy, gamma, beta, mean, rstd, N, C, HW, G, silu, dtype, dx, workspace, nullptr, nullptr, stream);
}

extern "C" int64_t mvip_groupnorm_backward_maxima(int64_t N, int64_t C, int64_t HW) {
    if (N <= 0 || C <= 0 || HW <= 0) return 0;
    return N * C * (int64_t)gn_plan(HW).chunks;
}

extern "C" int mvip_groupnorm_backward_fused(const void *x, const void *dy, const void *gamma, const void *beta,
                                             const float *mean, const float *rstd, int64_t N, int64_t C, int64_t HW, int G,
                                             int silu, int dtype, const void *dx_add, void *dx, float *maxima,
                                             void *workspace, void *stream) {
    return gn_backward_any(x, dy, gamma, beta, mean, rstd, N, C, HW, G, silu, dtype, dx, workspace, dx_add, maxima, stream);
}

extern "C" int mvip_absmax_scale_from_maxima(const float *maxima, int64_t count, float *scale2, void *stream) {
    if (count <= 0 || !maxima || !scale2) return MVIP_EINVAL;
    hipLaunchKernelGGL(gn_pmax_scale_kernel, dim3(1), dim3(256), 0, as_stream(stream), maxima, count, scale2);
    return check_launch();
}
