// Ray generation, ray-row assembly, stratified depths and the materialised sinusoidal encoding.
// All HBM-bound, one element per thread, arithmetic written in the reference's operation order
// (the library is built with -ffp-contract=off so a*b+c stays two roundings like torch's).
#include "rays_device.h"
#include <stdlib.h>

namespace mvip {

// dirs = ((x - W/2)/f, -(y - H/2)/f, -1); d[c] = sum_k dirs[k] * R[c][k]   (run_nerf_helpers.py:254-257)
__device__ __forceinline__ void pixel_ray(const float *__restrict__ c2w, int H, int W, float focal,
                                          int y, int x, float o[3], float d[3]) {
    const float dx = ((float)x - (float)W * .5f) / focal;
    const float dy = -(((float)y - (float)H * .5f) / focal);
    const float dz = -1.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        d[c] = (dx * c2w[c * 4 + 0] + dy * c2w[c * 4 + 1]) + dz * c2w[c * 4 + 2];
        o[c] = c2w[c * 4 + 3];
    }
}

__global__ void get_rays_kernel(const float *__restrict__ c2w, int H, int W, float focal, int y0,
                                int x0, int h, int w, float *__restrict__ ro, float *__restrict__ rd) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)h * w) return;
    const int y = y0 + (int)(i / w), x = x0 + (int)(i % w);
    float o[3], d[3];
    pixel_ray(c2w, H, W, focal, y, x, o, d);
#pragma unroll
    for (int c = 0; c < 3; ++c) { ro[i * 3 + c] = o[c]; rd[i * 3 + c] = d[c]; }
}

__device__ __forceinline__ void write_row(float *__restrict__ row, const float o[3], const float d[3],
                                          const float v[3], float near, float far) {
    // viewdirs = v / ||v||   (run.py:1188)
    const float n = sqrtf((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
    row[0] = o[0]; row[1] = o[1]; row[2] = o[2];
    row[3] = d[0]; row[4] = d[1]; row[5] = d[2];
    row[6] = near; row[7] = far;
    row[8] = v[0] / n; row[9] = v[1] / n; row[10] = v[2] / n;
}

__global__ void ray_rows_kernel(const float *__restrict__ ro, const float *__restrict__ rd,
                                const float *__restrict__ vsrc, float near, float far, int64_t B,
                                float *__restrict__ rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    float o[3], d[3], v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { o[c] = ro[i * 3 + c]; d[c] = rd[i * 3 + c]; v[c] = vsrc ? vsrc[i * 3 + c] : d[c]; }
    write_row(rows + i * 11, o, d, v, near, far);
}

__global__ void ray_rows_pose_kernel(const float *__restrict__ c2w, int H, int W, float focal,
                                     float near, float far, const int64_t *__restrict__ sel, int64_t B,
                                     float *__restrict__ rows) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    const int64_t p = sel ? sel[i] : i;
    float o[3], d[3];
    pixel_ray(c2w, H, W, focal, (int)(p / W), (int)(p % W), o, d);
    write_row(rows + i * 11, o, d, d, near, far);
}


__global__ void stratified_z_kernel(const float *__restrict__ rows, int ncols, int64_t B, int S,
                                    const float *__restrict__ t_vals, int lindisp,
                                    const float *__restrict__ t_rand, float *__restrict__ z) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * S) return;
    const int64_t r = i / S;
    const int s = (int)(i % S);
    const float near = rows[r * ncols + 6], far = rows[r * ncols + 7];
    z[i] = stratified_point(near, far, t_vals, s, S, lindisp, t_rand ? t_rand + i : nullptr);
}

// Four consecutive samples of one ray per thread (S % 4 == 0): the two per-ray reciprocals are taken once and
// the six depths a perturbed quad needs (s-1 .. s+4) cost six divides instead of 36 -- the one-sample kernel is
// bound by IEEE divides (0.2 of HBM).  Same expressions in the same order as z_at, so results are identical.
__global__ void stratified_z4_kernel(const float *__restrict__ rows, int ncols, int64_t B, int S,
                                     const float *__restrict__ t_vals, int lindisp,
                                     const float *__restrict__ t_rand, float *__restrict__ z) {
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // quad index
    const int Q = S / 4;
    if (q >= B * Q) return;
    const int64_t r = q / Q;
    const int s0 = (int)(q % Q) * 4;
    const float near = rows[r * ncols + 6], far = rows[r * ncols + 7];
    const float inear = 1.f / near, ifar = 1.f / far;
    auto at = [&](int s) {
        const float t = t_vals[s];
        if (lindisp) return 1.f / (inear * (1.f - t) + ifar * t);
        return near * (1.f - t) + far * t;
    };
    float zc[6];                                             // samples s0-1 .. s0+4 (clamped at the ends)
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int s = s0 - 1 + k;
        s = s < 0 ? 0 : (s > S - 1 ? S - 1 : s);
        zc[k] = (t_rand || (k >= 1 && k <= 4)) ? at(s) : 0.f;
    }
    float4 out;
    float *o = &out.x;
    float4 tr = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t_rand) tr = *reinterpret_cast<const float4 *>(t_rand + r * S + s0);
    const float *trp = &tr.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int s = s0 + k;
        const float c = zc[k + 1];
        if (!t_rand) { o[k] = c; continue; }
        const float upper = s < S - 1 ? .5f * (zc[k + 2] + c) : c;
        const float lower = s > 0 ? .5f * (c + zc[k]) : c;
        o[k] = lower + (upper - lower) * trp[k];
    }
    *reinterpret_cast<float4 *>(z + r * S + s0) = out;
}

// One wavefront per 16 rays, lane = sample (S <= 64 IT): the kernels above are bound by IEEE divides (two per sample even
// in the quad version: every depth is recomputed by its neighbours), not by memory.  Here each depth is computed ONCE --
// one divide per sample in lindisp mode -- and the stratum bounds come from the neighbouring lanes (DPP wave shifts); the
// two per-ray reciprocals are computed for the wave's rays at once, lane = ray, and broadcast through scalar registers.
// Four rays per trip with their loads issued together; a ray's row of S depths is one contiguous 4 S-byte store per
// register.  Same expressions in the same order as z_at: identical bits.
constexpr int ZW_RAYS = 16;
template <int IT>
__global__ void __launch_bounds__(256)
stratified_zw_kernel(const float *__restrict__ rows, int ncols, int64_t B, int S, const float *__restrict__ t_vals,
                     int lindisp, const float *__restrict__ t_rand, float *__restrict__ z) {
    const int lane = threadIdx.x & 63;
    const int64_t r0 = ((int64_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6))) * ZW_RAYS;     // scalar
    if (r0 >= B) return;
    const int nr = (int)((B - r0 < ZW_RAYS) ? B - r0 : ZW_RAYS);
    const int64_t rl = r0 + (lane < nr ? lane : nr - 1);
    const float near_l = rows[rl * ncols + 6], far_l = rows[rl * ncols + 7];
    const float inear_l = 1.f / near_l, ifar_l = 1.f / far_l;
    float t[IT];
#pragma unroll
    for (int i = 0; i < IT; ++i) t[i] = (i * 64 + lane < S) ? t_vals[i * 64 + lane] : 0.f;
    auto bc = [](float v, int k) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), k)); };
    for (int k0 = 0; k0 < nr; k0 += 4) {
        float tr[4][IT];
        if (t_rand) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int i = 0; i < IT; ++i)
                    tr[kk][i] = (k0 + kk < nr && i * 64 + lane < S) ? t_rand[(r0 + k0 + kk) * S + i * 64 + lane] : 0.f;
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const int k = k0 + kk;
            if (k >= nr) break;
            const float near = bc(near_l, k), far = bc(far_l, k), inear = bc(inear_l, k), ifar = bc(ifar_l, k);
            float zc[IT];
#pragma unroll
            for (int i = 0; i < IT; ++i)
                zc[i] = lindisp ? 1.f / (inear * (1.f - t[i]) + ifar * t[i]) : near * (1.f - t[i]) + far * t[i];
#pragma unroll
            for (int i = 0; i < IT; ++i) {
                const int sidx = i * 64 + lane;
                float o = zc[i];
                if (t_rand) {
                    const float prev_reg = i > 0 ? bc(zc[i > 0 ? i - 1 : 0], 63) : zc[i];        // sample 64 i - 1
                    const float next_reg = i + 1 < IT ? bc(zc[i + 1 < IT ? i + 1 : i], 0) : zc[i];
                    const float zl = dpp_from_prev(zc[i], prev_reg), zr = dpp_from_next(zc[i], next_reg);
                    const float upper = sidx < S - 1 ? .5f * (zr + zc[i]) : zc[i];
                    const float lower = sidx > 0 ? .5f * (zc[i] + zl) : zc[i];
                    o = lower + (upper - lower) * tr[kk][i];
                }
                if (sidx < S) z[(r0 + k) * S + sidx] = o;
            }
        }
    }
}

// run_nerf_helpers.py:27-52: channel c<3: x[c]; else m=c-3: octave m/6, fn (m%6)/3, dim m%3.
// One thread per (point, slot): slot 0 copies the three coordinates, slot 1 + 3 oct + dim evaluates sin AND cos of
// x[dim] * 2^oct (one argument, one range reduction) and writes channels 3 + 6 oct + dim and 6 + 6 oct + dim -- a wave's two
// store instructions together fill whole lines of the point's 4 (3 + 6 L)-byte row.  (One thread per output element with a
// 63-way division and a separate sinf / cosf each ran at 0.19 of HBM: profiles/r2_micro_hbm_kernels.jsonl.)
__global__ void posenc_kernel(const float *__restrict__ x, int64_t N, int L, float *__restrict__ y) {
    const int SL = 1 + 3 * L, C = 3 + 6 * L;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N * SL) return;
    const int64_t n = i / SL;
    const int slot = (int)(i - n * SL);
    float *yr = y + n * C;
    if (slot == 0) {
        yr[0] = x[n * 3]; yr[1] = x[n * 3 + 1]; yr[2] = x[n * 3 + 2];
        return;
    }
    const int m = slot - 1, oct = m / 3, dim = m - 3 * oct;
    const float a = x[n * 3 + dim] * (float)(1 << oct);
    float sn, cs;
    sincosf(a, &sn, &cs);                  // one range reduction for both (same values as sinf / cosf: same ocml kernels)
    yr[3 + 6 * oct + dim] = sn;
    yr[6 + 6 * oct + dim] = cs;
}

// The same encoding with the arithmetic the structure allows: the arguments of a coordinate are x, 2x, 4x, ... -- EXACT
// doublings -- so ONE range reduction per coordinate (fp64: r = x - n 2 pi) serves all octaves: the reduced angle is doubled
// and wrapped back into [-pi, pi] in fp64 (error 2^-53 per step, doubled L - 1 times: nothing at fp32), reduced to a quadrant,
// and a degree-7 / degree-8 polynomial pair (Cephes sinf / cosf coefficients, |r| <= pi/4) gives sin and cos -- ~35
// instruction-equivalents per octave instead of ~150 for an accurate sincosf of a large argument (posenc_kernel is bound by
// exactly that: 0.26 of HBM).  Agreement with fp64 sin / cos of the exact argument: <= 1.5e-7 absolute (sinf / cosf: <= 6e-8);
// |x| > 2^20 keeps the library functions.  thread = point; the workgroup's 256 x C tile is assembled in LDS (row stride C = 3 +
// 6 L is odd: conflict-free) and leaves as one contiguous run of 16-byte stores.
__device__ __forceinline__ void sincos_quadrant(double r, float &sn, float &cs) {
    const double q = rint(r * 0.63661977236758134308);                  // 2 / pi
    const float f = (float)fma(-q, 1.57079632679489661923, r);          // |f| <= pi / 4
    const float z = f * f;
    const float ps = f + f * z * (-1.6666654611e-1f + z * (8.3321608736e-3f + z * -1.9515295891e-4f));
    const float pc = 1.f - .5f * z + z * z * (4.166664568298827e-2f + z * (-1.388731625493765e-3f + z * 2.443315711809948e-5f));
    const int qi = (int)q & 3;
    const float a = (qi & 1) ? pc : ps, b = (qi & 1) ? ps : pc;
    sn = (qi & 2) ? -a : a;                                             // q = 1: sin = cos f, cos = -sin f; q = 2: both negated
    cs = ((qi + 1) & 2) ? -b : b;
}

template <int LMAX>
__global__ void __launch_bounds__(256) posenc_tile_kernel(const float *__restrict__ x, int64_t N, int L, float *__restrict__ y) {
    extern __shared__ float tile[];                                     // [256][C]
    const int C = 3 + 6 * L;
    const int64_t n0 = (int64_t)blockIdx.x * 256, n = n0 + threadIdx.x;
    if (n < N) {
        float *row = tile + threadIdx.x * C;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float xv = x[n * 3 + d];
            row[d] = xv;
            if (fabsf(xv) <= 1048576.f) {
                const double k = rint((double)xv * 0.15915494309189533577);                 // 1 / (2 pi)
                double r = fma(-k, 6.28318530717958647692, (double)xv);
                for (int o = 0; o < L; ++o) {
                    float sn, cs;
                    sincos_quadrant(r, sn, cs);
                    row[3 + 6 * o + d] = sn;
                    row[6 + 6 * o + d] = cs;
                    r += r;                                                                 // the next octave's angle, wrapped
                    r = r > 3.14159265358979323846 ? r - 6.28318530717958647692 : (r < -3.14159265358979323846 ? r + 6.28318530717958647692 : r);
                }
            } else {
                for (int o = 0; o < L; ++o) {
                    float sn, cs;
                    sincosf(xv * (float)(1 << o), &sn, &cs);
                    row[3 + 6 * o + d] = sn;
                    row[6 + 6 * o + d] = cs;
                }
            }
        }
    }
    __syncthreads();
    const int64_t cnt = ((N - n0 < 256) ? N - n0 : 256) * C;             // floats of this tile; its base n0 * C * 4 B is 16-byte aligned
    float *dst = y + n0 * C;
    const int64_t c4 = cnt >> 2;
    for (int64_t i = threadIdx.x; i < c4; i += 256)
        reinterpret_cast<float4 *>(dst)[i] = make_float4(tile[4 * i], tile[4 * i + 1], tile[4 * i + 2], tile[4 * i + 3]);
    for (int64_t i = (c4 << 2) + threadIdx.x; i < cnt; i += 256) dst[i] = tile[i];
}

static inline unsigned blocks_for(int64_t n, int t) { return (unsigned)((n + t - 1) / t); }

}  // namespace mvip

using namespace mvip;

extern "C" int mvip_get_rays(const float *c2w, int H, int W, float focal, int y0, int x0, int h, int w,
                             float *rays_o, float *rays_d, void *stream) {
    if (!c2w || !rays_o || !rays_d || H <= 0 || W <= 0 || h < 0 || w < 0 || y0 < 0 || x0 < 0 ||
        y0 + h > H || x0 + w > W || !(focal > 0.f)) return MVIP_EINVAL;
    const int64_t n = (int64_t)h * w;
    if (n == 0) return MVIP_OK;
    hipLaunchKernelGGL(get_rays_kernel, dim3(blocks_for(n, 256)), dim3(256), 0, as_stream(stream), c2w, H, W,
                       focal, y0, x0, h, w, rays_o, rays_d);
    return check_launch();
}

extern "C" int mvip_ray_rows(const float *rays_o, const float *rays_d, const float *viewdirs_src, float near,
                             float far, int64_t B, float *rows, void *stream) {
    if (B < 0 || (B > 0 && (!rays_o || !rays_d || !rows))) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    hipLaunchKernelGGL(ray_rows_kernel, dim3(blocks_for(B, 256)), dim3(256), 0, as_stream(stream), rays_o,
                       rays_d, viewdirs_src, near, far, B, rows);
    return check_launch();
}

extern "C" int mvip_ray_rows_from_pose(const float *c2w, int H, int W, float focal, float near, float far,
                                       const int64_t *sel, int64_t B, float *rows, void *stream) {
    if (!c2w || H <= 0 || W <= 0 || !(focal > 0.f) || B < 0 || (B > 0 && !rows)) return MVIP_EINVAL;
    if (!sel && B != (int64_t)H * W) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    hipLaunchKernelGGL(ray_rows_pose_kernel, dim3(blocks_for(B, 256)), dim3(256), 0, as_stream(stream), c2w, H,
                       W, focal, near, far, sel, B, rows);
    return check_launch();
}

extern "C" int mvip_stratified_z(const float *rows, int ncols, int64_t B, int S, const float *t_vals,
                                 int lindisp, const float *t_rand, float *z, void *stream) {
    if (B < 0 || S <= 0 || ncols < 8 || !t_vals || (B > 0 && (!rows || !z))) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    const bool aligned = ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(t_rand)) & 15) == 0;
    static const int wave_env = [] { const char *e = getenv("MVIP_STRATIFIED_WAVE"); return e ? atoi(e) : 1; }();   // A/B switch
    if (wave_env && S >= 32 && S <= 256 && B >= 4096) {
        const dim3 grid(blocks_for((B + ZW_RAYS - 1) / ZW_RAYS, 4));
#define MVIP_ZW(I) hipLaunchKernelGGL((stratified_zw_kernel<I>), grid, dim3(256), 0, as_stream(stream), rows, ncols, B, S, t_vals, \
                                      lindisp, t_rand, z)
        if (S <= 64) MVIP_ZW(1); else if (S <= 128) MVIP_ZW(2); else MVIP_ZW(4);
#undef MVIP_ZW
    } else if (S % 4 == 0 && aligned)
        hipLaunchKernelGGL(stratified_z4_kernel, dim3(blocks_for(B * (S / 4), 256)), dim3(256), 0, as_stream(stream),
                           rows, ncols, B, S, t_vals, lindisp, t_rand, z);
    else
        hipLaunchKernelGGL(stratified_z_kernel, dim3(blocks_for(B * S, 256)), dim3(256), 0, as_stream(stream), rows,
                           ncols, B, S, t_vals, lindisp, t_rand, z);
    return check_launch();
}

extern "C" int mvip_posenc(const float *x, int64_t N, int L, float *y, void *stream) {
    if (N < 0 || L < 0 || L > 16 || (N > 0 && (!x || !y))) return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    static const int tile_env = [] { const char *e = getenv("MVIP_POSENC_TILE"); return e ? atoi(e) : 1; }();       // A/B switch
    if (tile_env && L >= 1 && L <= 10 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)        // 256 x (3 + 6 L) floats of LDS <= 64 KB
        hipLaunchKernelGGL((posenc_tile_kernel<16>), dim3(blocks_for(N, 256)), dim3(256), 256 * (3 + 6 * L) * sizeof(float),
                           as_stream(stream), x, N, L, y);
    else
        hipLaunchKernelGGL(posenc_kernel, dim3(blocks_for(N * (1 + 3 * L), 256)), dim3(256), 0, as_stream(stream),
                           x, N, L, y);
    return check_launch();
}
