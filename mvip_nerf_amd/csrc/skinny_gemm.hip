// Weight gradients of the hash-grid model's small bias-free layers (DS_NeRF/run_nerf_helpers_tcnn.py:49-61,
// :72-84: tcnn FullyFusedMLPs 32->64->16 and 32->64->64->16):  dW[M][N] = sum_p dY[m][p] * X[n][p]  with
// M, N <= 64 and P = 10^5..10^7 points, both operands channel-major [C][P] as the rest of that model keeps them.
// A contraction this shape is a streaming read (HBM-bound: (M + N) x 4 B per point); the BLAS library runs it on
// 8-16 workgroups (0.36-0.74 ms per call at 1.3 M points, 37 % of the training iteration).  Here every wavefront
// of a full-chip grid owns a contiguous slice of the points, accumulates its partial dW on the matrix pipe
// (v_mfma_f32_32x32x2_f32, K = 2 points per step; lane (i, h) streams row i, 16 points = 64 bytes per iteration, so
// the two lanes of a row consume a whole 128-byte line), workgroups combine their four
// partials in LDS and write one [M][N] slab each; the caller sums the slabs (fixed order: bit-reproducible).
#include "common.h"

namespace mvip {

typedef float sg_f32x16 __attribute__((ext_vector_type(16)));

template <int TM, int TN>
__global__ void __launch_bounds__(256, 4)
skinny_wgrad_kernel(const float *__restrict__ A, const float *__restrict__ B, int M, int N, int64_t P,
                    float *__restrict__ slabs) {
    __shared__ float red[TM * TN * 1024];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    for (int k = threadIdx.x; k < TM * TN * 1024; k += 256) red[k] = 0.f;
    const int64_t groups = P / 32;                             // 32 points per iteration (caller guarantees P % 32 == 0)
    const int64_t total_waves = (int64_t)gridDim.x * 4, gw = (int64_t)blockIdx.x * 4 + wave;
    const int64_t per = (groups + total_waves - 1) / total_waves;
    const int64_t g0 = gw * per, g1 = (g0 + per < groups) ? g0 + per : groups;
    const float *pa[TM], *pb[TN];
#pragma unroll
    for (int t = 0; t < TM; ++t) { const int r = 32 * t + i; pa[t] = A + (int64_t)(r < M ? r : M - 1) * P + 16 * h; }
#pragma unroll
    for (int u = 0; u < TN; ++u) { const int r = 32 * u + i; pb[u] = B + (int64_t)(r < N ? r : N - 1) * P + 16 * h; }
    sg_f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    // lane (i, h) reads the 64 contiguous bytes [32 g + 16 h, +16) of its row: a wave consumes whole 128-byte lines
    // of 32 rows per iteration (with 16-byte pieces per iteration the lines were re-fetched from L2 up to four times:
    // 16 waves x 96 rows exceed the vector L1)
    for (int64_t g = g0; g < g1; ++g) {
        float4 a[TM][4], b[TN][4];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[t][q] = *reinterpret_cast<const float4 *>(pa[t] + 32 * g + 4 * q);
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) b[u][q] = *reinterpret_cast<const float4 *>(pb[u] + 32 * g + 4 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < TN; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][q].x, b[u][q].x, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][q].y, b[u][q].y, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][q].z, b[u][q].z, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][q].w, b[u][q].w, acc[t][u], 0, 0, 0);
                }
    }
    __syncthreads();                                           // `red` zeroed
    // the four waves add their tiles one after the other (plain read-modify-write between barriers: the order of
    // the fp32 additions is fixed)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < TN; ++u)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((t * TN + u) * 16 + r) * 64 + lane] += acc[t][u][r];
        }
        __syncthreads();
    }
    // register r = 4q + s of lane (j, h) holds row 8q + 4h + s, column j of its tile
    float *slab = slabs + (int64_t)blockIdx.x * M * N;
    for (int k = threadIdx.x; k < TM * TN * 1024; k += 256) {
        const int l = k & 63, r = (k >> 6) & 15, tu = k >> 10, t = tu / TN, u = tu % TN;
        const int row = 32 * t + 8 * (r >> 2) + 4 * (l >> 5) + (r & 3), col = 32 * u + (l & 31);
        if (row < M && col < N) slab[row * N + col] = red[k];
    }
}

}  // namespace mvip

using namespace mvip;

// number of [M][N] slabs mvip_skinny_wgrad writes for P points (the caller allocates slabs[count][M][N] and sums)
extern "C" int64_t mvip_skinny_wgrad_slabs(int64_t P) {
    if (P <= 0) return 0;
    int64_t blocks = (P / 32 + 31) / 32;                       // >= 8 iterations per wave
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    return blocks;
}

// dY [M][P], X [N][P] (row-major, P % 32 == 0, 1 <= M, N <= 64) -> slabs [mvip_skinny_wgrad_slabs(P)][M][N]
extern "C" int mvip_skinny_wgrad(const float *dY, const float *X, int64_t M, int64_t N, int64_t P, float *slabs,
                                 void *stream) {
    if (M < 1 || M > 64 || N < 1 || N > 64 || P <= 0 || P % 32 != 0) return MVIP_EINVAL;
    if (!dY || !X || !slabs) return MVIP_EINVAL;
    const dim3 grid((unsigned)mvip_skinny_wgrad_slabs(P)), block(256);
    hipStream_t st = as_stream(stream);
    const int tm = M > 32 ? 2 : 1, tn = N > 32 ? 2 : 1;
    if (tm == 2 && tn == 2) hipLaunchKernelGGL((skinny_wgrad_kernel<2, 2>), grid, block, 0, st, dY, X, (int)M, (int)N, P, slabs);
    else if (tm == 2) hipLaunchKernelGGL((skinny_wgrad_kernel<2, 1>), grid, block, 0, st, dY, X, (int)M, (int)N, P, slabs);
    else if (tn == 2) hipLaunchKernelGGL((skinny_wgrad_kernel<1, 2>), grid, block, 0, st, dY, X, (int)M, (int)N, P, slabs);
    else hipLaunchKernelGGL((skinny_wgrad_kernel<1, 1>), grid, block, 0, st, dY, X, (int)M, (int)N, P, slabs);
    return check_launch();
}
