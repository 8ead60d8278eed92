// Weight gradients of the hash-grid model's small bias-free layers (DS_NeRF/run_nerf_helpers_tcnn.py:49-61,
// :72-84: tcnn FullyFusedMLPs 32->64->16 and 32->64->64->16):  dW[M][N] = sum_p dY[m][p] * X[n][p]  with
// M, N <= 64 and P = 10^5..10^7 points, both operands channel-major [C][P] as the rest of that model keeps them.
// A contraction this shape is a streaming read (HBM-bound: (M + N) x 4 B per point); the BLAS library runs it on
// 8-16 workgroups (0.36-0.74 ms per call at 1.3 M points, 37 % of the training iteration).  Here every wavefront
// of a full-chip grid owns a share of the points and accumulates its partial dW on the matrix pipe
// (v_mfma_f32_32x32x2_f32, K = 2 points per step), workgroups combine their four partials in LDS and write one
// [M][N] slab each; the caller sums the slabs (fixed order: bit-reproducible).
#include "common.h"

namespace mvip {

typedef float sg_f32x16 __attribute__((ext_vector_type(16)));
typedef float sg_f32x4 __attribute__((ext_vector_type(4)));

constexpr int SG_PC = 64;                 // points per staged chunk
constexpr int SG_LD = SG_PC + 4;          // LDS row pitch (floats): 16-lane groups of the b128 reads hit all 64 banks

// Rows are P floats apart (megabytes), so an operand fetch in MFMA order -- one row per lane -- touches 32 pages per
// instruction (first version of this kernel: 1.6-2.1 TB/s).  Here the workgroup stages a chunk of 64 points of all
// M + N rows through LDS with row-contiguous loads (16 lanes x 16 B = 256 B per row, 4 rows per wave instruction) and
// the waves read their operands transposed: wave w owns points [16w, 16w+16) of the chunk, lane (i, h) reads the 8
// consecutive points 16w + 8h .. +7 of row i (two ds_read_b128), K = 2 step s pairs point s of both halves.
template <int TM, int TN>
__global__ void __launch_bounds__(256, 2)
skinny_wgrad_kernel(const float *__restrict__ A, const float *__restrict__ B, int M, int N, int64_t P,
                    float *__restrict__ slabs) {
    constexpr int ROWS = 32 * (TM + TN);                       // staged rows (clamped copies beyond M / N)
    constexpr int PASSES = ROWS / 16;                          // 16 rows per pass of the 256 threads
    constexpr int STAGE = ROWS * SG_LD;
    constexpr int RED = TM * TN * 1024;
    __shared__ float lds[STAGE > RED ? STAGE : RED];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, i = lane & 31, h = lane >> 5;
    const int lr = threadIdx.x >> 4, lc = threadIdx.x & 15;    // loader: row within a pass, 16-byte column
    const int64_t chunks = P / SG_PC;                          // caller guarantees P % 64 == 0
    const int64_t per = (chunks + gridDim.x - 1) / gridDim.x;
    const int64_t c0 = (int64_t)blockIdx.x * per, c1 = (c0 + per < chunks) ? c0 + per : chunks;
    const float *src[PASSES];
#pragma unroll
    for (int q = 0; q < PASSES; ++q) {
        const int r = 16 * q + lr;                             // staged row: [0, 32 TM) from A, then B
        const bool fromA = r < 32 * TM;
        const int rr = fromA ? (r < M ? r : M - 1) : ((r - 32 * TM) < N ? (r - 32 * TM) : N - 1);
        src[q] = (fromA ? A : B) + (int64_t)rr * P + 4 * lc;
    }
    sg_f32x16 acc[TM][TN];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
    sg_f32x4 stage[PASSES];
    {
        const int64_t cf = c0 < chunks ? c0 : chunks - 1;      // (a workgroup without work loads a valid chunk it never uses)
#pragma unroll
        for (int q = 0; q < PASSES; ++q) stage[q] = *reinterpret_cast<const sg_f32x4 *>(src[q] + SG_PC * cf);
    }
    for (int64_t c = c0; c < c1; ++c) {
        __syncthreads();                                       // previous chunk consumed
#pragma unroll
        for (int q = 0; q < PASSES; ++q)
            *reinterpret_cast<sg_f32x4 *>(lds + (16 * q + lr) * SG_LD + 4 * lc) = stage[q];
        __syncthreads();
        {                                                      // next chunk's loads fly under this chunk's MFMAs
            const int64_t cn = c + 1 < c1 ? c + 1 : c;
#pragma unroll
            for (int q = 0; q < PASSES; ++q) stage[q] = *reinterpret_cast<const sg_f32x4 *>(src[q] + SG_PC * cn);
        }
        sg_f32x4 a[TM][2], b[TN][2];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int v = 0; v < 2; ++v)
                a[t][v] = *reinterpret_cast<const sg_f32x4 *>(lds + (32 * t + i) * SG_LD + 16 * wave + 8 * h + 4 * v);
#pragma unroll
        for (int u = 0; u < TN; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v)
                b[u][v] = *reinterpret_cast<const sg_f32x4 *>(lds + (32 * (TM + u) + i) * SG_LD + 16 * wave + 8 * h + 4 * v);
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int u = 0; u < TN; ++u) {
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][v].x, b[u][v].x, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][v].y, b[u][v].y, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][v].z, b[u][v].z, acc[t][u], 0, 0, 0);
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][v].w, b[u][v].w, acc[t][u], 0, 0, 0);
                }
    }
    __syncthreads();
    float *red = lds;
    for (int k = threadIdx.x; k < RED; k += 256) red[k] = 0.f;
    __syncthreads();
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
    for (int k = threadIdx.x; k < RED; k += 256) {
        const int l = k & 63, r = (k >> 6) & 15, tu = k >> 10, t = tu / TN, u = tu % TN;
        const int row = 32 * t + 8 * (r >> 2) + 4 * (l >> 5) + (r & 3), col = 32 * u + (l & 31);
        if (row < M && col < N) slab[row * N + col] = red[k];
    }
}

// ---- forward / data gradient of the same layers:  Y[M][P] = act(W[M][N] X[N][P])  --------------------------------
// (the tcnn FullyFusedMLP forward, DS_NeRF/run_nerf_helpers_tcnn.py:88-112; dX = W^T dY is the same kernel with the
// transposed weight).  A streaming product: (N + M) x 4 B per point against 2 M N flops, HBM-bound for M, N <= 64.
// A wavefront owns 128 consecutive points: lane (n, kh) loads X[2s + kh][p0 + 4n .. 4n + 3] as ONE float4 (512 B
// contiguous per half-wave), component j of that quad is its B operand (k-slot kh, column n) of the K = 2 step s of
// accumulator tile j, so tile j holds the columns p0 + 4n + j and a lane's registers r of the four tiles are again four
// consecutive points of one output row: float4 stores, 512 B contiguous per half-wave.  The weights (<= 16 KB) are the
// same for every wave: lane (m, kh) keeps W[32t + m][2s + kh] of all steps in registers (N/2 per row tile).
// Exact fp32 (v_mfma_f32_32x32x2_f32, k-ordered accumulation).  RELU fuses the activation that follows the hidden layers.
// Registers: NS weight values + 4 NS of X in flight + 64 TM accumulators.  Two waves per SIMD (256 registers each) hold that up to
// TM = 1, NS = 16; the 64-input layers (NS = 32: 224 registers + addresses) and the two-tile form get the whole file -- with
// the two-wave bound <1, 32> spilled 132 registers to scratch inside its point loop (round 4 build; tests/test_kernel_scratch.py).
template <int TM, int NS, bool RELU>          // TM row tiles of 32, NS = N / 2 steps (N padded to 2 NS with zero weights)
__global__ void __launch_bounds__(256, (TM == 2 || NS == 32) ? 1 : 2)
skinny_fwd_kernel(const float *__restrict__ Wt, int64_t w_sm, int64_t w_sn, const float *__restrict__ X, int M, int N,
                  int64_t P, float *__restrict__ Y) {
    const int lane = threadIdx.x & 63, n = lane & 31, kh = lane >> 5;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;          // global wave index
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float a[TM][NS];
#pragma unroll
    for (int t = 0; t < TM; ++t)
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int m = 32 * t + n, k = 2 * s + kh;
            a[t][s] = (m < M && k < N) ? Wt[m * w_sm + k * w_sn] : 0.f;
        }
    for (int64_t p0 = gw * 128; p0 < P; p0 += nw * 128) {
        const int64_t pc = p0 + 4 * n;                         // P % 4 == 0: a quad is inside or outside as a whole
        const bool in = pc < P;
        sg_f32x16 acc[TM][4];
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][j][r] = 0.f;
        sg_f32x4 x[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {                         // all loads of the tile in flight before the first MFMA
            const int k = 2 * s + kh;
            x[s] = (in && k < N) ? *reinterpret_cast<const sg_f32x4 *>(X + (int64_t)k * P + pc) : sg_f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int t = 0; t < TM; ++t) {
                acc[t][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], x[s].x, acc[t][0], 0, 0, 0);
                acc[t][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], x[s].y, acc[t][1], 0, 0, 0);
                acc[t][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], x[s].z, acc[t][2], 0, 0, 0);
                acc[t][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s], x[s].w, acc[t][3], 0, 0, 0);
            }
        if (in) {
#pragma unroll
            for (int t = 0; t < TM; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = 32 * t + 8 * (r >> 2) + 4 * kh + (r & 3);
                    if (row < M) {
                        sg_f32x4 v = {acc[t][0][r], acc[t][1][r], acc[t][2][r], acc[t][3][r]};
                        if (RELU) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
                        *reinterpret_cast<sg_f32x4 *>(Y + (int64_t)row * P + pc) = v;
                    }
                }
        }
    }
}

// The same for M <= 16 output rows (the density net's 64 -> 16 layer, the colour head, the data gradient towards 16-wide
// inputs) on v_mfma_f32_16x16x4_f32: the 32-row tile above would spend half of its matrix work on zero rows.  Lane (m, kq)
// keeps W[m][4s + kq]; a wave step covers 64 points (lane's quad 4 (lane & 15) .. +3 of rows 4s + kq, one MFMA per component).
template <int NS4, bool RELU>                  // NS4 = N / 4 steps (N padded to 4 NS4 with zero weights)
__global__ void __launch_bounds__(256, 2)
skinny_fwd16_kernel(const float *__restrict__ Wt, int64_t w_sm, int64_t w_sn, const float *__restrict__ X, int M, int N,
                    int64_t P, float *__restrict__ Y) {
    const int lane = threadIdx.x & 63, n = lane & 15, kq = lane >> 4;
    const int64_t gw = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int64_t nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
    float a[NS4];
#pragma unroll
    for (int s = 0; s < NS4; ++s) {
        const int k = 4 * s + kq;
        a[s] = (n < M && k < N) ? Wt[n * w_sm + k * w_sn] : 0.f;
    }
    for (int64_t p0 = gw * 64; p0 < P; p0 += nw * 64) {
        const int64_t pc = p0 + 4 * n;
        const bool in = pc < P;
        sg_f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = sg_f32x4{0.f, 0.f, 0.f, 0.f};
        sg_f32x4 x[NS4];
#pragma unroll
        for (int s = 0; s < NS4; ++s) {
            const int k = 4 * s + kq;
            x[s] = (in && k < N) ? *reinterpret_cast<const sg_f32x4 *>(X + (int64_t)k * P + pc) : sg_f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int s = 0; s < NS4; ++s) {
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], x[s].x, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], x[s].y, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], x[s].z, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], x[s].w, acc[3], 0, 0, 0);
        }
        if (in) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kq + r;                    // accumulator register r of lane (n, kq): row 4 kq + r, column n
                if (row < M) {
                    sg_f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                    if (RELU) { v.x = v.x > 0.f ? v.x : 0.f; v.y = v.y > 0.f ? v.y : 0.f; v.z = v.z > 0.f ? v.z : 0.f; v.w = v.w > 0.f ? v.w : 0.f; }
                    *reinterpret_cast<sg_f32x4 *>(Y + (int64_t)row * P + pc) = v;
                }
            }
        }
    }
}

}  // namespace mvip

using namespace mvip;

// Y [M][P] = act(W X) with W[m][n] = w[m*w_sm + n*w_sn] (so the data gradient passes the same weight with the strides
// swapped), X [N][P], 1 <= M, N <= 64, P % 4 == 0, 16-byte aligned X / Y; relu != 0 applies max(., 0).
extern "C" int mvip_skinny_linear(const float *w, int64_t w_sm, int64_t w_sn, const float *X, int64_t M, int64_t N, int64_t P,
                                  int relu, float *Y, void *stream) {
    if (M < 1 || M > 64 || N < 1 || N > 64 || P < 0 || P % 4 != 0) return MVIP_EINVAL;
    if (P == 0) return MVIP_OK;
    if (!w || !X || !Y || (((uintptr_t)X | (uintptr_t)Y) & 15)) return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
    int64_t waves = (P + 127) / 128;
    int64_t blocks = (waves + 3) / 4;
    if (blocks > 2048) blocks = 2048;
    const dim3 grid((unsigned)blocks), block(256);
    if (M <= 16) {                                              // 16-row tiles: 64 points per wave step
        int64_t b16 = ((P + 63) / 64 + 3) / 4;
        if (b16 > 2048) b16 = 2048;
        const dim3 grid16((unsigned)b16), block16(256);
#define SKL16(NS_) do { \
        if (relu) hipLaunchKernelGGL((skinny_fwd16_kernel<NS_, true>), grid16, block16, 0, st, w, w_sm, w_sn, X, (int)M, (int)N, P, Y); \
        else hipLaunchKernelGGL((skinny_fwd16_kernel<NS_, false>), grid16, block16, 0, st, w, w_sm, w_sn, X, (int)M, (int)N, P, Y); \
    } while (0)
        if (N > 32) SKL16(16);
        else if (N > 16) SKL16(8);
        else SKL16(4);
#undef SKL16
        return check_launch();
    }
    const int tm = M > 32 ? 2 : 1, ns = N > 32 ? 32 : (N > 16 ? 16 : 8);
#define SKL(TM_, NS_) do { \
        if (relu) hipLaunchKernelGGL((skinny_fwd_kernel<TM_, NS_, true>), grid, block, 0, st, w, w_sm, w_sn, X, (int)M, (int)N, P, Y); \
        else hipLaunchKernelGGL((skinny_fwd_kernel<TM_, NS_, false>), grid, block, 0, st, w, w_sm, w_sn, X, (int)M, (int)N, P, Y); \
    } while (0)
    if (tm == 2 && ns == 32) SKL(2, 32);
    else if (tm == 2 && ns == 16) SKL(2, 16);
    else if (tm == 2) SKL(2, 8);
    else if (ns == 32) SKL(1, 32);
    else if (ns == 16) SKL(1, 16);
    else SKL(1, 8);
#undef SKL
    return check_launch();
}

// number of [M][N] slabs mvip_skinny_wgrad writes for P points (the caller allocates slabs[count][M][N] and sums)
extern "C" int64_t mvip_skinny_wgrad_slabs(int64_t P) {
    if (P <= 0) return 0;
    int64_t blocks = (P / SG_PC + 7) / 8;                      // >= 8 chunks per workgroup
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    return blocks;
}

// dY [M][P], X [N][P] (row-major, P % 64 == 0, 1 <= M, N <= 64) -> slabs [mvip_skinny_wgrad_slabs(P)][M][N]
extern "C" int mvip_skinny_wgrad(const float *dY, const float *X, int64_t M, int64_t N, int64_t P, float *slabs,
                                 void *stream) {
    if (M < 1 || M > 64 || N < 1 || N > 64 || P <= 0 || P % SG_PC != 0) return MVIP_EINVAL;
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
