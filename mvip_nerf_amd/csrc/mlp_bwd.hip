// Backward of the fused NeRF MLP: gradients of the 24 parameter tensors given d_raw.
// (Autograd through run_network / NeRF.forward, DS_NeRF/run.py:1108-1124,
//  DS_NeRF/run_nerf_helpers.py:104-127; the reference needs no gradient w.r.t. points.)
//
// Per tile of `tile_points` points, three kernels:
//   R  the stash-writing forward re-encodes the inputs and recomputes every activation, written once to a tile-blocked
//      stash ([row_tile][point_tile][32][32], 4 KB blocks).  Training keeps the stash of its own forward
//      (mlp_forward16_kernel<rays, STASH>, mlp_fwd16.hip; mvip_mlp_backward_stash) and skips R; the recompute path
//      runs the 32-point mlp_forward_kernel<STASH> of mlp_fwd.hip;
//   B1 delta propagation G_l = relu'(h_l) . W_{l+1}^T G_{l+1}: mlp_delta16_kernel (mlp_bwd16.hip: 16 points per wave, two
//      waves per SIMD; MVIP_DELTA16=0 selects mlp_delta_kernel below, the 32-point kernel with the same structure as the
//      32-point forward: a [256 x 32] gradient matrix in accumulator registers fed straight back as the B operand of the
//      next v_mfma_f32_32x32x2_f32 chain, TRANSPOSED weights through the LDS ring from a second packed image);
//   B2 mlp_wgrad_kernel           dW_l = G_l . Act_{l-1}^T: MFMA with the POINT index as the K dimension.  A workgroup
//      owns a slab of points and half of a 256x256 (or a 256x64 / 128x288) gradient: 4 waves x 8 accumulator tiles;
//      [32 x 32] operand blocks arrive by LDS-DMA in stages of 16 points (exact fp32: two workgroups per CU) or 32
//      points (split precision) with a source-side XOR swizzle so the per-lane ds_read_b128 of 4 consecutive points is
//      bank-conflict free; the stage loop is a PAIR of stages with a single exit (see wgrad_body); results leave as
//      fp32 atomics shaped as two 128-B row segments per wave instruction, directly into the natural [out][in] gradient
//      tensors.  Bias, sigma-row and rgb-row gradients ride along on the VALU from the operands already in registers.
// FLOPs: R + B1 + B2 ~= 2.9x the forward.  HBM: ~20 KB/point of stash traffic each way, i.e.
// ~90 FLOP/B -- still MFMA-bound at fp32 rates.
#include "common.h"
#include <stdlib.h>
#include "mlp_layout.h"
#include "mlp_device.h"
#include "mlp_device_f16.h"

namespace mvip {
using namespace mlp;

int mlp_forward_launch(const float *packed, const float *a, const float *b, int64_t p_begin, int64_t p_count,
                       int S, float *raw, float *stash, int64_t n_pt, bool from_rays, void *stream,
                       unsigned long long *clock_dbg = nullptr);
// split-precision (precision = 1) counterparts, mlp_fwd_f16x3.hip / mlp_bwd_f16x3.hip
int mlp_forward_f16x3_launch(const float *img, const float *a, const float *b, int64_t p_begin, int64_t p_count,
                             int S, float *raw, float *stash, int64_t n_pt, bool from_rays, void *stream);
int mlp_delta_f16x3_prepare(const float *image16, float *image_t16, void *stream);
int mlp_delta_f16x3_launch(const float *image_t16, const float *secb, const float *d_raw, int64_t p0, int64_t pc,
                           const float *act, int64_t act_n_pt, int64_t act_pt0, float *gst, int64_t n_pt,
                           void *stream);

// two-waves-per-SIMD delta kernel (precision 0), mlp_bwd16.hip
int mlp_delta16_prepare(const float *packed, float *image_t16, void *stream);
int mlp_delta16_launch(const float *image_t16, const float *secb, const float *d_raw, int64_t p0, int64_t pc,
                       const float *act, int64_t act_n_pt, int64_t act_pt0, float *gst, int64_t n_pt, void *stream);

// ------------------------------------------------------------------------------------------------
// transposed weight image for B1
// stream order: views^T (feature part), feature^T, then layers 7,6,5(h4 part),4,3,2,1 transposed
// ------------------------------------------------------------------------------------------------
constexpr int T_VIEWS_BLOCKS = 8 * 16;                 // out tiles: 256 feature units; K = 128 view units
constexpr int T_LAYER_BLOCKS = 8 * 32;
constexpr int T_TOTAL_BLOCKS = T_VIEWS_BLOCKS + 8 * T_LAYER_BLOCKS;      // 2176
constexpr int T_TOTAL_CHUNKS = T_TOTAL_BLOCKS / CHUNK_BLOCKS;            // 136
constexpr int T_FLOATS = T_TOTAL_BLOCKS * BLOCK_FLOATS;
static_assert(T_VIEWS_BLOCKS % (NSLOT * CHUNK_BLOCKS) == 0, "ring phase");

__device__ __forceinline__ float packed_w(const float *__restrict__ packed, int blk_off, int KG, int out, int k) {
    const int blk = blk_off + block_pos(out >> 5, k >> 3, KG);
    return packed[(int64_t)blk * BLOCK_FLOATS + ((k & 7) >> 2) * 128 + (out & 31) * 4 + (k & 3)];
}

__global__ void mlp_pack_transposed_kernel(const float *__restrict__ packed, float *__restrict__ pt) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T_FLOATS) return;
    const int blk = idx / BLOCK_FLOATS, r = idx % BLOCK_FLOATS;
    const int h = r / 128, i = (r % 128) / 4, s = r % 4;
    float v;
    if (blk < T_VIEWS_BLOCKS) {
        int ti, kg;
        block_tile(blk, 16, ti, kg);
        v = packed_w(packed, OFF_VIEWS, LV_KG, 8 * kg + 4 * h + s, 32 * ti + i);
    } else {
        const int m = (blk - T_VIEWS_BLOCKS) / T_LAYER_BLOCKS;      // 0: feature, 1..7: layers 7..1
        const int local = (blk - T_VIEWS_BLOCKS) % T_LAYER_BLOCKS;
        int ti, kg;
        block_tile(local, 32, ti, kg);
        const int out = 8 * kg + 4 * h + s, in = 32 * ti + i;
        if (m == 0) v = packed_w(packed, OFF_FEAT, LH_KG, out, in);
        else {
            const int l = 8 - m;
            if (l >= 6) v = packed_w(packed, OFF_L6 + (l - 6) * LH_BLOCKS, LH_KG, out, in);
            else if (l == 5) v = packed_w(packed, OFF_L5, L5_KG, out, 64 + in);
            else v = packed_w(packed, OFF_L1 + (l - 1) * LH_BLOCKS, LH_KG, out, in);
        }
    }
    pt[idx] = v;
}

// ------------------------------------------------------------------------------------------------
// B1: delta propagation
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 1) void mlp_delta_kernel(
    const float *__restrict__ packed_t, const float *__restrict__ secb, const float *__restrict__ d_raw,
    int64_t p_begin, int64_t p_count, const float *__restrict__ act, int64_t act_n_pt, int64_t act_pt0,
    float *__restrict__ gst, int64_t n_pt) {
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, hh = lane >> 5;
    const int64_t pt = (int64_t)blockIdx.x * 4 + wave;       // point tile inside this backward tile
    const int64_t pl = pt * 32 + j;
    const bool live = pl < p_count;

    Stream st{packed_t, lds, wave, lane};
    st.prologue(secb, T_TOTAL_CHUNKS);

    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) d = reinterpret_cast<const float4 *>(d_raw)[p_begin + pl];
    {   // d_raw^T rows 0..3 of tile GT_D (units 0..3 = registers 0..3 of the lower half-wave)
        f32x16 dt;
#pragma unroll
        for (int r = 0; r < 16; ++r) dt[r] = 0.f;
        if (hh == 0) { dt[0] = d.x; dt[1] = d.y; dt[2] = d.z; dt[3] = d.w; }
        store_tile(stash_block(gst, GT_D, n_pt, pt), dt, j, hh);
    }
    auto act_tile = [&](int row_tile) {
        return load_tile(stash_block(const_cast<float *>(act), row_tile, act_n_pt, act_pt0 + pt), j, hh);
    };
    auto put = [&](int row_tile, const f32x16 &t) { store_tile(stash_block(gst, row_tile, n_pt, pt), t, j, hh); };

    f32x16 vt[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) vt[t] = act_tile(AT_V + t);

    __syncthreads();
    const float *sb = lds + RING_FLOATS;
    APair32 a = st.first_pair();

    // grad wrt view-branch pre-activation: relu'(v) . (W_rgb^T d_rgb)
    f32x16 gv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int u0 = 32 * t + 8 * q + 4 * hh;
            const f32x4 w0 = *reinterpret_cast<const f32x4 *>(sb + SB_WRGB + u0);
            const f32x4 w1 = *reinterpret_cast<const f32x4 *>(sb + SB_WRGB + 128 + u0);
            const f32x4 w2 = *reinterpret_cast<const f32x4 *>(sb + SB_WRGB + 256 + u0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const float val = fmaf(w2[s], d.z, fmaf(w1[s], d.y, w0[s] * d.x));
                gv[t][4 * q + s] = vt[t][4 * q + s] > 0.f ? val : 0.f;
            }
        }
#pragma unroll
    for (int t = 0; t < 4; ++t) put(GT_V + t, gv[t]);

    // grad wrt feature = W_views[:, :256]^T gv   (feature_linear has no activation)
    f32x16 g[8], gn[8];
    run_layer<8, 16, false>(st, 0, a,
        [&](auto kg, auto s) { return gv[kg.value >> 2][4 * (kg.value & 3) + s.value]; }, NoPre{},
        [&](auto ti, const f32x16 &acc, int) { g[ti.value] = acc; put(GT_F + ti.value, acc); }, T_TOTAL_CHUNKS);

    // G_7 = relu'(h7) . (W_feat^T g_feat + w_alpha d_sigma)
    run_layer<8, 32, false>(st, T_VIEWS_BLOCKS / CHUNK_BLOCKS, a,
        [&](auto kg, auto s) { return g[kg.value >> 2][4 * (kg.value & 3) + s.value]; },
        [&](auto ti) { return act_tile(AT_H + 56 + ti.value); },
        [&](auto ti, const f32x16 &acc, const f32x16 &hv) {
            f32x16 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wa = *reinterpret_cast<const f32x4 *>(sb + SB_WALPHA + 32 * ti.value + 8 * q + 4 * hh);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    o[4 * q + s] = hv[4 * q + s] > 0.f ? fmaf(wa[s], d.w, acc[4 * q + s]) : 0.f;
            }
            gn[ti.value] = o;
            put(GT_G + 56 + ti.value, o);
        }, T_TOTAL_CHUNKS);
#pragma unroll
    for (int t = 0; t < 8; ++t) g[t] = gn[t];

    // G_m = relu'(h_m) . W_{m+1}^T G_{m+1},  m = 6..0   (layer 5 contributes its h4 columns only)
#pragma unroll 1
    for (int m = 6; m >= 0; --m) {
        run_layer<8, 32, false>(st, (T_VIEWS_BLOCKS + (7 - m) * T_LAYER_BLOCKS) / CHUNK_BLOCKS, a,
            [&](auto kg, auto s) { return g[kg.value >> 2][4 * (kg.value & 3) + s.value]; },
            [&](auto ti) { return act_tile(AT_H + 8 * m + ti.value); },
            [&](auto ti, const f32x16 &acc, const f32x16 &hv) {
                f32x16 o;
#pragma unroll
                for (int r = 0; r < 16; ++r) o[r] = hv[r] > 0.f ? acc[r] : 0.f;
                gn[ti.value] = o;
                put(GT_G + 8 * m + ti.value, o);
            }, T_TOTAL_CHUNKS);
#pragma unroll
        for (int t = 0; t < 8; ++t) g[t] = gn[t];
    }
}

// ------------------------------------------------------------------------------------------------
// B2: weight gradients
// ------------------------------------------------------------------------------------------------
struct Gemm {
    int g_tile0;            // first row tile of the G operand (gradient stash) this workgroup owns
    int n_off;              // output-row offset of that tile inside dW / db
    int a_tile0[2];         // activation stash row tiles: segment 0, segment 1
    int a_count0;           // k tiles in segment 0 (the rest of KT belong to segment 1)
    float *dW;              // natural [out][in] gradient tensor
    int ldw;                // its row length
    int col0[2];            // first column of each segment
    int valid[2];           // real columns in each segment (padding columns are skipped)
    float *db;              // bias gradient or null
    int extra;              // 0 none, 1 sigma row (alpha_linear), 2 rgb rows (rgb_linear)
    float *dWx, *dbx;       // gradients of the extra rows
};
constexpr int N_PRODUCTS = 20;
struct GemmTable { Gemm g[N_PRODUCTS]; };

constexpr int W_STAGE_BLOCKS = 14;                         // max blocks per stage (views-A: 4 G + 5 act + 4 V + d^T)
constexpr int W_STAGE_FLOATS = W_STAGE_BLOCKS * TILE_FLOATS;

// one 4 KB [32 units][32 points] block -> LDS, 16-B pieces XOR-swizzled by ((row>>1)&7) on the
// SOURCE side (LDS-DMA writes linearly)
__device__ __forceinline__ void stage_block_piece(const float *__restrict__ blk, float *lds_dst, int k, int lane) {
    const int r = 8 * k + (lane >> 3), c = lane & 7;
    glds16(blk + r * 32 + ((c ^ ((r >> 1) & 7)) << 2), lds_dst + k * 256);
}
__device__ __forceinline__ f32x4 read_piece(const float *tile, int row, int cw) {
    return *reinterpret_cast<const f32x4 *>(tile + row * 32 + ((cw ^ ((row >> 1) & 7)) << 2));
}

// half stages (16 points): the LDS image of a block is [32 units][16 points] = 2 KB, 16-B pieces XOR-swizzled by
// ((row>>2)&3) so that the 16 lanes of a ds_read_b128 phase (rows 16m..16m+15, one logical piece) hit 16 distinct
// 16-B bank groups
constexpr int HALF_FLOATS = 512;
template <int SP>
__device__ __forceinline__ f32x4 read_stage_piece(const float *tile, int row, int cw) {
    if constexpr (SP == 32) return read_piece(tile, row, cw);
    else return *reinterpret_cast<const f32x4 *>(tile + row * 16 + ((cw ^ ((row >> 2) & 3)) << 2));
}

// |d_raw| maximum of a backward call (bits of a non-negative float order like unsigned integers)
__global__ void absmax_kernel(const float *__restrict__ x, int64_t n, unsigned *__restrict__ out) {
    float m = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float v = fabsf(x[i]);
        m = (v == v && v < 3.0e38f) ? fmaxf(m, v) : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// PREC = 1: the point-contraction runs on v_mfma_f32_32x32x16_f16 with both operands split hi+lo in
// registers from the fp32 stash blocks (three products, fp32 accumulate).  Gradients are scaled by a
// per-call power of two (from max|d_raw|) before splitting and the result is un-scaled before the
// atomics, so the fp16 lo terms stay out of the subnormal range for any loss scale.
// BSPLIT (used by the f16 path of the 128x256 products): instead of each wave owning NTW gradient
// tiles x ALL act tiles, each wave owns ALL NTW(=4) gradient tiles x KT(=2) act tiles -- the same
// 8 accumulator tiles and 24 MFMAs per k-step, but 6 instead of 9 operand conversions.
// EXTRA (0 none, 1 sigma row, 2 rgb rows) is a template parameter: as a run-time field its branches sat inside the
// point-group loop and cost ~1 VALU move per MFMA in phi copies on every product (PMC, round 2: 2.7 VALU + 2.4 SALU
// per MFMA against 1.7 + 0.1 in the forward kernel); the per-block source addresses of a stage are hoisted into
// scalar base pointers (a stage advances every block by the same 4 KB), which removes the 64-bit multiply chain that
// issue_stage re-ran for each of the 12 blocks of every stage.
template <int NTW, int KT, int PREC = 0, bool BSPLIT = false, int EXTRA = 0, int SP = 32, int STAGE_BLOCKS = W_STAGE_BLOCKS>
__device__ __forceinline__ void wgrad_body(const Gemm &G, const float *__restrict__ act, int64_t act_n_pt,
                                           int64_t act_pt0, const float *__restrict__ gst, int64_t n_pt,
                                           int64_t pt0, int64_t pt1, float *lds, int wave, int lane,
                                           float gscale = 1.f, float inv_gscale = 1.f) {
    constexpr int NT = BSPLIT ? NTW : 4 * NTW;           // gradient tiles in a stage
    constexpr int KTT = BSPLIT ? 4 * KT : KT;            // act tiles in a stage
    const int a_first = BSPLIT ? 0 : wave * NTW;         // this wave's first gradient tile
    const int b_first = BSPLIT ? wave * KT : 0;          // this wave's first act tile
    const int i = lane & 31, hh = lane >> 5;
    constexpr int extra = EXTRA;
    constexpr int nextra = extra == 2 ? 5 : (extra == 1 ? 1 : 0);      // V tiles (4) + d^T tile
    constexpr int nblk = NT + KTT + nextra;
    const int g_tile0 = G.g_tile0, a0 = G.a_tile0[0], a1 = G.a_tile0[1], ac0 = G.a_count0;

    // SP = 32: wave w stages rows 8w..8w+7 of every block of the stage (one LDS-DMA per block and wave).
    // SP = 16: a 1-KB DMA piece is 16 rows x 16 points; wave w stages rows 16(w&1).. of the blocks b = (w>>1) mod 2.
    constexpr int BLK = SP == 32 ? TILE_FLOATS : HALF_FLOATS;       // floats of a block in LDS
    constexpr int NPG = SP / 8;                                     // 8-point groups per stage
    int lane_off;
    if constexpr (SP == 32) {
        const int r = 8 * wave + (lane >> 3), c = lane & 7;
        lane_off = r * 32 + ((c ^ ((r >> 1) & 7)) << 2);
    } else {
        const int r = 16 * (wave & 1) + (lane >> 2), c = lane & 3;
        lane_off = r * 32 + ((c ^ ((r >> 2) & 3)) << 2);
    }
    // wave-uniform source of block b of a stage: stage pt of block b = base + pt * 4 KB
    auto base_of = [&](int b) -> const float * {
        int tile;
        bool from_g = false;
        if (b < NT) { tile = g_tile0 + b; from_g = true; }
        else if (b < NT + KTT) { const int k = b - NT; tile = k < ac0 ? a0 + k : a1 + (k - ac0); }
        else if (extra == 2 && b < NT + KTT + 4) tile = AT_V + (b - NT - KTT);
        else { tile = GT_D; from_g = true; }
        return from_g ? gst + (int64_t)tile * n_pt * TILE_FLOATS
                      : act + ((int64_t)tile * act_n_pt + act_pt0) * TILE_FLOATS;
    };
    // SP = 32: every wave stages a piece of EVERY block.  SP = 16: waves 0, 1 stage the even blocks, waves 2, 3 the odd ones --
    // each wave keeps only the bases of ITS blocks, at compile-time positions (an array of all bases indexed by 2 k + (wave >> 1)
    // is a dynamically indexed private array: it went to scratch, 96 B per lane, and every stage re-loaded it from there).
    constexpr int NB = SP == 32 ? nblk : (nblk + 1) / 2;
    const int odd = SP == 32 ? 0 : (wave >> 1);
    const float *blk_base[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) blk_base[k] = base_of(SP == 32 ? k : 2 * k + odd);
    // `st` counts stages of SP points from the start of the tile
    auto issue_stage = [&](int64_t st, float *dst) {
        // scalar base + 32-bit per-lane byte offset (a backward tile spans < 2^32 bytes per stash row): the form the
        // global_load_lds instruction takes directly (saddr + voffset), no 64-bit vector address per block
        if constexpr (SP == 32) {
            const unsigned voff = (unsigned)(((unsigned)st * (unsigned)TILE_FLOATS + (unsigned)lane_off) * 4u);
#pragma unroll
            for (int b = 0; b < nblk; ++b)
                glds16(reinterpret_cast<const float *>(reinterpret_cast<const char *>(blk_base[b]) + voff),
                       dst + b * TILE_FLOATS + wave * 256);
        } else {
            const unsigned voff = (unsigned)((((unsigned)st >> 1) * (unsigned)TILE_FLOATS + ((unsigned)st & 1u) * 16u +
                                              (unsigned)lane_off) * 4u);
            float *d = dst + (wave & 1) * 256 + odd * HALF_FLOATS;
#pragma unroll
            for (int k = 0; k < NB; ++k)
                if (2 * k + odd < nblk)
                    glds16(reinterpret_cast<const float *>(reinterpret_cast<const char *>(blk_base[k]) + voff), d + 2 * k * HALF_FLOATS);
        }
    };

    f32x16 acc[NTW][KT];
#pragma unroll
    for (int a = 0; a < NTW; ++a)
#pragma unroll
        for (int b = 0; b < KT; ++b)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[a][b][q] = 0.f;
    float bsum[NTW];
#pragma unroll
    for (int a = 0; a < NTW; ++a) bsum[a] = 0.f;
    float xw[12], xb[3];                               // extra-row accumulators (sigma: xw[0..7], xb[0])
#pragma unroll
    for (int q = 0; q < 12; ++q) xw[q] = 0.f;
    xb[0] = xb[1] = xb[2] = 0.f;

    auto load_ops = [&](const float *stg, int pg, f32x4 (&A)[NTW], f32x4 (&Bv)[KT]) {
        const int cw = 2 * pg + hh;
#pragma unroll
        for (int a = 0; a < NTW; ++a) A[a] = read_stage_piece<SP>(stg + (wave * NTW + a) * BLK, i, cw);
#pragma unroll
        for (int b = 0; b < KT; ++b) Bv[b] = read_stage_piece<SP>(stg + (NT + b) * BLK, i, cw);
    };
    auto compute_stage = [&](const float *stg) {
        f32x4 A[NTW], Bv[KT], An[NTW], Bn[KT];
        load_ops(stg, 0, A, Bv);
#pragma unroll
        for (int pg = 0; pg < NPG; ++pg) {
            const int cw = 2 * pg + hh;
            if (pg < NPG - 1) load_ops(stg, pg + 1, An, Bn);      // operands of the next point group, one MFMA block early
#pragma unroll
            for (int a = 0; a < NTW; ++a) bsum[a] += (A[a][0] + A[a][1]) + (A[a][2] + A[a][3]);
#pragma unroll
            for (int a = 0; a < NTW; ++a)
#pragma unroll
                for (int b = 0; b < KT; ++b)
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc[a][b] = mfma(A[a][s], Bv[b][s], acc[a][b]);
            if (extra == 1 && wave == 0) {             // d(alpha_linear.weight)[k] = sum_p d_sigma[p] h7[k][p]
                const f32x4 ds = read_stage_piece<SP>(stg + (NT + KTT) * BLK, 3, cw);
                if constexpr (KT == 8) {
#pragma unroll
                    for (int b = 0; b < 8; ++b)
#pragma unroll
                        for (int s = 0; s < 4; ++s) xw[b] = fmaf(ds[s], Bv[b][s], xw[b]);
                }
                xb[0] += (ds[0] + ds[1]) + (ds[2] + ds[3]);
            }
            if (extra == 2 && wave == 0) {             // d(rgb_linear.weight)[c][k] = sum_p d_rgb[c][p] v[k][p]
                const float *xt = stg + (NT + KTT) * BLK;
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    const f32x4 dc = read_stage_piece<SP>(xt + 4 * BLK, cc, cw);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const f32x4 vv = read_stage_piece<SP>(xt + t * BLK, i, cw);
#pragma unroll
                        for (int s = 0; s < 4; ++s) xw[cc * 4 + t] = fmaf(dc[s], vv[s], xw[cc * 4 + t]);
                    }
                    xb[cc] += (dc[0] + dc[1]) + (dc[2] + dc[3]);
                }
            }
            if (pg < NPG - 1) {
#pragma unroll
                for (int a = 0; a < NTW; ++a) A[a] = An[a];
#pragma unroll
                for (int b = 0; b < KT; ++b) Bv[b] = Bn[b];
            }
        }
    };

    auto compute_stage_f16 = [&](const float *stg) {
#pragma unroll
        for (int ks = 0; ks < SP / 16; ++ks) {           // 16 points per k-step; lane (i,hh) takes points 16ks+8hh..+7
            const int cw = 4 * ks + 2 * hh;
            auto split8 = [&](const f32x4 &p0, const f32x4 &p1, float sc, h16x8 &hi, h16x8 &lo) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float v = (q < 4 ? p0[q & 3] : p1[q & 3]) * sc;
                    const _Float16 hv = (_Float16)v;
                    hi[q] = hv;
                    lo[q] = (_Float16)(v - (float)hv);
                }
            };
            h16x8 Ah[NTW], Al[NTW], Bh[KT], Bl[KT];
#pragma unroll
            for (int a = 0; a < NTW; ++a) {
                const float *t = stg + (a_first + a) * BLK;
                const f32x4 p0 = read_stage_piece<SP>(t, i, cw), p1 = read_stage_piece<SP>(t, i, cw + 1);
                bsum[a] += ((p0[0] + p0[1]) + (p0[2] + p0[3])) + ((p1[0] + p1[1]) + (p1[2] + p1[3]));
                split8(p0, p1, gscale, Ah[a], Al[a]);
            }
            f32x4 ds0, ds1;
            const bool do_alpha = extra == 1 && (BSPLIT || wave == 0);     // sigma-row dots over this wave's act tiles
            if (do_alpha) {
                ds0 = read_stage_piece<SP>(stg + (NT + KTT) * BLK, 3, cw);
                ds1 = read_stage_piece<SP>(stg + (NT + KTT) * BLK, 3, cw + 1);
                xb[0] += ((ds0[0] + ds0[1]) + (ds0[2] + ds0[3])) + ((ds1[0] + ds1[1]) + (ds1[2] + ds1[3]));
            }
#pragma unroll
            for (int b = 0; b < KT; ++b) {
                const float *t = stg + (NT + b_first + b) * BLK;
                const f32x4 p0 = read_stage_piece<SP>(t, i, cw), p1 = read_stage_piece<SP>(t, i, cw + 1);
                split8(p0, p1, 1.f, Bh[b], Bl[b]);
                if (do_alpha) {
                    if constexpr (KT <= 8) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) xw[b] = fmaf(ds1[q], p1[q], fmaf(ds0[q], p0[q], xw[b]));
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < NTW; ++a)
#pragma unroll
                for (int b = 0; b < KT; ++b) {
                    acc[a][b] = mfma16(Ah[a], Bh[b], acc[a][b]);
                    acc[a][b] = mfma16(Ah[a], Bl[b], acc[a][b]);
                    acc[a][b] = mfma16(Al[a], Bh[b], acc[a][b]);
                }
            if (extra == 2 && wave == 0) {             // rgb rows: fp32 VALU dots on the same 8 points
                const float *xt = stg + (NT + KTT) * BLK;
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    const f32x4 d0 = read_stage_piece<SP>(xt + 4 * BLK, cc, cw), d1 = read_stage_piece<SP>(xt + 4 * BLK, cc, cw + 1);
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const f32x4 v0 = read_stage_piece<SP>(xt + t * BLK, i, cw), v1 = read_stage_piece<SP>(xt + t * BLK, i, cw + 1);
#pragma unroll
                        for (int q = 0; q < 4; ++q) xw[cc * 4 + t] = fmaf(d1[q], v1[q], fmaf(d0[q], v0[q], xw[cc * 4 + t]));
                    }
                    xb[cc] += ((d0[0] + d0[1]) + (d0[2] + d0[3])) + ((d1[0] + d1[1]) + (d1[2] + d1[3]));
                }
            }
        }
    };
    auto compute = [&](const float *stg) {
        if constexpr (PREC == 1) compute_stage_f16(stg);
        else compute_stage(stg);
    };

    // double-buffered stages with COMPILE-TIME buffer addresses (so LDS-DMA writes into one buffer
    // provably do not alias the ds_reads of the other and no wait is inserted between them)
    static_assert(nblk <= STAGE_BLOCKS, "stage buffer");
    float *buf0 = lds, *buf1 = lds + STAGE_BLOCKS * BLK;
    // The loop body is a whole PAIR of stages with its only exit at the bottom, and an odd last stage runs after it:
    // with an exit between the two stages the accumulators reached the flush from two places and the register
    // allocator kept two copies of all eight tiles, moving 128 registers from one to the other every stage.
    const int64_t st0 = pt0 * (32 / SP), st1 = pt1 * (32 / SP);
    issue_stage(st0, buf0);
    __syncthreads();
    const int64_t n_pairs = (st1 - st0) >> 1;
    int64_t st = st0;
    for (int64_t pr = 0; pr < n_pairs; ++pr, st += 2) {
        issue_stage(st + 1, buf1);
        compute(buf0);
        __syncthreads();          // stage st+1 landed (vmcnt(0)) and everyone is done with buf0
        if (st + 2 < st1) issue_stage(st + 2, buf0);
        compute(buf1);
        __syncthreads();
    }
    if (st < st1) compute(buf0);

    // ---- flush: fp32 atomics, 32 consecutive columns per half-wave (two 128-B row segments) ----
#pragma unroll
    for (int a = 0; a < NTW; ++a) {
        const int n0 = G.n_off + 32 * (a_first + a);
#pragma unroll
        for (int b = 0; b < KT; ++b) {
            const int bg = b_first + b;                       // act tile index inside the product
            const int seg = bg < ac0 ? 0 : 1;
            const int kcol = 32 * (seg ? bg - ac0 : bg) + i;
            if (kcol < G.valid[seg]) {
                float *dst = G.dW + (int64_t)(n0 + 4 * hh) * G.ldw + G.col0[seg] + kcol;
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    atomicAdd(dst + (int64_t)(8 * (q >> 2) + (q & 3)) * G.ldw, acc[a][b][q] * inv_gscale);
            }
        }
        if (G.db && (!BSPLIT || (a & 3) == wave)) {           // BSPLIT: every wave holds all NTW sums, wave w flushes tiles w, w + 4
            const float tot = bsum[a] + __shfl_xor(bsum[a], 32, 64);
            if (hh == 0) atomicAdd(G.db + n0 + i, tot);
        }
    }
    if (extra == 1 && (BSPLIT || wave == 0)) {
#pragma unroll
        for (int b = 0; b < (KT < 8 ? KT : 8); ++b) {
            const float tot = xw[b] + __shfl_xor(xw[b], 32, 64);
            if (hh == 0) atomicAdd(G.dWx + 32 * (b_first + b) + i, tot);
        }
        const float tb = xb[0] + __shfl_xor(xb[0], 32, 64);
        if (lane == 0 && wave == 0) atomicAdd(G.dbx, tb);
    }
    if (extra == 2 && wave == 0) {
#pragma unroll
        for (int cc = 0; cc < 3; ++cc) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float tot = xw[cc * 4 + t] + __shfl_xor(xw[cc * 4 + t], 32, 64);
                if (hh == 0) atomicAdd(G.dWx + cc * 128 + 32 * t + i, tot);
            }
            const float tb = xb[cc] + __shfl_xor(xb[cc], 32, 64);
            if (lane == 0) atomicAdd(G.dbx + cc, tb);
        }
    }
}

// The product table lives in device memory so that `tab[blockIdx.y]` is a handful of scalar loads
// (a by-value kernel argument indexed at run time is demoted to scratch, and its reloads --
// vector memory operations -- serialise every LDS-DMA behind an s_waitcnt vmcnt(0)).
__global__ void mlp_wgrad_table_kernel(GemmTable tab, Gemm *__restrict__ out) {
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < N_PRODUCTS; ++k) out[k] = tab.g[k];
    }
}

// blockIdx.y: 0..15 the eight 256x256 products as two 128-row halves each, 16..17 the two 256x64
// products with the encoding, 18..19 the view branch in two column groups.  No wave holds more than
// 128 accumulator registers.
// SP = 16 (half stages, 56 KB of LDS, <= 128 + 128 registers): TWO workgroups per CU, so the barrier / LDS-latency bubble
// at every stage boundary of one workgroup runs under the other's MFMAs.
// WHOLE (the split-precision kernel with 32-point stages, one workgroup per CU): the eight 256 x 256 products are NOT cut in two
// row halves -- one workgroup stages 8 gradient + 8 activation blocks (64 KB) per 32 points and every wave holds 8 x 2
// accumulator tiles (256 registers).  Two halves stage the same 8 activation blocks twice (96 KB of LDS-DMA per product and
// stage instead of 64; the kernel requests 29 KB per point at ~5 TB/s over the chip, MI355X_MICROARCH.md "ldsdma-fill": 6.4 TB/s at
// best) and convert them twice: 10 instead of 12 operand-tile conversions per 48 MFMAs.  43.6-44.0 -> 41.2-41.3 ms per configs[2]
// iteration (profiles/r4_wgrad_experiments.json, which also holds what did NOT help: a three-slot stage ring, fewer
// flushes, 512-thread workgroups with two waves per SIMD).  grid.y = 12: ids 0..7 the whole products (table entries 2 id),
// 8..11 the entries 16..19.  MVIP_WGRAD_WHOLE=0 (host) launches the two-halves form.
constexpr int W_WHOLE_BLOCKS = 17;                         // 8 G + 8 act + the sigma row's d^T block
template <int PREC, int SP, bool WHOLE = false>
__global__ __launch_bounds__(256, SP == 16 ? 2 : 1) void mlp_wgrad_kernel(const Gemm *__restrict__ tab,
                                                          const float *__restrict__ act, int64_t act_n_pt,
                                                          int64_t act_pt0, const float *__restrict__ gst,
                                                          int64_t n_pt, int stages_per_slab,
                                                          const unsigned *__restrict__ absmax_bits) {
    static_assert(!WHOLE || (PREC == 1 && SP == 32), "whole products: the split-precision kernel");
    constexpr int SB = WHOLE ? W_WHOLE_BLOCKS : W_STAGE_BLOCKS;
    __shared__ __attribute__((aligned(16))) float lds[2 * SB * (SP == 32 ? TILE_FLOATS : HALF_FLOATS)];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int id = WHOLE ? (blockIdx.y < 8 ? 2 * blockIdx.y : blockIdx.y + 8) : blockIdx.y;
    const int64_t pt0 = (int64_t)blockIdx.x * stages_per_slab;
    int64_t pt1 = pt0 + stages_per_slab;
    if (pt1 > n_pt) pt1 = n_pt;
    if (pt0 >= pt1) return;
    const Gemm G = tab[id];
    float gs = 1.f, igs = 1.f;
    if constexpr (PREC == 1) {                     // scale max|d_raw| to [64, 128)
        const float mx = __uint_as_float(*absmax_bits);
        int e = ((__float_as_int(mx) >> 23) & 255) - 127;
        if (!(mx > 0.f) || e > 120) e = 6;
        if (e < -110) e = -110;
        gs = __int_as_float((127 + 6 - e) << 23);
        igs = __int_as_float((127 + e - 6) << 23);
    }
    if constexpr (WHOLE) {
        if (id < 16) {                                // both row halves: 8 gradient tiles from g_tile0, rows from n_off = 0
            if (G.extra == 1) wgrad_body<8, 2, 1, true, 1, SP, SB>(G, act, act_n_pt, act_pt0, gst, n_pt, pt0, pt1, lds, wave, lane, gs, igs);
            else wgrad_body<8, 2, 1, true, 0, SP, SB>(G, act, act_n_pt, act_pt0, gst, n_pt, pt0, pt1, lds, wave, lane, gs, igs);
        }
        else if (id < 18) wgrad_body<2, 2, PREC, false, 0, SP, SB>(G, act, act_n_pt, act_pt0, gst, n_pt, pt0, pt1, lds, wave, lane, gs, igs);
        else if (id == 18) wgrad_body<1, 5, PREC, false, 2, SP, SB>(G, act, act_n_pt, act_pt0, gst, n_pt, pt0, pt1, lds, wave, lane, gs, igs);
        else wgrad_body<1, 4, PREC, false, 0, SP, SB>(G, act, act_n_pt, act_pt0, gst, n_pt, pt0, pt1, lds, wave, lane, gs, igs);
        return;
    }
    if (id < 16) {                                    // 256 x 256 products; product 14 also carries the sigma row
        if (G.extra == 1) {
            if constexpr (PREC == 1) wgrad_body<4, 2, 1, true, 1, SP>(G, act, act_n_pt, act_pt0, gst, n_pt, pt0, pt1, lds, wave, lane, gs, igs);
            else wgrad_body<1, 8, 0, false, 1, SP>(G, act, act_n_pt, act_pt0, gst, n_pt, pt0, pt1, lds, wave, lane, gs, igs);
        } else {
            if constexpr (PREC == 1) wgrad_body<4, 2, 1, true, 0, SP>(G, act, act_n_pt, act_pt0, gst, n_pt, pt0, pt1, lds, wave, lane, gs, igs);
            else wgrad_body<1, 8, 0, false, 0, SP>(G, act, act_n_pt, act_pt0, gst, n_pt, pt0, pt1, lds, wave, lane, gs, igs);
        }
    }
    else if (id < 18) wgrad_body<2, 2, PREC, false, 0, SP>(G, act, act_n_pt, act_pt0, gst, n_pt, pt0, pt1, lds, wave, lane, gs, igs);
    else if (id == 18) wgrad_body<1, 5, PREC, false, 2, SP>(G, act, act_n_pt, act_pt0, gst, n_pt, pt0, pt1, lds, wave, lane, gs, igs);
    else wgrad_body<1, 4, PREC, false, 0, SP>(G, act, act_n_pt, act_pt0, gst, n_pt, pt0, pt1, lds, wave, lane, gs, igs);
}

static_assert(sizeof(Gemm) * N_PRODUCTS <= 1024 * 4 - 16, "table room (+ the absmax word)");

// ------------------------------------------------------------------------------------------------
// host orchestration
// ------------------------------------------------------------------------------------------------
static int64_t n_point_tiles(int64_t tile_points) { return ((tile_points + 127) / 128) * 4; }
constexpr int TABLE_FLOATS = 1024;                      // room for the 11-entry product table


static GemmTable make_table(float *const *g) {
    GemmTable t{};
    auto hidden = [&](int slot, int g_tile, int act_tile0, float *dW, int col0, int ldw, float *db) {
        for (int half = 0; half < 2; ++half) {          // rows 0..127 and 128..255 of the product
            Gemm &G = t.g[2 * slot + half];
            G.g_tile0 = g_tile + 4 * half; G.n_off = 128 * half;
            G.a_tile0[0] = act_tile0; G.a_tile0[1] = 0; G.a_count0 = 8;
            G.dW = dW; G.ldw = ldw; G.col0[0] = col0; G.col0[1] = 0; G.valid[0] = 256; G.valid[1] = 0;
            G.db = db; G.extra = 0; G.dWx = G.dbx = nullptr;
        }
    };
    hidden(0, GT_G + 8, AT_H + 0, g[2], 0, 256, g[3]);
    hidden(1, GT_G + 16, AT_H + 8, g[4], 0, 256, g[5]);
    hidden(2, GT_G + 24, AT_H + 16, g[6], 0, 256, g[7]);
    hidden(3, GT_G + 32, AT_H + 24, g[8], 0, 256, g[9]);
    hidden(4, GT_G + 40, AT_H + 32, g[10], 63, 319, g[11]);       // layer 5, h4 columns
    hidden(5, GT_G + 48, AT_H + 40, g[12], 0, 256, g[13]);
    hidden(6, GT_G + 56, AT_H + 48, g[14], 0, 256, g[15]);
    hidden(7, GT_F, AT_H + 56, g[P_WF], 0, 256, g[P_BF]);         // feature_linear ...
    t.g[14].extra = 1; t.g[14].dWx = g[P_WA]; t.g[14].dbx = g[P_BA];   // ... + the sigma row, once
    auto encoded = [&](int id, int l, int ldw, bool bias) {      // products with the 63-channel encoding
        Gemm &G = t.g[id];
        G.g_tile0 = GT_G + 8 * l; G.n_off = 0; G.a_tile0[0] = AT_EMB; G.a_count0 = 2; G.dW = g[2 * l]; G.ldw = ldw;
        G.col0[0] = 0; G.valid[0] = 63; G.db = bias ? g[2 * l + 1] : nullptr; G.extra = 0;
    };
    encoded(16, 0, 63, true);
    encoded(17, 5, 319, false);
    {   // view branch, columns 0..127 of the feature + the 27 direction channels (+ rgb_linear rows, bias)
        Gemm &G = t.g[18];
        G.g_tile0 = GT_V; G.n_off = 0; G.a_tile0[0] = AT_FEAT; G.a_tile0[1] = AT_EDIR; G.a_count0 = 4; G.dW = g[P_WV];
        G.ldw = 283; G.col0[0] = 0; G.col0[1] = 256; G.valid[0] = 128; G.valid[1] = 27; G.db = g[P_BV];
        G.extra = 2; G.dWx = g[P_WR]; G.dbx = g[P_BR];
    }
    {   // view branch, feature columns 128..255
        Gemm &G = t.g[19];
        G.g_tile0 = GT_V; G.n_off = 0; G.a_tile0[0] = AT_FEAT + 4; G.a_count0 = 4; G.dW = g[P_WV]; G.ldw = 283;
        G.col0[0] = 128; G.valid[0] = 128; G.db = nullptr; G.extra = 0;
    }
    return t;
}

// kept_act != nullptr: activations of ALL P points were stashed by the training forward
// (mvip_mlp_forward_*_stash); otherwise they are recomputed tile by tile from (a, b).
static int backward_impl(const float *packed, const float *a, const float *b, int64_t P, int S,
                         const float *d_raw, float *const *grads_host, void *workspace, int64_t tile_points,
                         bool from_rays, const float *kept_act, int precision, void *stream) {
    if (tile_points < 128) return MVIP_EINVAL;
    tile_points = (tile_points / 128) * 128;
    hipStream_t s = as_stream(stream);
    float *ws = reinterpret_cast<float *>(workspace);
    float *packed_t = ws;
    const int64_t n_pt_max = n_point_tiles(tile_points);
    Gemm *tab_dev = reinterpret_cast<Gemm *>(ws + T_FLOATS);
    float *act_ws = ws + T_FLOATS + TABLE_FLOATS;
    float *gst = act_ws + (int64_t)AT_TILES * n_pt_max * TILE_FLOATS;
    // precision 1: `packed` is the f16x3 image; the transposed image is rebuilt from it (hi+lo is exact)
    // exact fp32: the 16-points-per-wave delta kernel with two waves per SIMD (MVIP_DELTA16=0 selects the 32-point one;
    // A-B switch, same results up to summation order)
    static const bool delta16 = [] { const char *e = getenv("MVIP_DELTA16"); return e ? atoi(e) != 0 : true; }();
    if (precision == 1) { int rc = mlp_delta_f16x3_prepare(packed, packed_t, stream); if (rc != MVIP_OK) return rc; }
    else if (delta16) { int rc = mlp_delta16_prepare(packed, packed_t, stream); if (rc != MVIP_OK) return rc; }
    else hipLaunchKernelGGL(mlp_pack_transposed_kernel, dim3((T_FLOATS + 255) / 256), dim3(256), 0, s, packed, packed_t);
    const GemmTable tab = make_table(grads_host);
    hipLaunchKernelGGL(mlp_wgrad_table_kernel, dim3(1), dim3(64), 0, s, tab, tab_dev);
    unsigned *absmax = reinterpret_cast<unsigned *>(ws + T_FLOATS + TABLE_FLOATS - 4);   // tail of the table block
    if (precision == 1) {
        zero_words(absmax, 1, s);
        hipLaunchKernelGGL(absmax_kernel, dim3(256), dim3(256), 0, s, d_raw, P * 4, absmax);
    }
    const int64_t n_pt_all = n_point_tiles(P);
    for (int64_t p0 = 0; p0 < P; p0 += tile_points) {
        const int64_t pc = (P - p0 < tile_points) ? (P - p0) : tile_points;
        const int64_t n_pt = n_point_tiles(pc);
        const float *act = act_ws;
        int64_t act_n_pt = n_pt, act_pt0 = 0;
        if (kept_act) { act = kept_act; act_n_pt = n_pt_all; act_pt0 = p0 / 32; }
        else {
            int rc = precision == 1
                         ? mlp_forward_f16x3_launch(packed, a, b, p0, pc, S, nullptr, act_ws, n_pt, from_rays, stream)
                         : mlp_forward_launch(packed, a, b, p0, pc, S, nullptr, act_ws, n_pt, from_rays, stream);
            if (rc != MVIP_OK) return rc;
        }
        const dim3 grid1((unsigned)(n_pt / 4)), block(256);
        if (precision == 1) {
            int rc = mlp_delta_f16x3_launch(packed_t, packed + SEC_A_FLOATS, d_raw, p0, pc, act, act_n_pt, act_pt0, gst,
                                            n_pt, stream);
            if (rc != MVIP_OK) return rc;
        } else if (delta16) {
            int rc = mlp_delta16_launch(packed_t, packed + SEC_A_FLOATS, d_raw, p0, pc, act, act_n_pt, act_pt0, gst, n_pt, stream);
            if (rc != MVIP_OK) return rc;
        } else
            hipLaunchKernelGGL(mlp_delta_kernel, grid1, block, 0, s, packed_t, packed + SEC_A_FLOATS, d_raw, p0, pc, act,
                               act_n_pt, act_pt0, gst, n_pt);
        // stages (32 points each) per workgroup; every workgroup ends with a flush of its 128 x 256 partial product as
        // fp32 atomics (512 wave-instructions).  Measured (round 2, tools/wgrad_ab.py): 32 / 64 / 128 / 256 stages give
        // 63.2 / 62.7 / 62.9 / 62.8 ms per training iteration -- the flush is not what holds this kernel at 0.6 of the
        // matrix peak.  MVIP_WGRAD_STAGES overrides (tuning).
        static const int sps_cap = [] { const char *e = getenv("MVIP_WGRAD_STAGES"); const int v = e ? atoi(e) : 0; return v > 0 ? v : 64; }();
        int sps = (int)((n_pt + 63) / 64);
        if (sps < 1) sps = 1;
        if (sps > sps_cap) sps = sps_cap;
        const dim3 grid2((unsigned)((n_pt + sps - 1) / sps), N_PRODUCTS);
        // 16-point stages with two workgroups per CU for the exact-fp32 products (round 2, same box: 56.9 vs 58.1 ms per
        // training iteration); the split-precision products are bound by their operand conversions on the VALU and run
        // 1 % slower that way, so they keep 32-point stages.  MVIP_WGRAD_HALF=0/1 forces one shape (tuning / A-B switch).
        static const int half_env = [] { const char *e = getenv("MVIP_WGRAD_HALF"); return e ? atoi(e) : -1; }();
        const int half_stages = half_env >= 0 ? half_env : (precision == 1 ? 0 : 1);
        // split precision: whole 256 x 256 products per workgroup (see mlp_wgrad_kernel); MVIP_WGRAD_WHOLE=0 = two row halves (A/B)
        static const int whole_env = [] { const char *e = getenv("MVIP_WGRAD_WHOLE"); return e ? atoi(e) : 1; }();
        if (precision == 1) {
            if (half_stages) hipLaunchKernelGGL((mlp_wgrad_kernel<1, 16>), grid2, block, 0, s, tab_dev, act, act_n_pt, act_pt0, gst, n_pt, sps, absmax);
            else if (whole_env) {
                const dim3 grid2w((unsigned)((n_pt + sps - 1) / sps), 12);
                hipLaunchKernelGGL((mlp_wgrad_kernel<1, 32, true>), grid2w, block, 0, s, tab_dev, act, act_n_pt, act_pt0, gst, n_pt, sps, absmax);
            } else hipLaunchKernelGGL((mlp_wgrad_kernel<1, 32>), grid2, block, 0, s, tab_dev, act, act_n_pt, act_pt0, gst, n_pt, sps, absmax);
        } else {
            if (half_stages) hipLaunchKernelGGL((mlp_wgrad_kernel<0, 16>), grid2, block, 0, s, tab_dev, act, act_n_pt, act_pt0, gst, n_pt, sps, absmax);
            else hipLaunchKernelGGL((mlp_wgrad_kernel<0, 32>), grid2, block, 0, s, tab_dev, act, act_n_pt, act_pt0, gst, n_pt, sps, absmax);
        }
    }
    return check_launch();
}

}  // namespace mvip

using namespace mvip;

extern "C" int64_t mvip_mlp_backward_workspace_bytes(int64_t tile_points) {
    if (tile_points < 128) return -1;
    const int64_t n_pt = n_point_tiles(tile_points);
    return ((int64_t)T_FLOATS + TABLE_FLOATS + (int64_t)(AT_TILES + GT_TILES) * n_pt * TILE_FLOATS) * 4;
}

extern "C" int mvip_mlp_backward_rays(const float *packed, const float *rows, const float *z, int64_t B, int S,
                                      const float *d_raw, float *const *grads_host, void *workspace,
                                      int64_t tile_points, int precision, void *stream) {
    if (B < 0 || S <= 0) return MVIP_EINVAL;
    if (precision != 0 && precision != 1) return MVIP_EUNSUP;
    if (B == 0) return MVIP_OK;
    if (!packed || !rows || !z || !d_raw || !grads_host || !workspace) return MVIP_EINVAL;
    for (int i = 0; i < P_COUNT; ++i) if (!grads_host[i]) return MVIP_EINVAL;
    return backward_impl(packed, rows, z, B * S, S, d_raw, grads_host, workspace, tile_points, true, nullptr, precision,
                         stream);
}

extern "C" int mvip_mlp_backward_points(const float *packed, const float *pts, const float *dirs, int64_t P,
                                        const float *d_raw, float *const *grads_host, void *workspace,
                                        int64_t tile_points, int precision, void *stream) {
    if (P < 0) return MVIP_EINVAL;
    if (precision != 0 && precision != 1) return MVIP_EUNSUP;
    if (P == 0) return MVIP_OK;
    if (!packed || !pts || !dirs || !d_raw || !grads_host || !workspace) return MVIP_EINVAL;
    for (int i = 0; i < P_COUNT; ++i) if (!grads_host[i]) return MVIP_EINVAL;
    return backward_impl(packed, pts, dirs, P, 1, d_raw, grads_host, workspace, tile_points, false, nullptr, precision,
                         stream);
}

extern "C" int64_t mvip_mlp_stash_floats(int64_t P) {
    return P <= 0 ? 0 : (int64_t)AT_TILES * n_point_tiles(P) * TILE_FLOATS;
}

extern "C" int mvip_mlp_forward_rays_stash(const float *packed, const float *rows, const float *z, int64_t B, int S,
                                           float *raw, float *stash, int precision, void *stream) {
    if (B < 0 || S <= 0) return MVIP_EINVAL;
    if (precision != 0 && precision != 1) return MVIP_EUNSUP;
    if (B == 0) return MVIP_OK;
    if (!packed || !rows || !z || !raw || !stash) return MVIP_EINVAL;
    if (precision == 1)
        return mlp_forward_f16x3_launch(packed, rows, z, 0, B * S, S, raw, stash, n_point_tiles(B * S), true, stream);
    return mlp_forward_launch(packed, rows, z, 0, B * S, S, raw, stash, n_point_tiles(B * S), true, stream);
}

extern "C" int mvip_mlp_forward_points_stash(const float *packed, const float *pts, const float *dirs, int64_t P,
                                             float *raw, float *stash, int precision, void *stream) {
    if (P < 0) return MVIP_EINVAL;
    if (precision != 0 && precision != 1) return MVIP_EUNSUP;
    if (P == 0) return MVIP_OK;
    if (!packed || !pts || !dirs || !raw || !stash) return MVIP_EINVAL;
    if (precision == 1)
        return mlp_forward_f16x3_launch(packed, pts, dirs, 0, P, 1, raw, stash, n_point_tiles(P), false, stream);
    return mlp_forward_launch(packed, pts, dirs, 0, P, 1, raw, stash, n_point_tiles(P), false, stream);
}

extern "C" int mvip_mlp_backward_stash(const float *packed, const float *stash, int64_t P, const float *d_raw,
                                       float *const *grads_host, void *workspace, int64_t tile_points, int precision,
                                       void *stream) {
    if (P < 0) return MVIP_EINVAL;
    if (precision != 0 && precision != 1) return MVIP_EUNSUP;
    if (P == 0) return MVIP_OK;
    if (!packed || !stash || !d_raw || !grads_host || !workspace) return MVIP_EINVAL;
    for (int i = 0; i < P_COUNT; ++i) if (!grads_host[i]) return MVIP_EINVAL;
    return backward_impl(packed, nullptr, nullptr, P, 1, d_raw, grads_host, workspace, tile_points, false, stash,
                         precision, stream);
}
