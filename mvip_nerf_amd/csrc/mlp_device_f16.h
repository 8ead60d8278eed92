// Device-side building blocks of the split-precision ("f16x3") kernels; see mlp_fwd_f16x3.hip.
#pragma once
#include "common.h"
#include "mlp_layout.h"
#include "mlp_device.h"

namespace mvip {
using namespace mlp;

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int F_GROUP_BLOCKS = 64;                         // blocks (1 KB) per barrier group
constexpr int F_RING_FLOATS = 2 * F_GROUP_BLOCKS * BLOCK_FLOATS;   // 128 KB
constexpr int F_LDS_FLOATS = F_RING_FLOATS + SEC_B_FLOATS;
constexpr int F_TOTAL_GROUPS = (TOTAL_BLOCKS + F_GROUP_BLOCKS - 1) / F_GROUP_BLOCKS;
static_assert(OFF_L1 % F_GROUP_BLOCKS == 0 && OFF_L5 % F_GROUP_BLOCKS == 0 && OFF_L6 % F_GROUP_BLOCKS == 0 &&
              OFF_FEAT % F_GROUP_BLOCKS == 0 && OFF_VIEWS % F_GROUP_BLOCKS == 0 && LH_BLOCKS % F_GROUP_BLOCKS == 0,
              "layers must start on barrier-group boundaries");

struct StreamF {
    const float *img;       // f16x3 image viewed as floats (same byte layout granularity: 256 floats = 1 KB)
    float *lds;
    int wave, lane;
    int total_blocks = TOTAL_BLOCKS;
    // issue the 64 blocks of barrier group g into ring half (g & 1): wave w stages the 16 consecutive blocks
    // 16w .. 16w+15 as four runs of four, each run sharing one address computation and one M0 write through the
    // instruction's immediate offset (every extra SALU/VALU instruction costs this single-wave-per-SIMD kernel
    // ~4 cycles of matrix-pipe idle time)
    __device__ __forceinline__ void issue_group(int g) const {
        if (g * F_GROUP_BLOCKS < total_blocks) {
            const int nblk = (total_blocks - g * F_GROUP_BLOCKS) < F_GROUP_BLOCKS ? (total_blocks - g * F_GROUP_BLOCKS)
                                                                                 : F_GROUP_BLOCKS;
            const float *src = img + (int64_t)g * F_GROUP_BLOCKS * BLOCK_FLOATS + wave * (16 * BLOCK_FLOATS) + lane * 4;
            float *dst = lds + (g & 1) * (F_GROUP_BLOCKS * BLOCK_FLOATS) + wave * (16 * BLOCK_FLOATS);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (wave * 16 + 4 * c < nblk) {              // groups are whole multiples of 4 blocks
                    glds16o<0>(src + c * 4 * BLOCK_FLOATS, dst + c * 4 * BLOCK_FLOATS);
                    glds16o<1024>(src + c * 4 * BLOCK_FLOATS, dst + c * 4 * BLOCK_FLOATS);
                    glds16o<2048>(src + c * 4 * BLOCK_FLOATS, dst + c * 4 * BLOCK_FLOATS);
                    glds16o<3072>(src + c * 4 * BLOCK_FLOATS, dst + c * 4 * BLOCK_FLOATS);
                }
            }
        }
    }
    // The 128 KB ring is wider than ds_read's 16-bit immediate offset.  Left alone the compiler materialises a
    // separate address for every block of the upper half (and parks those addresses in AGPRs): two extra
    // instructions per operand read.  `hi` is the lane's address in the upper half made opaque to the
    // optimiser, so every read is `ds_read_b128 v, base offset:imm` off one of two live registers.
    unsigned hi_addr = 0, lo_addr = 0;
    __device__ __forceinline__ void init_bases() {
        unsigned a = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float *)(lds + F_GROUP_BLOCKS * BLOCK_FLOATS) + lane * 16;
        asm volatile("v_mov_b32 %0, %1" : "=v"(hi_addr) : "v"(a));
        lo_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float *)lds + lane * 16;
    }
    // Operand prefetch the compiler's wait-count pass does not see: the pass answers every pending LDS read with
    // s_waitcnt lgkmcnt(0) in front of the next MFMA, i.e. it waits out the prefetch issued one MFMA earlier --
    // a full LDS latency per 96-cycle k-step (matrix pipe 48 % busy).  These reads are paired with explicit
    // s_waitcnt lgkmcnt(N) in run_layer_f (LDS operations complete in order).
    template <int BLK>
    __device__ __forceinline__ h16x8 read_block_async() const {
        constexpr int blk = BLK % (2 * F_GROUP_BLOCKS);
        h16x8 v;
        if constexpr (blk < F_GROUP_BLOCKS)
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lo_addr), "n"(blk * 1024));
        else
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(hi_addr), "n"((blk - F_GROUP_BLOCKS) * 1024));
        return v;
    }
    template <int BLK>      // BLK = absolute block index in the stream (compile time)
    __device__ __forceinline__ h16x8 read_block() const {
        constexpr int blk = BLK % (2 * F_GROUP_BLOCKS);
        if constexpr (blk < F_GROUP_BLOCKS) {
            return *reinterpret_cast<const h16x8 *>(lds + blk * BLOCK_FLOATS + lane * 4);
        } else {
            typedef __attribute__((address_space(3))) const h16x8 *lds_ptr;
            return *(lds_ptr)(uintptr_t)(hi_addr + (blk - F_GROUP_BLOCKS) * (BLOCK_FLOATS * 4));
        }
    }
};

__device__ __forceinline__ f32x16 mfma16(h16x8 a, h16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

struct Frag { h16x8 hi[2], lo[2]; };            // one 32-unit activation tile as B operands (k-steps 0,1)

__device__ __forceinline__ Frag split_tile(const f32x16 &x) {
    // (a v_fma_mix_f32 per element for the residual was tried: fewer conversions, but the register allocator
    //  answered with more AGPR<->VGPR moves and the kernel got 3 % slower)
    Frag f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = x[8 * s + j];
            const _Float16 hv = (_Float16)v;
            f.hi[s][j] = hv;
            f.lo[s][j] = (_Float16)(v - (float)hv);
        }
    return f;
}

// bias + (optional) ReLU for this mode: v_max_f32 instead of compare+select
template <bool RELU>
__device__ __forceinline__ f32x16 bias_act(const f32x16 &acc, const float *bias32, int hh) {
    f32x16 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(bias32 + 8 * q + 4 * hh);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float x = acc[4 * q + s] + b[s];
            r[4 * q + s] = RELU ? fmaxf(x, 0.f) : x;
        }
    }
    return r;
}

// activation only (the bias came in through the MFMA C operand, see run_layer_f)
template <bool RELU>
__device__ __forceinline__ f32x16 act_only(const f32x16 &acc) {
    f32x16 r;
#pragma unroll
    for (int q = 0; q < 16; ++q) r[q] = RELU ? fmaxf(acc[q], 0.f) : acc[q];
    return r;
}

struct APair { h16x8 h, l; };      // [Ah | Al] fragments of one k-step

// One layer: NT output tiles, KS k-steps of 16; BASE = absolute block offset of the layer.
// bfrag(ks) -> (hi, lo) B fragments of k-step ks.  A operands are read from LDS TWO k-steps ahead
// (a k-step is only 96 MFMA cycles, less than the loaded LDS latency); `a0`/`a1` carry the fragments
// of the current and the next k-step across tiles and layers.  Reads never cross a barrier-group
// boundary early: the next group is only guaranteed to have landed after its barrier.
// bias32 (optional): the layer's bias vector in LDS; it enters as the C operand of each tile's first MFMA, so the
// epilogue has no additions left (acc register 4q + s of lane half hh holds output unit 32 T + 8q + 4hh + s).
template <int BASE, int NT, int KS, bool LAST, class BFrag, class Pre, class Epi>
__device__ __forceinline__ void run_layer_f(const StreamF &st, APair &a0, APair &a1, BFrag bfrag, Pre pre, Epi epi,
                                            const float *bias32 = nullptr) {
    using PV = decltype(pre(ic<0>{}));
    f32x16 accs[2];
    PV pvs[2];
    static_for<NT>([&](auto ti) {
        constexpr int T = decltype(ti)::value;
        pvs[T & 1] = pre(ti);
        f32x16 &acc = accs[T & 1];
        if (bias32) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 b = *reinterpret_cast<const f32x4 *>(bias32 + 32 * T + 8 * q + 4 * (st.lane >> 5));
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[4 * q + r] = b[r];
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
        static_for<KS>([&](auto ks) {
            constexpr int K = decltype(ks)::value;
            constexpr int blk = BASE + 2 * (T * KS + K);                     // Ah block of this k-step
            constexpr int left = LAST ? (NT * KS - (T * KS + K) - 1) : 1000;  // k-steps after this one
            constexpr bool group_end = (blk + 2) % F_GROUP_BLOCKS == 0;       // this is the last k-step of its group
            constexpr bool next_is_group_end = (blk + 4) % F_GROUP_BLOCKS == 0;
            if constexpr (blk % F_GROUP_BLOCKS == 0) st.issue_group(blk / F_GROUP_BLOCKS + 1);
            // a0 (this step's fragments) must have landed.  If the previous step issued an asynchronous prefetch
            // (a1's two reads), exactly those two may still be in flight: lgkmcnt(2); otherwise drain.
            constexpr bool prev_async = (blk % F_GROUP_BLOCKS != 0) && ((blk + 2) % F_GROUP_BLOCKS != 0) && left >= 1;
            if constexpr (prev_async) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a0.h), "+v"(a0.l));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0.h), "+v"(a0.l));
            APair a2 = a1;
            // fragments of k-step +2 live in the same group iff neither this nor the next step ends it
            if constexpr (left >= 2 && !group_end && !next_is_group_end)
                a2 = APair{st.template read_block_async<blk + 4>(), st.template read_block_async<blk + 5>()};
            const auto b = bfrag(ks);
            acc = mfma16(a0.h, b.first, acc);
            acc = mfma16(a0.h, b.second, acc);
            acc = mfma16(a0.l, b.first, acc);
            if constexpr (group_end) {
                __syncthreads();                                             // next group landed, this half is free
                if constexpr (left >= 1) a1 = APair{st.template read_block<blk + 2>(), st.template read_block<blk + 3>()};
                if constexpr (left >= 2) a2 = APair{st.template read_block<blk + 4>(), st.template read_block<blk + 5>()};
            } else if constexpr (next_is_group_end) {
                a2 = a1;                                                      // refilled after the next barrier
            }
            a0 = a1; a1 = a2;
            if constexpr (K == 1 && T > 0) epi(ic<T - 1>{}, accs[(T - 1) & 1], pvs[(T - 1) & 1]);
        });
    });
    epi(ic<NT - 1>{}, accs[(NT - 1) & 1], pvs[(NT - 1) & 1]);
}

struct FragPair { h16x8 first, second; };


}  // namespace mvip
