// Split-precision ("f16x3") inference forward of the 8x256 NeRF MLP with TWO waves per SIMD (round 5).
// (run_network / NeRF.forward of no-grad renders, DS_NeRF/run.py:1108-1124, DS_NeRF/run_nerf_helpers.py:104-127, in the
//  arithmetic of mlp_fwd_f16x3.hip: W x ~= Wh.xh + Wh.xl + Wl.xh on the fp16 matrix pipe, fp32 accumulate.)
//
// mlp_fwd_f16x3.hip gives a wave 32 points (32x32x16 MFMA, 16 registers per accumulator tile, ~450 registers): ONE wave per
// SIMD, and every instruction of that wave that is not an MFMA -- operand reads, waits, the fp32 -> hi/lo conversions of the
// epilogue -- leaves the matrix pipe idle (0.565 busy, profiles/r4_pmc_mfma_util.json).  The exact-fp32 forward got out of
// the same corner with 16 points per wave (mlp_fwd16.hip, 0.92 busy); this is that design on v_mfma_f32_16x16x32_f16:
//   * a wave owns 16 points; an output tile is 16 units x 16 points = 4 accumulator registers; the activations of a layer are
//     8 + 8 fragment quads (hi, lo) = 64 registers, two sets + operands fit 256: a workgroup is 8 waves = 2 per SIMD sharing ONE
//     weight ring, and each wave's conversions and reads issue under its partner's MFMAs;
//   * tools/micro/mfma_shape_wall.hip (profiles/r5_micro_mfma_shape_wall.jsonl) measured the loop this kernel is built around
//     -- 2 ds_read_b128 + 3 MFMAs per k-step, full-entropy operands, all CUs, wall time under the part's power limit:
//     32x32x16 with one wave per SIMD sustains 1.41-1.62 PFLOP/s of fp16 products, 16x16x32 with two waves 1.68-1.74.
// Register trick, 16x16x32 edition: accumulator register i of lane (n = lane & 15, g = lane >> 4) holds unit 4g + i of the tile
// for point n.  A K = 32 step takes 8 halves per lane: element j of lane (n, g) is k-slot (g, j).  Two consecutive 16-unit
// tiles T0 = 2s, T1 = 2s + 1 give the B fragment of k-step s with NO data movement: slot (g, j) = unit 32 s + U(g, j),
// U(g, j) = 4g + j for j < 4 (tile T0's registers), 16 + 4g + (j - 4) for j >= 4 (tile T1's).  The packed image stores the
// A fragments in that k order: block (tile `to`, k-step s, hi | lo) = 64 lanes x 8 halves, lane (m, g) element j =
// W[16 to + m][32 s + U(g, j)].  Same bytes per layer as every other image (1 KB blocks, [Ah | Al] per k-step), same layer
// offsets (mlp_layout.h), streamed through LDS by DMA in chunks, one barrier per chunk.
// Operand reads and the bias reads are issued by hand (asm ds_read_b128) TWO k-steps ahead and waited for with counted
// s_waitcnt lgkmcnt(2): the compiler's own wait insertion answers every pending LDS read with lgkmcnt(0) in front of the next
// MFMA (see mlp_device_f16.h); nothing the compiler knows about touches LDS inside the trunk.
#include <stdlib.h>
#include "mlp_device16.h"

namespace mvip {
using namespace mlp;

namespace f16h {
using f16p::f32x4;
using f16p::ic;
using f16p::static_for;
using f16p::glds;
using f16p::act16;

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int WG_POINTS = 128;
constexpr int KS_L0 = 2, KS_LH = 8, KS_L5 = 10, KS_LV = 9;          // k-steps (32 input units) per output tile
static_assert(16 * KS_L0 * 2 == L0_BLOCKS && 16 * KS_LH * 2 == LH_BLOCKS && 16 * KS_L5 * 2 == L5_BLOCKS &&
              8 * KS_LV * 2 == LV_BLOCKS, "same block counts as every other image");

__host__ __device__ constexpr int unit_of(int g, int j) { return j < 4 ? 4 * g + j : 16 + 4 * g + (j - 4); }

struct APairH { h16x8 h, l; };
struct BPairH { h16x8 hi, lo; };

__device__ __forceinline__ f32x4 mfma32(h16x8 a, h16x8 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// CHB blocks (1 KB) per chunk = per barrier, NSL ring slots
template <int CHB, int NSL>
struct StreamH {
    static_assert(CHB % 8 == 0 && CHB >= 16, "a chunk is cut into one run of blocks per wave");
    static constexpr int chb = CHB, nsl = NSL;
    static constexpr int ring_floats = NSL * CHB * BLOCK_FLOATS;
    const float *img;
    float *lds;
    int wave, lane;
    unsigned lo_addr = 0, hi_addr = 0, sb_addr = 0;     // LDS byte addresses: ring (+ 64 KB), section B + this lane's bias quad
    __device__ __forceinline__ void init() {
        const unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) const float *)lds;
        lo_addr = base + lane * 16;
        unsigned hi = base + 65536u + lane * 16;
        asm volatile("v_mov_b32 %0, %1" : "=v"(hi_addr) : "v"(hi));          // opaque: one live register, immediate offsets off it
        sb_addr = base + ring_floats * 4 + (lane >> 4) * 16;
    }
    template <int C>
    __device__ __forceinline__ void issue_chunk() const {
        if constexpr (C * CHB < TOTAL_BLOCKS) {
            constexpr int nblk = (TOTAL_BLOCKS - C * CHB) < CHB ? (TOTAL_BLOCKS - C * CHB) : CHB;
            constexpr int per_wave = CHB / 8;
            const float *src = img + (int64_t)C * (CHB * BLOCK_FLOATS) + wave * (per_wave * BLOCK_FLOATS) + lane * 4;
            float *dst = lds + (C % NSL) * (CHB * BLOCK_FLOATS) + wave * (per_wave * BLOCK_FLOATS);
            if (nblk == CHB || wave * per_wave < nblk)
                static_for<per_wave>([&](auto p) { glds<decltype(p)::value * 1024>(src, dst); });
        }
    }
    template <int BI>
    __device__ __forceinline__ h16x8 read_async() const {
        constexpr int off = ((BI / CHB) % NSL) * CHB * 1024 + (BI % CHB) * 1024;
        h16x8 v;
        if constexpr (off < 65536) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(lo_addr), "n"(off));
        else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(hi_addr), "n"(off - 65536));
        return v;
    }
    // four consecutive floats of section B at float offset OFF + 4 g (the bias quad of an output tile for this lane's rows)
    template <int OFF>
    __device__ __forceinline__ f32x4 read_sb_async() const {
        f32x4 v;
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(sb_addr), "n"(OFF * 4));
        return v;
    }
};

// fp32 activation quad -> the hi / lo halves at elements 4 HALF .. 4 HALF + 3 of the fragment pair
template <int HALF>
__device__ __forceinline__ void split_into(const f32x4 &v, h16x8 &hi, h16x8 &lo) {
#ifdef MVIP_EXPERIMENT_F16W16_NO_SPLIT                                          // timing experiment only: bits moved, nothing converted
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 &H = reinterpret_cast<u32x4 &>(hi), &L = reinterpret_cast<u32x4 &>(lo);
    H[2 * HALF] = __builtin_bit_cast(unsigned, v[0]); H[2 * HALF + 1] = __builtin_bit_cast(unsigned, v[1]);
    L[2 * HALF] = __builtin_bit_cast(unsigned, v[2]); L[2 * HALF + 1] = __builtin_bit_cast(unsigned, v[3]);
    return;
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const _Float16 hv = (_Float16)v[i];
        hi[4 * HALF + i] = hv;
        lo[4 * HALF + i] = (_Float16)(v[i] - (float)hv);
    }
}

// NaN-preserving ReLU of one value (one integer max; see mlp_device.h)
__device__ __forceinline__ float relu1(float v) {
    const int bits = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, bits > 0 ? bits : 0);
}

// The epilogue of an output tile in STAGES of two vector instructions.  The SIMD has ONE vector issue port: a 16x16x32 MFMA holds
// it for 8 of its 16 cycles, so a gap between two MFMAs hides ~8 cycles of other vector work (MI355X_MICROARCH.md, row
// "vector-instruction ISSUE cost") -- and the unstaged epilogue, a burst of ~20 conversions per tile wherever the compiler
// put it, cost this kernel 16 % (tools/micro/f16w16_variants.hip: 60.1 ms with, 50.5 ms without the conversions).  Staged,
// layer_h places one stage per MFMA gap of the NEXT tile.
//   S0, S1  v = relu(acc)            S2  hi = f16(v)            S3, S4  r = v - f32(hi)  (one v_fma_mix_f32 each: the
//   conversion back is the instruction's own operand widening)            S5  lo = f16(r)
// The stages are INLINE ASSEMBLY: as plain expressions the compiler's instruction selection gathers all of them in front of the
// first use of the finished fragment (one burst again, whatever sched_barrier says about the order of the MFMAs around it),
// turns the residual into a packed v_pk_add_f32 (an anti-lever beside MFMAs, same table) and converts hi back with two
// instructions; volatile asm statements keep their order relative to the operand reads and waits of the k-step.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
template <int HALF, bool RELU, int S>
__device__ __forceinline__ void split_stage(const f32x4 &acc, f32x4 &v, f32x4 &r, h16x8 &hi, h16x8 &lo) {
    if constexpr (S == 0 || S == 1) {
        constexpr int i0 = 2 * S;
        float t0, t1;
        if constexpr (RELU)
            asm volatile("v_max_i32 %0, 0, %2\n\tv_max_i32 %1, 0, %3" : "=&v"(t0), "=&v"(t1) : "v"(acc[i0]), "v"(acc[i0 + 1]));
        else
            asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3" : "=&v"(t0), "=&v"(t1) : "v"(acc[i0]), "v"(acc[i0 + 1]));
        v[i0] = t0; v[i0 + 1] = t1;
    }
    if constexpr (S == 2) {
        unsigned h0, h1;
        asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5" : "=&v"(h0), "=&v"(h1)
                     : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]));
        u32x4 H = __builtin_bit_cast(u32x4, hi);
        H[2 * HALF] = h0; H[2 * HALF + 1] = h1;
        hi = __builtin_bit_cast(h16x8, H);
    }
    if constexpr (S == 3 || S == 4) {                    // r = v - f32(hi): the f16 operand is widened by the instruction itself
        constexpr int i0 = 2 * (S - 3);
        const unsigned hp = __builtin_bit_cast(u32x4, hi)[2 * HALF + (S - 3)];
        float t0, t1;
        asm volatile("v_fma_mix_f32 %0, %2, -1.0, %3 op_sel_hi:[1,0,0]\n\tv_fma_mix_f32 %1, %2, -1.0, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
                     : "=&v"(t0), "=&v"(t1) : "v"(hp), "v"(v[i0]), "v"(v[i0 + 1]));
        r[i0] = t0; r[i0 + 1] = t1;
    }
    if constexpr (S == 5) {
        unsigned l0, l1;
        asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\tv_cvt_pk_f16_f32 %1, %4, %5" : "=&v"(l0), "=&v"(l1)
                     : "v"(r[0]), "v"(r[1]), "v"(r[2]), "v"(r[3]));
        u32x4 L = __builtin_bit_cast(u32x4, lo);
        L[2 * HALF] = l0; L[2 * HALF + 1] = l1;
        lo = __builtin_bit_cast(h16x8, L);
    }
}
// acc += a0 * b0; acc += a1 * b1 (fused, in this order), pinned in place
__device__ __forceinline__ void fmac2(float &acc, float a0, float b0, float a1, float b1) {
    asm volatile("v_fmac_f32 %0, %1, %2\n\tv_fmac_f32 %0, %3, %4" : "+v"(acc) : "v"(a0), "v"(b0), "v"(a1), "v"(b1));
}
__device__ __forceinline__ void relu2(float &o0, float &o1, float i0, float i1) {
    asm volatile("v_max_i32 %0, 0, %2\n\tv_max_i32 %1, 0, %3" : "=&v"(o0), "=&v"(o1) : "v"(i0), "v"(i1));
}

// One layer: NTO output tiles of 16 units x KS k-steps of 32 input units; BASE = absolute index of the layer's first block.
// bsrc(ks) -> (hi, lo) B fragments of k-step ks.  epi(to, stage, acc) runs stage `stage` (0 .. NS - 1) of the epilogue of the
// finished tile `to` (the bias is already in acc: it enters as the C operand of the tile's first MFMA); the stages of tile
// to - 1 are placed one per MFMA gap of tile `to`, from its second k-step on: stage s < LATE in gap s, stage s >= LATE in gap
// 6 + (s - LATE) (the stages that consume section-B values their stage 0 asked for by asynchronous read: two k-step waits
// later those have landed).  a0 / a1 carry the A fragments of the current and the next k-step across tiles and layers, `bias`
// the bias quad of the NEXT tile to start (read one tile ahead; NEXT_BIAS = section-B offset of the bias vector of the layer
// after this one, whose first quad this layer's last tile fetches, or -1).
template <int BASE, int NTO, int KS, bool LAST, int BIAS, int NEXT_BIAS, int NS, int LATE, class St, class BSrc, class Epi>
__device__ __forceinline__ void layer_h(const St &st, APairH &a0, APairH &a1, f32x4 &bias, BSrc bsrc, Epi epi) {
    constexpr int CHB = St::chb;
    constexpr int GAPS = 3 * (KS - 1);                                          // MFMA gaps of a tile after its first k-step
    constexpr int PER = LATE < NS ? 1 : (NS + GAPS - 1) / GAPS;                 // stages per gap (2 in layer 0: two k-steps per tile)
    static_assert(LATE >= NS || (LATE <= 6 && 6 + (NS - LATE) <= GAPS), "late stages need k-steps 3 ...");
    f32x4 accs[2];
    static_for<NTO>([&](auto to_) {
        constexpr int TO = decltype(to_)::value;
        f32x4 &acc = accs[TO & 1];
        static_for<KS>([&](auto ks_) {
            constexpr int K = decltype(ks_)::value;
            constexpr int bi = BASE + 2 * (TO * KS + K);                       // the Ah block of this k-step
            constexpr int left = LAST ? (NTO * KS - (TO * KS + K) - 1) : 1000; // k-steps after this one in the whole stream
            auto gap = [&](auto m_) {                                           // the epilogue stages that belong into gap (K, m)
                if constexpr (TO > 0 && K >= 1) {
                    constexpr int q = 3 * (K - 1) + decltype(m_)::value;
                    static_for<PER>([&](auto r_) {
                        constexpr int e = q * PER + decltype(r_)::value;        // early numbering: stage e in gap e / PER
                        if constexpr (e < LATE && e < NS) epi(ic<TO - 1>{}, ic<e>{}, accs[(TO - 1) & 1]);
                    });
                    if constexpr (q >= 6 && LATE + (q - 6) < NS) epi(ic<TO - 1>{}, ic<LATE + (q - 6)>{}, accs[(TO - 1) & 1]);
                }
            };
            if constexpr (bi % CHB == 0) st.template issue_chunk<bi / CHB + 2>();
            // a0 must have landed; the two reads issued one k-step ago (a1) may still be in flight.  LDS operations complete
            // in order, so "at most two outstanding" also covers the bias quad (and an epilogue's section-B quads) issued before them.
            if constexpr (left >= 1) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(a0.h), "+v"(a0.l), "+v"(bias));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0.h), "+v"(a0.l), "+v"(bias));
            if constexpr (K == 0) acc = bias;
            const BPairH b = bsrc(ks_);
            acc = mfma32(a0.h, b.hi, acc);
            __builtin_amdgcn_sched_barrier(0);
            APairH a2 = a1;
            if constexpr (K == 0) {                                             // the next tile's bias quad, a whole tile ahead
                if constexpr (TO + 1 < NTO) bias = st.template read_sb_async<BIAS + 16 * (TO + 1)>();
                else if constexpr (NEXT_BIAS >= 0) bias = st.template read_sb_async<NEXT_BIAS>();
            }
            if constexpr (left >= 2) a2 = APairH{st.template read_async<bi + 4>(), st.template read_async<bi + 5>()};
            gap(ic<0>{});
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma32(a0.h, b.lo, acc);
            __builtin_amdgcn_sched_barrier(0);
            gap(ic<1>{});
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma32(a0.l, b.hi, acc);
            __builtin_amdgcn_sched_barrier(0);
            gap(ic<2>{});
            __builtin_amdgcn_sched_barrier(0);
#ifndef MVIP_EXPERIMENT_F16W16_NO_BARRIER                                       // timing experiment only: results are wrong without it
            if constexpr ((bi + 2) % CHB == 0) __syncthreads();                 // chunk consumed; the one after next has landed
#endif
            a0 = a1; a1 = a2;
        });
    });
    // the last tile's epilogue has no next tile of this layer to hide under (once per layer); what its early stages asked
    // section B for must have landed before the late ones run
    // (inline assembly reads the accumulators: the compiler's hazard pass does not pace it behind the last MFMA)
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(accs[(NTO - 1) & 1]));
    static_for<NS>([&](auto s_) {
        if constexpr (decltype(s_)::value == LATE)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0.h), "+v"(a0.l), "+v"(a1.h), "+v"(a1.l), "+v"(bias));
        epi(ic<NTO - 1>{}, s_, accs[(NTO - 1) & 1]);
    });
}

// STASH = true is the TRAINING forward (train_precision = 1): every activation tile is also written, in fp32, to the stash the
// split-precision backward kernels read ([row tile of 32 units][point tile of 32][32][32], mlp_device.h -- a wave's 16 x 16 tile
// is four stores of four 64-byte row segments, the two waves that share a point tile filling the other half of each 128-byte
// row), and the ReLU SIGN masks of the trunk and view-branch tiles (all the delta kernel reads of them): the same bytes as
// mlp_forward_f16x3_kernel<.., true> writes.
template <bool FROM_RAYS, bool STASH, int CHB, int NSL>
__global__ void __launch_bounds__(512, 2)
mlp_forward_f16x3_w16_kernel(const float *__restrict__ img, const float *__restrict__ in_a, const float *__restrict__ in_b,
                             int64_t P, int S, float *__restrict__ raw, float *__restrict__ stash = nullptr, int64_t n_pt = 0) {
    using St = StreamH<CHB, NSL>;
    __shared__ __attribute__((aligned(16))) float lds[St::ring_floats + SEC_B_FLOATS];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 15, g = lane >> 4;
    int64_t p = (int64_t)blockIdx.x * WG_POINTS + wave * 16 + n;
    const bool live = p < P;
    if (!live) p = P - 1;

    St st{img, lds, wave, lane};
    st.init();
    for (int b = wave; b < SEC_B_FLOATS / BLOCK_FLOATS; b += 8)
        glds<0>(img + SEC_A_FLOATS + b * BLOCK_FLOATS + lane * 4, lds + St::ring_floats + b * BLOCK_FLOATS);
    st.template issue_chunk<0>();
    st.template issue_chunk<1>();

    float px, py, pz, vx, vy, vz;
    if constexpr (FROM_RAYS) {
        const int64_t ray = p / S;
        const float *row = in_a + ray * 11;
        const float zz = in_b[p];
        px = row[0] + row[3] * zz; py = row[1] + row[4] * zz; pz = row[2] + row[5] * zz;
        vx = row[8]; vy = row[9]; vz = row[10];
    } else {
        px = in_a[p * 3]; py = in_a[p * 3 + 1]; pz = in_a[p * 3 + 2];
        vx = in_b[p * 3]; vy = in_b[p * 3 + 1]; vz = in_b[p * 3 + 2];
    }
    // stash: 16-unit tile `t16` (two per 32-unit row tile) of this wave's 16 points -> rows 16 (t16 & 1) + 4 g + i of the block
    // (row tile t16 >> 1, point tile 4 blockIdx + wave / 2), columns 16 (wave & 1) + n; the block address is wave-uniform.
    // MI >= 0: the row tile's ReLU sign mask -- 16 bits per lane of the 32-point layout (bit 4 q + s <-> row 8 q + 4 hh + s of
    // lane (column, hh)).  This lane (n, g) holds rows 4 g + i of each 16-unit tile, i.e. q = 2 (t16 & 1) + (g >> 1), hh = g & 1:
    // one nibble per tile; the other two nibbles sit in lane l ^ 32 (g ^ 2), one half-wave swap joins them.
    const int64_t pt_wave = (int64_t)blockIdx.x * 4 + (wave >> 1);
    const int stash_lane = (4 * g) * 32 + 16 * (wave & 1) + n;
    unsigned pm = 0;
    auto stash16 = [&](auto t16_, const f32x4 &t, auto mi_) {
        if constexpr (STASH) {
            constexpr int t16 = decltype(t16_)::value, MI = decltype(mi_)::value;
            float *q = stash + ((int64_t)(t16 >> 1) * n_pt + pt_wave) * 1024 + (t16 & 1) * 512 + stash_lane;
            stash_store<true>(q, t[0]); stash_store<true>(q + 32, t[1]); stash_store<true>(q + 64, t[2]); stash_store<true>(q + 96, t[3]);      // non-temporal: mlp_device.h
            if constexpr (MI >= 0) {
                unsigned nib = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) nib |= (t[i] > 0.f ? 1u : 0u) << i;
                const int sh = 4 * (g >> 1) + 8 * (t16 & 1);
                if constexpr ((t16 & 1) == 0) pm = nib << sh;
                else {
                    pm |= nib << sh;
                    const unsigned full = pm | __builtin_bit_cast(unsigned, dpp_xor<32>(__builtin_bit_cast(float, pm)));
                    if (lane < 32) *mask_slot(stash, MI, n_pt, pt_wave, (g & 1) * 32 + 16 * (wave & 1) + n) = (unsigned short)full;
                }
            }
        }
    };
    // encoded point: 64 units (63 + a zero) = two k-steps of B fragments, split like every activation
    h16x8 emb_h[2], emb_l[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        f32x4 q0, q1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#ifdef MVIP_EXPERIMENT_F16W16_NO_ENCODING                                       // timing experiment only (no sin / cos)
            q0[i] = px * (float)(i + 1); q1[i] = py * (float)(i + 1);
#else
            q0[i] = f16p::enc_channel<63>(px, py, pz, 32 * s + 4 * g + i);
            q1[i] = f16p::enc_channel<63>(px, py, pz, 32 * s + 16 + 4 * g + i);
#endif
        }
        split_into<0>(q0, emb_h[s], emb_l[s]);
        split_into<1>(q1, emb_h[s], emb_l[s]);
        if (s == 0) { stash16(ic<2 * AT_EMB>{}, q0, ic<-1>{}); stash16(ic<2 * AT_EMB + 1>{}, q1, ic<-1>{}); }
        else { stash16(ic<2 * AT_EMB + 2>{}, q0, ic<-1>{}); stash16(ic<2 * AT_EMB + 3>{}, q1, ic<-1>{}); }
    }

    __syncthreads();                                   // chunks 0, 1 and section B have landed
    const float *sb = lds + St::ring_floats;
    APairH a0{st.template read_async<0>(), st.template read_async<1>()};
    f32x4 bias = st.template read_sb_async<SB_BIAS>();
    APairH a1{st.template read_async<2>(), st.template read_async<3>()};

    h16x8 h_h[8], h_l[8], o_h[8], o_l[8];
    f32x4 ev, er;                                       // epilogue state between stages: relu'd tile, residual
    // layer 0: 63(+1) -> 256
    layer_h<OFF_L0, 16, KS_L0, false, SB_BIAS, SB_BIAS + 256, 6, 6>(st, a0, a1, bias,
        [&](auto ks) { return BPairH{emb_h[ks.value], emb_l[ks.value]}; },
        [&](auto to, auto sg, const f32x4 &acc) {
            split_stage<to.value & 1, true, sg.value>(acc, ev, er, o_h[to.value >> 1], o_l[to.value >> 1]);
            if constexpr (sg.value == 2) stash16(ic<2 * AT_H + to.value>{}, ev, ic<AT_H + (to.value >> 1)>{});
        });
#pragma unroll
    for (int t = 0; t < 8; ++t) { h_h[t] = o_h[t]; h_l[t] = o_l[t]; }
    // layers 1..4
    static_for<4>([&](auto li) {
        constexpr int l = 1 + decltype(li)::value;
        layer_h<OFF_L1 + (l - 1) * LH_BLOCKS, 16, KS_LH, false, SB_BIAS + l * 256, SB_BIAS + (l + 1) * 256, 6, 6>(st, a0, a1, bias,
            [&](auto ks) { return BPairH{h_h[ks.value], h_l[ks.value]}; },
            [&](auto to, auto sg, const f32x4 &acc) {
                split_stage<to.value & 1, true, sg.value>(acc, ev, er, o_h[to.value >> 1], o_l[to.value >> 1]);
                if constexpr (sg.value == 2) stash16(ic<2 * (AT_H + 8 * l) + to.value>{}, ev, ic<AT_H + 8 * l + (to.value >> 1)>{});
            });
#pragma unroll
        for (int t = 0; t < 8; ++t) { h_h[t] = o_h[t]; h_l[t] = o_l[t]; }
    });
    // layer 5: cat[encoded point (64), h4 (256)] -> 256
    layer_h<OFF_L5, 16, KS_L5, false, SB_BIAS + 5 * 256, SB_BIAS + 6 * 256, 6, 6>(st, a0, a1, bias,
        [&](auto ks) {
            if constexpr (ks.value < 2) return BPairH{emb_h[ks.value], emb_l[ks.value]};
            else return BPairH{h_h[ks.value - 2], h_l[ks.value - 2]};
        },
        [&](auto to, auto sg, const f32x4 &acc) {
            split_stage<to.value & 1, true, sg.value>(acc, ev, er, o_h[to.value >> 1], o_l[to.value >> 1]);
            if constexpr (sg.value == 2) stash16(ic<2 * (AT_H + 40) + to.value>{}, ev, ic<AT_H + 40 + (to.value >> 1)>{});
        });
#pragma unroll
    for (int t = 0; t < 8; ++t) { h_h[t] = o_h[t]; h_l[t] = o_l[t]; }
    // layer 6
    layer_h<OFF_L6, 16, KS_LH, false, SB_BIAS + 6 * 256, SB_BIAS + 7 * 256, 6, 6>(st, a0, a1, bias,
        [&](auto ks) { return BPairH{h_h[ks.value], h_l[ks.value]}; },
        [&](auto to, auto sg, const f32x4 &acc) {
            split_stage<to.value & 1, true, sg.value>(acc, ev, er, o_h[to.value >> 1], o_l[to.value >> 1]);
            if constexpr (sg.value == 2) stash16(ic<2 * (AT_H + 48) + to.value>{}, ev, ic<AT_H + 48 + (to.value >> 1)>{});
        });
#pragma unroll
    for (int t = 0; t < 8; ++t) { h_h[t] = o_h[t]; h_l[t] = o_l[t]; }
    // layer 7; sigma = alpha_linear(h7) accumulated in fp32 from the fp32 activations, tile by tile: stage 0 also asks for the
    // tile's quad of the sigma row, stages 6 and 7 (two k-steps later) consume it
    float sigma = 0.f;
    f32x4 wq;
    layer_h<OFF_L6 + LH_BLOCKS, 16, KS_LH, false, SB_BIAS + 7 * 256, SB_BFEAT, 8, 6>(st, a0, a1, bias,
        [&](auto ks) { return BPairH{h_h[ks.value], h_l[ks.value]}; },
        [&](auto to, auto sg, const f32x4 &acc) {
            constexpr int S_ = decltype(sg)::value;
            if constexpr (S_ == 0) wq = st.template read_sb_async<SB_WALPHA + 16 * decltype(to)::value>();
            if constexpr (S_ < 6) split_stage<to.value & 1, true, S_>(acc, ev, er, o_h[to.value >> 1], o_l[to.value >> 1]);
            if constexpr (S_ == 2) stash16(ic<2 * (AT_H + 56) + decltype(to)::value>{}, ev, ic<AT_H + 56 + (decltype(to)::value >> 1)>{});
            if constexpr (S_ == 6) {
                asm volatile("" : "+v"(wq));                       // ordered behind the k-step wait that covers the read
                fmac2(sigma, wq[0], ev[0], wq[1], ev[1]);
            }
            if constexpr (S_ == 7) fmac2(sigma, wq[2], ev[2], wq[3], ev[3]);
        });
#pragma unroll
    for (int t = 0; t < 8; ++t) { h_h[t] = o_h[t]; h_l[t] = o_l[t]; }
    sigma += __shfl_xor(sigma, 16, 64);
    sigma += __shfl_xor(sigma, 32, 64);
    sigma += sb[SB_BALPHA];
    // feature = feature_linear(h7), no activation
    layer_h<OFF_FEAT, 16, KS_LH, false, SB_BFEAT, SB_BVIEWS, 6, 6>(st, a0, a1, bias,
        [&](auto ks) { return BPairH{h_h[ks.value], h_l[ks.value]}; },
        [&](auto to, auto sg, const f32x4 &acc) {
            split_stage<to.value & 1, false, sg.value>(acc, ev, er, o_h[to.value >> 1], o_l[to.value >> 1]);
            if constexpr (sg.value == 2) stash16(ic<2 * AT_FEAT + to.value>{}, ev, ic<-1>{});
        });
    // view branch: cat[feature (256), encoded direction (27 + 5)] -> 128, relu; rgb = rgb_linear(v) in the epilogue.
    // The direction encoding is formed only now (8 fewer live registers through the trunk).
    h16x8 ed_h, ed_l;
    {
        f32x4 q0, q1;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#ifdef MVIP_EXPERIMENT_F16W16_NO_ENCODING
            q0[i] = vx * (float)(i + 1); q1[i] = vy * (float)(i + 1);
#else
            q0[i] = f16p::enc_channel<27>(vx, vy, vz, 4 * g + i);
            q1[i] = f16p::enc_channel<27>(vx, vy, vz, 16 + 4 * g + i);
#endif
        }
        split_into<0>(q0, ed_h, ed_l);
        split_into<1>(q1, ed_h, ed_l);
        stash16(ic<2 * AT_EDIR>{}, q0, ic<-1>{});
        stash16(ic<2 * AT_EDIR + 1>{}, q1, ic<-1>{});
    }
    float r0 = 0.f, r1 = 0.f, r2 = 0.f;
    f32x4 w0, w1, w2;
    layer_h<OFF_VIEWS, 8, KS_LV, true, SB_BVIEWS, -1, 8, 2>(st, a0, a1, bias,
        [&](auto ks) {
            if constexpr (ks.value < 8) return BPairH{o_h[ks.value], o_l[ks.value]};
            else return BPairH{ed_h, ed_l};
        },
        [&](auto to, auto sg, const f32x4 &acc) {
            constexpr int S_ = decltype(sg)::value, TO_ = decltype(to)::value;
            if constexpr (S_ == 0) {                                // the three rgb rows' quads of this tile, and relu
                w0 = st.template read_sb_async<SB_WRGB + 16 * TO_>();
                w1 = st.template read_sb_async<SB_WRGB + 128 + 16 * TO_>();
                w2 = st.template read_sb_async<SB_WRGB + 256 + 16 * TO_>();
                float t0, t1;
                relu2(t0, t1, acc[0], acc[1]);
                ev[0] = t0; ev[1] = t1;
            }
            if constexpr (S_ == 1) {
                float t0, t1;
                relu2(t0, t1, acc[2], acc[3]);
                ev[2] = t0; ev[3] = t1;
                stash16(ic<2 * AT_V + TO_>{}, ev, ic<64 + (TO_ >> 1)>{});
            }
            if constexpr (S_ == 2) {
                asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2));
                fmac2(r0, w0[0], ev[0], w0[1], ev[1]);
            }
            if constexpr (S_ == 3) fmac2(r0, w0[2], ev[2], w0[3], ev[3]);
            if constexpr (S_ == 4) fmac2(r1, w1[0], ev[0], w1[1], ev[1]);
            if constexpr (S_ == 5) fmac2(r1, w1[2], ev[2], w1[3], ev[3]);
            if constexpr (S_ == 6) fmac2(r2, w2[0], ev[0], w2[1], ev[1]);
            if constexpr (S_ == 7) fmac2(r2, w2[2], ev[2], w2[3], ev[3]);
        });
    r0 += __shfl_xor(r0, 16, 64); r1 += __shfl_xor(r1, 16, 64); r2 += __shfl_xor(r2, 16, 64);
    r0 += __shfl_xor(r0, 32, 64); r1 += __shfl_xor(r1, 32, 64); r2 += __shfl_xor(r2, 32, 64);
    if (live && g == 0 && raw)
        reinterpret_cast<float4 *>(raw)[p] = make_float4(r0 + sb[SB_BRGB], r1 + sb[SB_BRGB + 1], r2 + sb[SB_BRGB + 2], sigma);
}

// ---- packing ---------------------------------------------------------------------------------------------------
struct ParamPtrsH { const float *p[P_COUNT]; };

__device__ __forceinline__ float weight_at_h(const ParamPtrsH &pp, int layer_blk_off, int row, int k) {
    // (layer identified by its block offset) -> W[row][k] with the fp32 image's K padding rules
    if (layer_blk_off == OFF_L0) return k < 63 ? pp.p[P_W0][row * 63 + k] : 0.f;
    if (layer_blk_off == OFF_L5) {
        if (k < 63) return pp.p[10][row * 319 + k];
        if (k >= 64) return pp.p[10][row * 319 + 63 + (k - 64)];
        return 0.f;
    }
    if (layer_blk_off == OFF_VIEWS) return k < 283 ? pp.p[P_WV][row * 283 + k] : 0.f;
    if (layer_blk_off == OFF_FEAT) return pp.p[P_WF][row * 256 + k];
    if (layer_blk_off >= OFF_L6) return pp.p[2 * (6 + (layer_blk_off - OFF_L6) / LH_BLOCKS)][row * 256 + k];
    return pp.p[2 * (1 + (layer_blk_off - OFF_L1) / LH_BLOCKS)][row * 256 + k];
}

__global__ void mlp_pack_f16x3_w16_kernel(ParamPtrsH pp, _Float16 *__restrict__ img) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;          // one fp16 element of section A
    if (idx >= SEC_A_FLOATS * 2) return;
    const int blk = idx / 512, e = idx % 512;                        // 512 halves per 1-KB block
    const int lane = e / 8, j = e % 8, m = lane & 15, g = lane >> 4;
    int off, ks;                                                      // layer block offset, k-steps per tile
    if (blk < OFF_L1) { off = OFF_L0; ks = KS_L0; }
    else if (blk < OFF_L5) { off = OFF_L1 + ((blk - OFF_L1) / LH_BLOCKS) * LH_BLOCKS; ks = KS_LH; }
    else if (blk < OFF_L6) { off = OFF_L5; ks = KS_L5; }
    else if (blk < OFF_FEAT) { off = OFF_L6 + ((blk - OFF_L6) / LH_BLOCKS) * LH_BLOCKS; ks = KS_LH; }
    else if (blk < OFF_VIEWS) { off = OFF_FEAT; ks = KS_LH; }
    else { off = OFF_VIEWS; ks = KS_LV; }
    const int local = blk - off, pair = local / 2, lo = local % 2;   // [Ah | Al] per k-step
    const int to = pair / ks, s = pair % ks;
    const float w = weight_at_h(pp, off, 16 * to + m, 32 * s + unit_of(g, j));
    const _Float16 wh = (_Float16)w;
    img[idx] = lo ? (_Float16)(w - (float)wh) : wh;
}

template <bool FROM_RAYS>
static int launch_w16(const float *img, const float *a, const float *b, int64_t P, int S, float *raw, float *stash, void *stream) {
    // ring geometry: 4 slots of 16 KB (77 KB of LDS: a second workgroup's allocation fits beside a finishing one, see the LDS
    // note in mlp_fwd16.hip).  -DMVIP_F16W16_BOTH_RINGS also builds 3 slots of 32 KB (half the barriers, 109 KB; MVIP_F16W16_RING=1).
    static const int ring = [] { const char *e = getenv("MVIP_F16W16_RING"); return e ? atoi(e) : 0; }();
    (void)ring;
    const int64_t wgs = (P + WG_POINTS - 1) / WG_POINTS;
    const dim3 grid((unsigned)wgs), block(512);
    hipStream_t s = as_stream(stream);
    if (stash) {
        if constexpr (FROM_RAYS)
            hipLaunchKernelGGL((mlp_forward_f16x3_w16_kernel<true, true, 16, 4>), grid, block, 0, s, img, a, b, P, S, raw, stash, wgs * 4);
        else return MVIP_EUNSUP;
        return check_launch();
    }
#ifdef MVIP_F16W16_BOTH_RINGS                        // A/B builds only: the 3 x 32 KB ring measured equal (profiles/r5_f16x3_w16_experiments.json)
    if (ring == 1) hipLaunchKernelGGL((mlp_forward_f16x3_w16_kernel<FROM_RAYS, false, 32, 3>), grid, block, 0, s, img, a, b, P, S, raw, (float *)nullptr, (int64_t)0);
    else
#endif
    hipLaunchKernelGGL((mlp_forward_f16x3_w16_kernel<FROM_RAYS, false, 16, 4>), grid, block, 0, s, img, a, b, P, S, raw, (float *)nullptr, (int64_t)0);
    return check_launch();
}

}  // namespace f16h
}  // namespace mvip

using namespace mvip;

// Image of the two-waves-per-SIMD split-precision forward: [section A as fp16 hi / lo fragments in 16x16x32 order | section B
// copied from the fp32 image]; PACKED_FLOATS floats like every other image.
extern "C" int mvip_mlp_pack_f16x3_w16(const float *const *params_host, const float *packed_f32, float *image, void *stream) {
    if (!params_host || !image || !packed_f32) return MVIP_EINVAL;
    f16h::ParamPtrsH pp;
    for (int i = 0; i < mlp::P_COUNT; ++i) {
        if (!params_host[i]) return MVIP_EINVAL;
        pp.p[i] = params_host[i];
    }
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(f16h::mlp_pack_f16x3_w16_kernel, dim3((mlp::SEC_A_FLOATS * 2 + 255) / 256), dim3(256), 0, s, pp,
                       reinterpret_cast<_Float16 *>(image));
    if (hipMemcpyAsync(image + mlp::SEC_A_FLOATS, packed_f32 + mlp::SEC_A_FLOATS, sizeof(float) * mlp::SEC_B_FLOATS,
                       hipMemcpyDeviceToDevice, s) != hipSuccess)
        return check_launch();
    return check_launch();
}

// raw [B, S, 4] of the rays' sample points (rows [B, 11], depths z [B, S]): NeRF.forward under no_grad at
// inference_precision = 1 (DS_NeRF/run.py:1108-1124); same values as mvip_mlp_forward_rays_f16x3 up to fp32 summation order.
extern "C" int mvip_mlp_forward_rays_f16x3_w16(const float *image, const float *rows, const float *z, int64_t B, int S,
                                               float *raw, void *stream) {
    if (B < 0 || S <= 0) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    if (!image || !rows || !z || !raw) return MVIP_EINVAL;
    return f16h::launch_w16<true>(image, rows, z, B * S, S, raw, nullptr, stream);
}

extern "C" int mvip_mlp_forward_points_f16x3_w16(const float *image, const float *pts, const float *dirs, int64_t P,
                                                 float *raw, void *stream) {
    if (P < 0) return MVIP_EINVAL;
    if (P == 0) return MVIP_OK;
    if (!image || !pts || !dirs || !raw) return MVIP_EINVAL;
    return f16h::launch_w16<false>(image, pts, dirs, P, 1, raw, nullptr, stream);
}

// Training forward (train_precision = 1) on the two-wave kernel: raw AND the activation stash of mvip_mlp_stash_floats(B*S) floats
// (fp32 tiles + ReLU sign masks) that mvip_mlp_backward_stash(precision = 1) consumes -- the bytes mvip_mlp_forward_rays_stash
// (precision = 1) writes, up to fp32 summation order (DS_NeRF/run.py:948-974 renders under loss.backward(), :1030).
extern "C" int mvip_mlp_forward_rays_stash_f16x3_w16(const float *image, const float *rows, const float *z, int64_t B, int S,
                                                     float *raw, float *stash, void *stream) {
    if (B < 0 || S <= 0) return MVIP_EINVAL;
    if (B == 0) return MVIP_OK;
    if (!image || !rows || !z || !raw || !stash) return MVIP_EINVAL;
    return f16h::launch_w16<true>(image, rows, z, B * S, S, raw, stash, stream);
}
