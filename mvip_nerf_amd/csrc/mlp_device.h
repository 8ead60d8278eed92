// Device-side building blocks shared by the fused MLP forward and backward kernels.
#pragma once
#include <utility>
#include <type_traits>
#include "common.h"
#include "mlp_layout.h"

namespace mvip {
using namespace mlp;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NSLOT = 4;                                  // LDS ring slots (16 KB each)
constexpr int RING_FLOATS = NSLOT * CHUNK_FLOATS;         // 64 KB
constexpr int LDS_FLOATS = RING_FLOATS + SEC_B_FLOATS;    // + 13 KB small vectors

template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, I...>) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
    static_for_impl(f, std::make_integer_sequence<int, N>{});
}
template <int V> using ic = std::integral_constant<int, V>;

// one global_load_lds_dwordx4: 64 lanes x 16 B -> 1 KB at the wave-uniform LDS address `dst`
__device__ __forceinline__ void glds16(const float *src_lane, float *dst_wave) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src_lane,
                                     (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
}

// the same with an immediate byte offset that the instruction adds to BOTH addresses: the four 1-KB pieces a
// wave stages per chunk share one address computation and one M0 write
template <int OFF>
__device__ __forceinline__ void glds16o(const float *src_lane, float *dst_wave) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src_lane,
                                     (__attribute__((address_space(3))) void *)dst_wave, 16, OFF, 0);
}

// The weight stream: section A of a packed image consumed front to back through the LDS ring.
// Protocol (g = chunk being computed): loads of chunk g+2 are issued at the first block of
// chunk g; the __syncthreads at the end of chunk g (which waits vmcnt(0)) makes chunk g+2
// visible to every wave.  Hence chunk g+1 is already complete while g is computed, which is
// what allows reading the first block of g+1 before that barrier (A-operand prefetch).
struct APair32 { f32x4 x, y; };          // A operands of the two tiles of a pair for one k-group

struct Stream {
    const float *packed;     // section A base (a section B may follow at SEC_A_FLOATS)
    float *lds;
    int wave, lane;

    __device__ __forceinline__ void issue_chunk(int g, int slot, int total_chunks = TOTAL_CHUNKS) const {
        if (g < total_chunks) {
            const float *src = packed + (int64_t)g * CHUNK_FLOATS + wave * (4 * BLOCK_FLOATS) + lane * 4;
            float *dst = lds + slot * CHUNK_FLOATS + wave * (4 * BLOCK_FLOATS);
            glds16o<0>(src, dst);
            glds16o<1024>(src, dst);
            glds16o<2048>(src, dst);
            glds16o<3072>(src, dst);
        }
    }
    __device__ __forceinline__ void load_section_b(const float *secb) const {
        for (int b = wave; b < SEC_B_FLOATS / BLOCK_FLOATS; b += 4)
            glds16(secb + b * BLOCK_FLOATS + lane * 4, lds + RING_FLOATS + b * BLOCK_FLOATS);
    }
    __device__ __forceinline__ void prologue(const float *secb, int total_chunks = TOTAL_CHUNKS) const {
        load_section_b(secb);
        issue_chunk(0, 0, total_chunks);
        issue_chunk(1, 1, total_chunks);
    }
    template <int BI>
    __device__ __forceinline__ f32x4 read_block() const {
        constexpr int off = ((BI / CHUNK_BLOCKS) % NSLOT) * CHUNK_FLOATS + (BI % CHUNK_BLOCKS) * BLOCK_FLOATS;
        return *reinterpret_cast<const f32x4 *>(lds + off + lane * 4);
    }
    __device__ __forceinline__ f32x4 first_block() const { return read_block<0>(); }
    __device__ __forceinline__ struct APair32 first_pair() const;
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ APair32 Stream::first_pair() const { return APair32{read_block<0>(), read_block<1>()}; }

// One linear layer on the wave's 32 columns: NT output tiles x KG k-groups, A from the ring,
// B from registers through bop(kg, s).  Tiles run in PAIRS on two independent accumulator chains whose
// MFMAs alternate (see mlp_layout.h), the stream holding the pair's blocks interleaved.  pre(ti) runs
// before a tile's chain (e.g. to start loads the epilogue needs) and its result is handed to
// epi(ti, acc, pre_value), which consumes each finished 32x32 tile.  `g0` = chunk index of the layer's
// first block (a multiple of NSLOT).
template <int NT, int KG, bool LAST, class BOp, class Pre, class Epi>
__device__ __forceinline__ void run_layer(const Stream &st, int g0, APair32 &a, BOp bop, Pre pre, Epi epi,
                                          int total_chunks = TOTAL_CHUNKS) {
    static_assert(NT % 2 == 0, "tiles are consumed in pairs");
    // Two pairs of accumulator tiles alternate and the epilogues of pair P-1 are issued two k-groups into
    // pair P: their accumulators are long complete by then, so no MFMA->read wait states are needed and
    // the VALU work hides under the MFMA chains of the current pair.
    using PV = decltype(pre(ic<0>{}));
    f32x16 accs[4];
    PV pvs[4];
    static_for<NT / 2>([&](auto pi) {
        constexpr int P = decltype(pi)::value;
        constexpr int S = 2 * (P & 1);
        pvs[S] = pre(ic<2 * P>{});
        pvs[S + 1] = pre(ic<2 * P + 1>{});
        f32x16 &acc0 = accs[S], &acc1 = accs[S + 1];
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
        static_for<KG>([&](auto kg) {
            constexpr int bi = P * 2 * KG + 2 * decltype(kg)::value;          // blocks bi (tile 2P), bi+1 (tile 2P+1)
            constexpr bool last_step = LAST && (bi + 1 == NT * KG - 1);
            acc0 = mfma(a.x[0], bop(kg, ic<0>{}), acc0);
            acc1 = mfma(a.y[0], bop(kg, ic<0>{}), acc1);
            // The next step's two A operands are read HERE, six MFMAs (384 cycles) ahead of their use, and pinned
            // by scheduling barriers: left to itself the scheduler sinks the reads to just before the next
            // step's first MFMA (s_waitcnt lgkmcnt(0) right behind them), and issued at the very top of the step
            // the compiler's wait for the CURRENT operands (lgkmcnt(0)) would cover them as well.
            __builtin_amdgcn_sched_barrier(0);
            APair32 an = a;
            if constexpr (!last_step) an = APair32{st.template read_block<bi + 2>(), st.template read_block<bi + 3>()};
            __builtin_amdgcn_sched_barrier(0);
            acc0 = mfma(a.x[1], bop(kg, ic<1>{}), acc0);
            acc1 = mfma(a.y[1], bop(kg, ic<1>{}), acc1);
            if constexpr (bi % CHUNK_BLOCKS == 0) {              // chunk g+2 is staged during the first step of chunk g
                st.issue_chunk(g0 + bi / CHUNK_BLOCKS + 2, (bi / CHUNK_BLOCKS + 2) % NSLOT, total_chunks);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc0 = mfma(a.x[2], bop(kg, ic<2>{}), acc0);
            acc1 = mfma(a.y[2], bop(kg, ic<2>{}), acc1);
            acc0 = mfma(a.x[3], bop(kg, ic<3>{}), acc0);
            acc1 = mfma(a.y[3], bop(kg, ic<3>{}), acc1);
#ifdef MVIP_EXPERIMENT_NO_BARRIER          // timing experiment only: results are wrong without the barrier
            if constexpr ((bi + 1) % CHUNK_BLOCKS == CHUNK_BLOCKS - 1) __builtin_amdgcn_s_waitcnt(0);
#else
            if constexpr ((bi + 1) % CHUNK_BLOCKS == CHUNK_BLOCKS - 1) __syncthreads();
#endif
            a = an;
            if constexpr (decltype(kg)::value == 1 && P > 0) {
                constexpr int Q = 2 * ((P - 1) & 1);
                epi(ic<2 * P - 2>{}, accs[Q], pvs[Q]);
                epi(ic<2 * P - 1>{}, accs[Q + 1], pvs[Q + 1]);
            }
        });
    });
    constexpr int Q = 2 * ((NT / 2 - 1) & 1);
    epi(ic<NT - 2>{}, accs[Q], pvs[Q]);
    epi(ic<NT - 1>{}, accs[Q + 1], pvs[Q + 1]);
}
struct NoPre { template <class T> __device__ __forceinline__ int operator()(T) const { return 0; } };

// accumulator tile (+ bias row, natural unit order) -> activation tile
// The epilogue is VALU work squeezed between MFMAs (tools/clock_probe.py: ~450 cycles per tile, 60 % of the
// kernel's non-MFMA time when ReLU was compare + select), so it is kept to the fewest instructions: packed
// fp32 adds (v_pk_add_f32, two elements each) and ReLU as ONE integer max: for IEEE floats max_i32(bits, 0)
// is x for x > 0, +0 for x <= -0 and keeps a (positive-sign) NaN a NaN like x < 0 ? 0 : x does.
template <bool RELU>
__device__ __forceinline__ f32x16 bias_relu(const f32x16 &acc, const float *bias32, int hh) {
#ifdef MVIP_EXPERIMENT_NO_EPILOGUE         // timing experiment only
    return acc;
#endif
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x16 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(bias32 + 8 * q + 4 * hh);
#pragma unroll
        for (int s = 0; s < 4; s += 2) {
            const f32x2 a2 = {acc[4 * q + s], acc[4 * q + s + 1]};
            const f32x2 b2 = {b[s], b[s + 1]};
            const f32x2 x = a2 + b2;
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                float v = x[e];
                if (RELU) {
                    const int bits = __builtin_bit_cast(int, v);
                    v = __builtin_bit_cast(float, bits > 0 ? bits : 0);
                }
                r[4 * q + s + e] = v;
            }
        }
    }
    return r;
}

// this lane's share of sum_u w[u] * act[u][point]; the other half-wave holds the rest
template <int NT>
__device__ __forceinline__ float dot_tiles(const f32x16 *tiles, const float *w, int hh) {
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + 32 * t + 8 * q + 4 * hh);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc = fmaf(wv[s], tiles[t][4 * q + s], acc);
        }
    return acc;
}

// Sinusoidal encoding written straight into accumulator-tile layout: register r = 4q+s of lane
// half hh holds channel c = 32t + 8q + 4hh + s (C real channels, zero padded to a tile).
// Channel order as Embedder.embed: x,y,z, then per octave sin(xyz*2^k), cos(xyz*2^k).
template <int C>
__device__ __forceinline__ void encode_tile(float x, float y, float z, int hh, int t, f32x16 &out) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int c = 32 * t + 8 * (r >> 2) + 4 * hh + (r & 3);
        const int m = c >= 3 ? c - 3 : 0;
        const int oct = m / 6, rem = m - 6 * oct;
        const int d = rem >= 3 ? rem - 3 : rem;
        const int dsel = c < 3 ? c : d;
        const float xv = dsel == 0 ? x : (dsel == 1 ? y : z);
        const float arg = xv * __int_as_float((127 + oct) << 23);      // x * 2^oct, exact
        float sn, cs;
        sincosf(arg, &sn, &cs);
        float val = rem < 3 ? sn : cs;
        if (c < 3) val = xv;
        if (c >= C) val = 0.f;
        out[r] = val;
    }
}

// pts = rays_o + rays_d * z (two roundings, like the reference's broadcast mul + add,
// DS_NeRF/run.py:1783); view dirs are the pre-normalised columns 8..10 of the ray row.
template <bool FROM_RAYS>
__device__ __forceinline__ void load_point(const float *__restrict__ a, const float *__restrict__ b, int64_t p,
                                           int S, float &px, float &py, float &pz, float &vx, float &vy,
                                           float &vz) {
    if constexpr (FROM_RAYS) {
        const int64_t ray = p / S;
        const float *row = a + ray * 11;
        const float zz = b[p];
        px = row[0] + row[3] * zz; py = row[1] + row[4] * zz; pz = row[2] + row[5] * zz;
        vx = row[8]; vy = row[9]; vz = row[10];
    } else {
        px = a[p * 3]; py = a[p * 3 + 1]; pz = a[p * 3 + 2];
        vx = b[p * 3]; vy = b[p * 3 + 1]; vz = b[p * 3 + 2];
    }
}

// Stash layout shared by the backward kernels: [row_tile][point_tile][32 units][32 points] fp32,
// i.e. contiguous 4 KB blocks.  A wave writes/reads its own accumulator tile with 16 dword
// accesses at compile-time offsets (128 B contiguous per half-wave); the weight-gradient GEMM
// reads whole blocks by LDS-DMA.
constexpr int TILE_FLOATS = 1024;
__device__ __forceinline__ float *stash_block(float *base, int row_tile, int64_t n_pt, int64_t pt) {
    return base + ((int64_t)row_tile * n_pt + pt) * TILE_FLOATS;
}
// Stash stores of the SPLIT-PRECISION training kernels are NON-TEMPORAL (global_store ... nt, round 5): 8 KB per point stream out
// of them.  Measured (profiles/r5_nt_stash_ab.json): the stash-writing split-precision forward 33.9 -> 27.4 ms per configs[2]
// iteration, the iteration 154 -> 144 ms -- with the SAME L2 hit rate (0.73) and the SAME HBM bytes either way: as ordinary
// write-back stores they fill the XCD's 4-MB L2 with dirty lines and the kernel's weight DMA queues behind them whenever the
// write-backs back up; as streaming stores they drain without that.  The exact-fp32 kernels (NT = false) write the same bytes
// over twice the time, measure equal to slightly slower with the hint, and keep ordinary stores.
template <bool NT>
__device__ __forceinline__ void stash_store(float *p, float v) {
#ifdef MVIP_EXPERIMENT_PLAIN_STASH                 // A/B build only: ordinary stores everywhere
    *p = v;
#else
    if constexpr (NT) __builtin_nontemporal_store(v, p);
    else *p = v;
#endif
}
template <bool NT = false>
__device__ __forceinline__ void store_tile(float *__restrict__ block, const f32x16 &tile, int j, int hh) {
    float *p = block + hh * 128 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) stash_store<NT>(p + (8 * (r >> 2) + (r & 3)) * 32, tile[r]);
}
__device__ __forceinline__ f32x16 load_tile(const float *__restrict__ block, int j, int hh) {
    const float *p = block + hh * 128 + j;
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = p[(8 * (r >> 2) + (r & 3)) * 32];
    return t;
}

__device__ __forceinline__ unsigned short *mask_slot(float *base, int mi, int64_t n_pt, int64_t pt, int lane) {
    return reinterpret_cast<unsigned short *>(base + ((int64_t)(79 + (mi >> 5)) * n_pt + pt) * TILE_FLOATS + (mi & 31) * 32) + lane;
}
__device__ __forceinline__ unsigned tile_sign_mask(const f32x16 &t) {
    unsigned m = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) m |= (t[r] > 0.f ? 1u : 0u) << r;
    return m;
}

// row-tile indices of the activation stash (forward, STASH=true) ...
constexpr int AT_H = 0;        // h_l at 8l .. 8l+7, l = 0..7
constexpr int AT_FEAT = 64;    // feature_linear output (no activation)
constexpr int AT_V = 72;       // view-branch hidden (4 tiles)
constexpr int AT_EMB = 76;     // encoded point (2 tiles, 63 real channels)
constexpr int AT_EDIR = 78;    // encoded direction (1 tile, 27 real channels)
// ReLU SIGN masks of the 64 trunk tiles (mask index = tile) and the 4 view-branch tiles (64 + t): one bit per (unit, point),
// 16 bits per lane = 128 B per tile and 32 points instead of the tile's 4 KB.  The split-precision delta kernel needs nothing
// else of these activations (relu'(h) = [h > 0]); its reads were 9.7 KB per point, half of that kernel's HBM traffic.
// Written by the split-precision stash forward only; 32 mask entries (32 floats each) per 1024-float block.
constexpr int AT_MASK = 79;
constexpr int AT_MASK_TILES = 3;          // ceil(68 / 32)
constexpr int AT_TILES = AT_MASK + AT_MASK_TILES;
// ... and of the pre-activation-gradient stash written by the delta kernel
constexpr int GT_G = 0;        // G_l at 8l .. 8l+7
constexpr int GT_F = 64;       // grad wrt feature
constexpr int GT_V = 72;       // grad wrt view-branch pre-activation (4 tiles)
constexpr int GT_D = 76;       // rows 0..3: d_raw transposed (rgb0, rgb1, rgb2, sigma)
constexpr int GT_TILES = 77;

}  // namespace mvip
