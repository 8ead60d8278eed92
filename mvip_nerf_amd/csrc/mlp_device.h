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

// The weight stream: section A of a packed image consumed front to back through the LDS ring.
// Protocol (g = chunk being computed): loads of chunk g+2 are issued at the first block of
// chunk g; the __syncthreads at the end of chunk g (which waits vmcnt(0)) makes chunk g+2
// visible to every wave.  Hence chunk g+1 is already complete while g is computed, which is
// what allows reading the first block of g+1 before that barrier (A-operand prefetch).
struct Stream {
    const float *packed;     // section A base (a section B may follow at SEC_A_FLOATS)
    float *lds;
    int wave, lane;

    __device__ __forceinline__ void issue_chunk(int g, int slot, int total_chunks = TOTAL_CHUNKS) const {
        if (g < total_chunks) {
            const float *src = packed + (int64_t)g * CHUNK_FLOATS + wave * (4 * BLOCK_FLOATS) + lane * 4;
            float *dst = lds + slot * CHUNK_FLOATS + wave * (4 * BLOCK_FLOATS);
#pragma unroll
            for (int b = 0; b < 4; ++b) glds16(src + b * BLOCK_FLOATS, dst + b * BLOCK_FLOATS);
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
};

__device__ __forceinline__ f32x16 mfma(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// One linear layer on the wave's 32 columns: NT output tiles x KG k-groups, A from the ring,
// B from registers through bop(kg, s).  pre(ti) runs before a tile's MFMA chain (e.g. to start
// loads the epilogue needs) and its result is handed to epi(ti, acc, pre_value), which consumes
// each finished 32x32 tile.  `g0` = chunk index of the layer's first block (a multiple of NSLOT).
template <int NT, int KG, bool LAST, class BOp, class Pre, class Epi>
__device__ __forceinline__ void run_layer(const Stream &st, int g0, f32x4 &a, BOp bop, Pre pre, Epi epi,
                                          int total_chunks = TOTAL_CHUNKS) {
    // Two accumulator tiles alternate and the epilogue of tile ti-1 is issued two blocks into tile
    // ti: its accumulator is long complete by then, so no MFMA->read wait states are needed and its
    // VALU work hides under the dependency-paced MFMA chain of the current tile.
    using PV = decltype(pre(ic<0>{}));
    f32x16 accs[2];
    PV pvs[2];
    static_for<NT>([&](auto ti) {
        constexpr int T = decltype(ti)::value;
        pvs[T & 1] = pre(ti);
        f32x16 &acc = accs[T & 1];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        static_for<KG>([&](auto kg) {
            constexpr int bi = T * KG + decltype(kg)::value;
            constexpr bool last_block = LAST && (bi == NT * KG - 1);
            if constexpr (bi % CHUNK_BLOCKS == 0)
                st.issue_chunk(g0 + bi / CHUNK_BLOCKS + 2, (bi / CHUNK_BLOCKS + 2) % NSLOT, total_chunks);
            f32x4 an = a;
            if constexpr (!last_block) an = st.template read_block<bi + 1>();
            acc = mfma(a[0], bop(kg, ic<0>{}), acc);
            acc = mfma(a[1], bop(kg, ic<1>{}), acc);
            acc = mfma(a[2], bop(kg, ic<2>{}), acc);
            acc = mfma(a[3], bop(kg, ic<3>{}), acc);
            if constexpr (bi % CHUNK_BLOCKS == CHUNK_BLOCKS - 1) __syncthreads();
            a = an;
            if constexpr (decltype(kg)::value == 1 && T > 0) epi(ic<T - 1>{}, accs[(T - 1) & 1], pvs[(T - 1) & 1]);
        });
    });
    epi(ic<NT - 1>{}, accs[(NT - 1) & 1], pvs[(NT - 1) & 1]);
}
struct NoPre { template <class T> __device__ __forceinline__ int operator()(T) const { return 0; } };

// accumulator tile (+ bias row, natural unit order) -> activation tile
template <bool RELU>
__device__ __forceinline__ f32x16 bias_relu(const f32x16 &acc, const float *bias32, int hh) {
    f32x16 r;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 b = *reinterpret_cast<const f32x4 *>(bias32 + 8 * q + 4 * hh);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float x = acc[4 * q + s] + b[s];
            r[4 * q + s] = RELU ? (x < 0.f ? 0.f : x) : x;
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
__device__ __forceinline__ void store_tile(float *__restrict__ block, const f32x16 &tile, int j, int hh) {
    float *p = block + hh * 128 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) p[(8 * (r >> 2) + (r & 3)) * 32] = tile[r];
}
__device__ __forceinline__ f32x16 load_tile(const float *__restrict__ block, int j, int hh) {
    const float *p = block + hh * 128 + j;
    f32x16 t;
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = p[(8 * (r >> 2) + (r & 3)) * 32];
    return t;
}

// row-tile indices of the activation stash (forward, STASH=true) ...
constexpr int AT_H = 0;        // h_l at 8l .. 8l+7, l = 0..7
constexpr int AT_FEAT = 64;    // feature_linear output (no activation)
constexpr int AT_V = 72;       // view-branch hidden (4 tiles)
constexpr int AT_EMB = 76;     // encoded point (2 tiles, 63 real channels)
constexpr int AT_EDIR = 78;    // encoded direction (1 tile, 27 real channels)
constexpr int AT_TILES = 79;
// ... and of the pre-activation-gradient stash written by the delta kernel
constexpr int GT_G = 0;        // G_l at 8l .. 8l+7
constexpr int GT_F = 64;       // grad wrt feature
constexpr int GT_V = 72;       // grad wrt view-branch pre-activation (4 tiles)
constexpr int GT_D = 76;       // rows 0..3: d_raw transposed (rgb0, rgb1, rgb2, sigma)
constexpr int GT_TILES = 77;

}  // namespace mvip
