// Pieces shared by the 16-points-per-wave kernels (mlp_fwd16.hip: inference and stash-writing forward;
// mlp_bwd16.hip: delta propagation): the weight ring, the layer loop on v_mfma_f32_16x16x4_f32, the epilogue.
#pragma once
#include "common.h"
#include "mlp_layout.h"
#include "mlp_device.h"
#include <type_traits>

namespace mvip {
using namespace mlp;

namespace f16p {

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int V> using ic = std::integral_constant<int, V>;
template <int N, class F, int I = 0>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) { f(ic<I>{}); static_for<N, F, I + 1>(static_cast<F &&>(f)); }
}

// channel c of the sinusoidal encoding of (x, y, z) with C real channels (Embedder.embed order)
template <int C>
__device__ __forceinline__ float enc_channel(float x, float y, float z, int c) {
    const int m = c >= 3 ? c - 3 : 0;
    const int oct = m / 6, rem = m - 6 * oct;
    const int d = rem >= 3 ? rem - 3 : rem;
    const int dsel = c < 3 ? c : d;
    const float xv = dsel == 0 ? x : (dsel == 1 ? y : z);
    const float arg = xv * __int_as_float((127 + oct) << 23);
    float val = rem < 3 ? sinf(arg) : cosf(arg);
    if (c < 3) val = xv;
    if (c >= C) val = 0.f;
    return val;
}

constexpr int NSLOT16 = 4;
constexpr int RING16_FLOATS = NSLOT16 * CHUNK_FLOATS;            // 64 KB
constexpr int LDS16_FLOATS = RING16_FLOATS + SEC_B_FLOATS;       // + 13 KB of small vectors
constexpr int WG_POINTS = 128;

// in-tiles (of 16 input units) per output tile, per layer
constexpr int NTI_L0 = 4, NTI_LH = 16, NTI_L5 = 20, NTI_LV = 18;
static_assert(16 * NTI_L0 == L0_BLOCKS && 16 * NTI_LH == LH_BLOCKS && 16 * NTI_L5 == L5_BLOCKS &&
              8 * NTI_LV == LV_BLOCKS, "same block counts as the 32-point image");

// 16-byte reads through a generic pointer or through an explicit LDS (address space 3) pointer: a pointer that has passed
// through an `asm` register constraint has lost its address space, and a generic f32x4 load of it is a flat_load, not a ds_read
typedef __attribute__((address_space(3))) const float lds_cfloat;
__device__ __forceinline__ f32x4 ld4(const float *p) { return *reinterpret_cast<const f32x4 *>(p); }
__device__ __forceinline__ f32x4 ld4(lds_cfloat *p) { return *(const __attribute__((address_space(3))) f32x4 *)p; }

template <int OFF>
__device__ __forceinline__ void glds(const float *src_lane, float *dst_wave) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src_lane,
                                     (__attribute__((address_space(3))) void *)dst_wave, 16, OFF, 0);
}

// NSLOT = ring slots of 16 KB.  WRAP (the persistent kernel, mlp_fwd16.hip): chunk indices past the end of the image wrap to
// the start, i.e. the NEXT tile's first chunks are staged while this tile's last layers run; TOTAL_CHUNKS % NSLOT == 0 keeps the
// slot of a chunk a compile-time function of its index across tiles.
template <int NSLOT = NSLOT16, bool WRAP = false>
struct Stream16T {
    const float *packed;
    float *lds;
    int wave, lane;
    int total_chunks = TOTAL_CHUNKS;     // chunks in the image being streamed (the transposed image has more)
    lds_cfloat *lds_read = nullptr;      // WRAP: operand reads go through this LDS pointer (the persistent kernel makes it opaque per tile)
    static constexpr int nslot = NSLOT;
    static constexpr bool wrap = WRAP;
    // 16 KB chunk g into ring slot `slot`: 2 KB per wave, one address set-up for both pieces
    __device__ __forceinline__ void issue_chunk(int g, int slot) const {
        if (WRAP && g >= total_chunks) g -= total_chunks;
        if (g < total_chunks) {
            const float *src = packed + (int64_t)g * CHUNK_FLOATS + wave * 512 + lane * 4;
            float *dst = lds + slot * CHUNK_FLOATS + wave * 512;
            glds<0>(src, dst);
            glds<1024>(src, dst);
        }
    }
    template <int BI>
    __device__ __forceinline__ f32x4 read_block() const {
        constexpr int off = ((BI / CHUNK_BLOCKS) % NSLOT) * CHUNK_FLOATS + (BI % CHUNK_BLOCKS) * BLOCK_FLOATS;
        if constexpr (WRAP) return ld4(lds_read + off + lane * 4);
        else return ld4(lds + off + lane * 4);
    }
};
using Stream16 = Stream16T<>;

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// One layer: NTO output tiles x NTI input tiles; BASE = absolute index of the layer's first block.
// bsrc(ti) -> the activation tile (f32x4) feeding input units 16 ti .. 16 ti + 15; pre(to) runs at the top of output tile
// `to` (e.g. starts the loads its epilogue needs) and its value reaches epi(to, acc, pre_value), which consumes the tile.
// bias == nullptr (compile-time HASB = false): the accumulators start at zero.
struct NoPre16 { template <class T> __device__ __forceinline__ int operator()(T) const { return 0; } };
template <int BASE, int NTO, int NTI, bool LAST, bool HASB, class St, class Bias, class BSrc, class Pre, class Epi>
__device__ __forceinline__ void layer16x(const St &st, f32x4 &a, Bias bias, BSrc bsrc, Pre pre, Epi epi) {
    // Two accumulator tiles alternate; the epilogue of tile t-1 is issued one block into tile t, where its VALU
    // instructions run under MFMAs instead of after a drained chain.
    using PV = decltype(pre(ic<0>{}));
    f32x4 accs[2];
    PV pvs[2];
    static_for<NTO>([&](auto to_) {
        constexpr int TO = decltype(to_)::value;
        f32x4 &acc = accs[TO & 1];
        pvs[TO & 1] = pre(to_);
        // the layer's bias enters as the C operand of the tile's first MFMA (one LDS read per tile instead of a read
        // plus four adds in the epilogue: 202.9 -> 201.0 ms on the bench launch)
        if constexpr (HASB) acc = ld4(bias + 16 * TO + 4 * (st.lane >> 4));
        else acc = f32x4{0.f, 0.f, 0.f, 0.f};
        static_for<NTI>([&](auto ti_) {
            constexpr int TI = decltype(ti_)::value;
            constexpr int bi = BASE + TO * NTI + TI;
            constexpr bool last_block = LAST && (TO == NTO - 1) && (TI == NTI - 1);
            if constexpr (bi % CHUNK_BLOCKS == 0) st.issue_chunk(bi / CHUNK_BLOCKS + 2, (bi / CHUNK_BLOCKS + 2) % St::nslot);
            // The next block's operands are read right AFTER this block's first MFMA and pinned there.  Left alone
            // the scheduler hoists whole chunks of operand reads to the top and spills them; and read BEFORE the
            // first MFMA, the compiler's wait for THIS block's operands (it emits lgkmcnt(0), not lgkmcnt(1))
            // would also wait out the read just issued -- a full LDS latency exposed every other block, in both
            // SIMD partners at once.
            const f32x4 b = bsrc(ti_);
            acc = mfma4(a[0], b[0], acc);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 an = a;
            if constexpr (!last_block) an = st.template read_block<bi + 1>();
            else if constexpr (St::wrap) an = st.template read_block<0>();     // the next tile's first block (staged two chunks ago)
            __builtin_amdgcn_sched_barrier(0);
            acc = mfma4(a[1], b[1], acc);
            acc = mfma4(a[2], b[2], acc);
            acc = mfma4(a[3], b[3], acc);
            // keep the tile's chain inside the tile: in the view layer the optimiser otherwise sinks all MFMAs below
            // all operand reads (reads first, 4 KB of them spilled, then the MFMAs fed from scratch)
            if constexpr (TI == NTI - 1) asm volatile("" : "+v"(acc));
#ifndef MVIP_EXPERIMENT_NO_BARRIER          // timing experiment only: results are wrong without the barrier
            if constexpr (bi % CHUNK_BLOCKS == CHUNK_BLOCKS - 1) __syncthreads();
#endif
            a = an;
            if constexpr (TO > 0 && TI == 1) epi(ic<TO - 1>{}, accs[(TO - 1) & 1], pvs[(TO - 1) & 1]);
        });
    });
    epi(ic<NTO - 1>{}, accs[(NTO - 1) & 1], pvs[(NTO - 1) & 1]);
}
template <int BASE, int NTO, int NTI, bool LAST, class St, class Bias, class BSrc, class Epi>
__device__ __forceinline__ void layer16(const St &st, f32x4 &a, Bias bias, BSrc bsrc, Epi epi) {
    layer16x<BASE, NTO, NTI, LAST, true>(st, a, bias, bsrc, NoPre16{}, [&](auto to, const f32x4 &acc, int) { epi(to, acc); });
}

// epilogue of one output tile: the bias already entered through the MFMA C operand (layer16), what is left is the
// NaN-preserving ReLU -- one integer max per value (see mlp_device.h)
template <bool RELU>
__device__ __forceinline__ f32x4 act16(const f32x4 &acc) {
#ifdef MVIP_EXPERIMENT_NO_EPILOGUE         // timing experiment only
    return acc;
#endif
    f32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = acc[i];
        if (RELU) {
            const int bits = __builtin_bit_cast(int, v);
            v = __builtin_bit_cast(float, bits > 0 ? bits : 0);
        }
        r[i] = v;
    }
    return r;
}

}  // namespace f16p
}  // namespace mvip
