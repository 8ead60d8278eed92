// Epilogues that write MFMA OPERANDS instead of fp32 tensors: the producer of an activation hands it to the next
// contraction in that contraction's own operand format, scaled by a power of two that is known BEFORE the launch
// (a rigorous bound of the tensor's magnitude, guidance/transformer_cm.py), so the separate absolute-maximum pass and
// the fp32 -> split-plane pass between two contractions disappear (DS_NeRF/guidance/sd_utils.py:390-403: the
// unet(...) call; 11 launches per transformer block).
//
// A 32 x 32 accumulator tile of v_mfma_f32_32x32x*: register r of lane (l32, kg) holds row 8 (r >> 2) + 4 kg + (r & 3),
// column l32.  A split-plane fragment is 8 consecutive ROWS (channels) of one column (token): rows 8 q .. 8 q + 7 are
// registers 4 q .. 4 q + 3 of the two lanes (l32, 0) and (l32, 1).  One v_permlane32_swap per register pair trades the
// halves, after which the lower half-wave owns the fragments q = 0, 2 and the upper one q = 1, 3 of its column.
#pragma once
#include "common.h"

namespace mvip {

__device__ __forceinline__ void sink_split8(const float (&v)[8], uint4 &hi, uint4 &lo) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    unsigned h[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h2 a, b;
        a.x = (_Float16)v[2 * i]; a.y = (_Float16)v[2 * i + 1];
        b.x = (_Float16)(v[2 * i] - (float)a.x); b.y = (_Float16)(v[2 * i + 1] - (float)a.y);
        h[i] = __builtin_bit_cast(unsigned, a); l[i] = __builtin_bit_cast(unsigned, b);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// v[16]: one accumulator tile's finished values of this lane.  Returns the two 8-row fragments this lane owns after
// the exchange: frag[0] = rows 8 q0 .., frag[1] = rows 8 q1 .. with q0 = kg, q1 = 2 + kg.
__device__ __forceinline__ void sink_exchange(const float (&v)[16], float (&f0)[8], float (&f1)[8]) {
    // v_permlane32_swap_b32 a, b: lanes 32..63 of a <-> lanes 0..31 of b.  With a = the q-even register and b = the q-odd
    // register of a pair: afterwards a = {own q-even | lower half's q-odd}, b = {upper half's q-even | own q-odd}, i.e.
    // (a, b) are elements (e, 4 + e) of the fragment q = kg (pair 0, 1) resp. q = 2 + kg (pair 2, 3) in BOTH halves.
    // Written as inline assembly: hipcc 7.2 folds the second result of __builtin_amdgcn_permlane32_swap into the
    // first when both feed the same packed conversion (observed: fragment elements 4..7 came out as copies of 0..3;
    // tools/micro/permlane_swap.hip shows the instruction itself doing what is written here).  The leading s_nop covers
    // the VALU-write -> permlane-read wait states the compiler would otherwise insert.
    float a[8], b[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { a[e] = v[e]; b[e] = v[4 + e]; a[4 + e] = v[8 + e]; b[4 + e] = v[12 + e]; }
    asm volatile("s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %8\n\t"
                 "v_permlane32_swap_b32 %1, %9\n\t"
                 "v_permlane32_swap_b32 %2, %10\n\t"
                 "v_permlane32_swap_b32 %3, %11\n\t"
                 "v_permlane32_swap_b32 %4, %12\n\t"
                 "v_permlane32_swap_b32 %5, %13\n\t"
                 "v_permlane32_swap_b32 %6, %14\n\t"
                 "v_permlane32_swap_b32 %7, %15\n\t"
                 "s_nop 1"
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                   "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]));
#pragma unroll
    for (int e = 0; e < 4; ++e) { f0[e] = a[e]; f0[4 + e] = b[e]; f1[e] = a[4 + e]; f1[4 + e] = b[4 + e]; }
}

// Store one tile as split planes.  planes: [N][CK][kg 2][hl 2][P][16 B]; blk8 = index of the tile's first 8-row block
// inside the sample's plane set (row / 8), n_blk8 = valid 8-row blocks of the set (blocks beyond it are not written);
// p = this lane's column.  kg = lane >> 5.
__device__ __forceinline__ void sink_store_planes(char *planes_n, int64_t P, int blk8, int n_blk8, int64_t p, int kg,
                                                  const float (&v)[16], bool col_ok = true, bool write_lo = true) {
    float f0[8], f1[8];
    sink_exchange(v, f0, f1);
    uint4 hi, lo;
    const int b0 = blk8 + kg, b1 = blk8 + 2 + kg;
    if (col_ok && b0 < n_blk8) {
        sink_split8(f0, hi, lo);
        uint4 *dst = reinterpret_cast<uint4 *>(planes_n) + ((int64_t)(b0 >> 1) * 4 + (b0 & 1) * 2) * P + p;
        dst[0] = hi;
        if (write_lo) dst[P] = lo;                    // fp16 mode: the consumer fetches hi planes only
    }
    if (col_ok && b1 < n_blk8) {
        sink_split8(f1, hi, lo);
        uint4 *dst = reinterpret_cast<uint4 *>(planes_n) + ((int64_t)(b1 >> 1) * 4 + (b1 & 1) * 2) * P + p;
        dst[0] = hi;
        if (write_lo) dst[P] = lo;
    }
}

// Store one TRANSPOSED tile (lane = channel row, registers = 32 tokens) as two attention V fragments (csrc/attention.hip):
// registers 0..7 are the fragment of the tile's first 16 keys, 8..15 of its second 16, already in the kernel's key order.
// frag16: [.. s16][hl 2][lane 64][16 B], vt = the (head, dt) block's base, s16 = first 16-key group of the tile.
__device__ __forceinline__ void sink_store_vfrag(char *vt, int s16, int lane, const float (&v)[16], bool write_lo = true) {
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        float t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t[j] = v[8 * s + j];
        uint4 hi, lo;
        sink_split8(t, hi, lo);
        uint4 *dst = reinterpret_cast<uint4 *>(vt) + ((int64_t)(s16 + s) * 2) * 64 + lane;
        dst[0] = hi;
        if (write_lo) dst[64] = lo;
    }
}

}  // namespace mvip
