// Packed weight image of the 8x256 NeRF MLP (DS_NeRF/run_nerf_helpers.py:86-127) as the fused
// kernels consume it.
//
// Section A -- MFMA operand blocks, in consumption order.  A "block" is the A operand of four
// consecutive v_mfma_f32_32x32x2_f32 steps for one 32-row output tile: 64 lanes x 4 floats =
// 256 floats (1 KB).  Lane (i = lane&31, h = lane>>5) of block (ti, kg) holds
//        W[32*ti + i][8*kg + 4*h + s],  s = 0..3          (row = output unit, col = input unit)
// stored as float index h*128 + i*4 + s, so the block is read with one conflict-free
// ds_read_b128 per lane and staged global->LDS by one global_load_lds_dwordx4 per wave.
// Why this k order: a 32x32 accumulator tile holds, in register r = 4q+s of lane (j, h),
// row 8q+4h+s of the tile (column j).  Using accumulator register r directly as the B operand
// of a K=2 MFMA step therefore contributes input units {32t+8q+s (h=0), 32t+8q+s+4 (h=1)};
// the block layout above is exactly the matching A operand.  Activations never leave registers.
//
// Order of the blocks of one layer in the stream: output tiles are consumed in PAIRS -- blocks of tiles
// (2P, 2P+1) alternate, k-group by k-group -- so that a wave runs two independent accumulator chains, reads
// both A operands of a step with one pair of LDS reads and has eight MFMAs (512 cycles) between barriers,
// DMA issues and operand waits.  (Measured effect on the forward: 0.6 %; tools/micro/mfma_rate.hip shows the
// matrix pipe itself sustains 64.0 cycles per fp32 32x32x2 MFMA for dependent and independent chains alike.)
//
// Section B -- small vectors (biases, the 256->1 sigma row, the 128->3 rgb rows), natural order.
#pragma once

namespace mvip { namespace mlp {

constexpr int BLOCK_FLOATS = 256;

// position of block (tile ti, k-group kg) inside a layer with KG k-groups per tile, and its inverse
__host__ __device__ constexpr int block_pos(int ti, int kg, int KG) { return (ti >> 1) * (2 * KG) + 2 * kg + (ti & 1); }
__host__ __device__ inline void block_tile(int local, int KG, int &ti, int &kg) {
    const int pair = local / (2 * KG), rem = local % (2 * KG);
    kg = rem >> 1;
    ti = 2 * pair + (rem & 1);
}
constexpr int CHUNK_BLOCKS = 16;                         // 16 KB staged per barrier
constexpr int CHUNK_FLOATS = CHUNK_BLOCKS * BLOCK_FLOATS;

// layer table: N tiles (of 32 outputs), K groups (of 8 inputs)
constexpr int L0_NT = 8, L0_KG = 8;                      // 63(+1) -> 256
constexpr int LH_NT = 8, LH_KG = 32;                     // 256 -> 256
constexpr int L5_NT = 8, L5_KG = 40;                     // 64 + 256 -> 256 (skip: encoded input first)
constexpr int LV_NT = 4, LV_KG = 36;                     // 256 + 27(+5) -> 128

constexpr int L0_BLOCKS = L0_NT * L0_KG;                 // 64
constexpr int LH_BLOCKS = LH_NT * LH_KG;                 // 256
constexpr int L5_BLOCKS = L5_NT * L5_KG;                 // 320
constexpr int LV_BLOCKS = LV_NT * LV_KG;                 // 144

// block offsets of the layers in the stream
constexpr int OFF_L0 = 0;
constexpr int OFF_L1 = OFF_L0 + L0_BLOCKS;               // layers 1..4
constexpr int OFF_L5 = OFF_L1 + 4 * LH_BLOCKS;
constexpr int OFF_L6 = OFF_L5 + L5_BLOCKS;               // layers 6, 7
constexpr int OFF_FEAT = OFF_L6 + 2 * LH_BLOCKS;
constexpr int OFF_VIEWS = OFF_FEAT + LH_BLOCKS;
constexpr int TOTAL_BLOCKS = OFF_VIEWS + LV_BLOCKS;      // 2320
constexpr int TOTAL_CHUNKS = TOTAL_BLOCKS / CHUNK_BLOCKS;  // 145
static_assert(TOTAL_BLOCKS % CHUNK_BLOCKS == 0, "stream must be whole chunks");
static_assert(L0_BLOCKS % (4 * CHUNK_BLOCKS) == 0 && LH_BLOCKS % (4 * CHUNK_BLOCKS) == 0 &&
              L5_BLOCKS % (4 * CHUNK_BLOCKS) == 0, "layers must keep the 4-slot ring phase");

constexpr int SEC_A_FLOATS = TOTAL_BLOCKS * BLOCK_FLOATS;   // 593,920

// Section B offsets (floats, relative to the start of section B)
constexpr int SB_BIAS = 0;            // 8 x 256, layer l at l*256
constexpr int SB_BFEAT = 2048;        // 256
constexpr int SB_BVIEWS = 2304;       // 128
constexpr int SB_WALPHA = 2432;       // 256
constexpr int SB_BALPHA = 2688;       // 1 (+3 pad)
constexpr int SB_WRGB = 2692;         // 3 x 128
constexpr int SB_BRGB = 3076;         // 3 (+1 pad)
constexpr int SEC_B_FLOATS = 13 * BLOCK_FLOATS;             // 3328 (13 KB, whole 1-KB pieces)
static_assert(SB_BRGB + 4 <= SEC_B_FLOATS, "section B overflow");

constexpr int PACKED_FLOATS = SEC_A_FLOATS + SEC_B_FLOATS;  // 597,248

// state-dict order of the 24 parameter tensors
enum Param {
    P_W0 = 0, P_B0 = 1,            // pts_linears.i.weight / bias at 2i, 2i+1
    P_WV = 16, P_BV = 17,          // views_linears.0
    P_WF = 18, P_BF = 19,          // feature_linear
    P_WA = 20, P_BA = 21,          // alpha_linear
    P_WR = 22, P_BR = 23,          // rgb_linear
    P_COUNT = 24
};

}}  // namespace mvip::mlp
