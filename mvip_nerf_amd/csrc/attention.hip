// Multi-head attention of the SD UNet's transformer blocks (the unet(...) call of the SDS step,
// DS_NeRF/guidance/sd_utils.py:390-403 / :240; block structure from the published SD-1.5 architecture:
// self-attention over 4096 / 1024 / 256 / 64 image tokens with 8 heads of 40 / 80 / 160 channels, and
// cross-attention onto the 77 prompt tokens) as ONE flash-style kernel on the fp16 matrix cores in split
// precision ("f16x3": both operands of both products split into fp16 hi + lo, the three leading products
// accumulated in fp32 -- fp32-grade results, ~1e-6 relative).  Nothing of size [Lq, Lk] ever touches HBM.
//
// Orientation (keys on the MFMA rows, queries on the lanes):
//   S^T[key][query] = sum_c K[key][c] Q[c][query]          A = K fragment, B = Q fragment (contraction over channels)
//   softmax over keys = over the 16 accumulator registers of a lane + ONE cross-half exchange; the running
//   maximum / sum of a query live in the lane that owns the query, so rescaling O is a per-lane multiply;
//   O^T[d][query]  = sum_key V[d][key] P^T[key][query]      A = V fragment, B = P^T straight from the accumulators
//   (an accumulator tile is a legal B operand of the next MFMA when that MFMA contracts over the tile's ROW
//   index; the k order inside a 16-key step is then key = 16 s + 8 (j >> 2) + 4 h + (j & 3) for element j of lane
//   half h, and the V fragments are packed in exactly that order by attn_pack_v_kernel).
//
// Operand formats (all produced by small packers so every LDS fill is a 1-KiB DMA piece and every fragment
// one ds_read_b128):
//   Q, K : "split planes" [N][chunks][kg 2][hl 2][tokens][8 halves] (the format of csrc/conv3x3.hip): plane
//          (kg, hl) of 16-channel chunk ck holds channels ck*16 + kg*8 + 0..7 of every token as the hi / lo
//          fp16 term; head h owns chunks h*NCH .. h*NCH + NCH - 1 (head dim padded to a multiple of 16 with
//          zero channels: 40 -> 48).  Values are pre-scaled by a power of two (scale2 = {s, 1/s}).
//   V    : A fragments [N][heads][DT][LkP/16][hl 2][lane 64][8 halves], rows = head channels padded to DT*32,
//          k = keys in the permuted order above, zero beyond Lk.
//   out  : fp32 channel-major [N][heads*D][LqP] (what the output projection's split-plane writer reads).
//
// Workgroup = 4 (or 8) waves = 128 (256) queries of one (sample, head); each wave owns 32 queries for the whole key loop.
// K / V tiles of KT keys are staged through LDS by DMA (global_load_lds), double-buffered, one barrier per
// tile; 124 registers and 28 KB of LDS for the 40-channel heads (32-key tiles) so that four waves share a SIMD and
// one wave's softmax (VALU) runs under the others' MFMAs.
#include "common.h"
#include "plane_sink.h"

namespace mvip {
namespace attn {

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void glds16b(const void *src_lane, void *dst_wave) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src_lane,
                                     (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
}

struct AttnArgs {
    const char *qs, *ks, *vp;
    const float *q_scale2, *k_scale2, *v_scale2;
    float *o;
    int N, heads, Lq, LqP, Lk, LkP, D, qblocks;
    float softmax_scale;
    // operand strides: tokens per Q plane / per K plane, 16-key groups per (head, row tile) block of V.  The packers
    // of this file write compact operands (Lq, LkP, LkP / 16); a GEMM epilogue (mvip_gemm_f16x3_sinks) writes them at
    // its own column count P.
    int q_stride, k_stride, v_groups;
    // op != nullptr: the result leaves as the output projection's operand planes [N][heads*D/16][2][2][LqP][8 halves],
    // still multiplied by V's scale (|softmax(..) V| <= |V|max, so V's power of two fits) -- no fp32 tensor, no
    // absolute-maximum pass, no split pass between attention and its output projection
    char *op;
};

constexpr float P_SHIFT = 10.0f;          // probabilities travel as p * 2^10 so that their fp16 lo terms stay normal

// NW waves per workgroup = NW * 32 queries sharing every K / V tile: the tiles arrive by LDS-DMA, whose rate per CU
// (11-13 B/clk, MI355X_MICROARCH.md) is what bounds this kernel at 128 queries per workgroup (14 KB of operands per
// 21 MFMAs per wave = 21 B/clk/CU at full matrix rate); 256 queries halve the bytes per FLOP.
// F16 (the reference's --fp16 mode): ONE fp16 product per step in both contractions -- the hi halves of Q, K, V and of the
// probabilities only; the lo halves of the operand images are neither fetched nor expected to be written.
template <int NCH, int DT, int KT, int NW = 4, bool F16 = false>
__global__ void __launch_bounds__(NW * 64, (NW == 8 ? 2 : (NCH <= 3 ? (KT == 32 ? 4 : 2) : (NCH <= 5 ? 2 : 1))))
attn_f16x3_kernel(const AttnArgs a) {
    constexpr int KB = NCH * 4 * KT * 16;                 // K tile bytes: [NCH][kg][hl][KT][16 B]
    constexpr int VB = DT * (KT / 16) * 2 * 1024;         // V tile bytes: [DT][KT/16][hl][64][16 B]
    constexpr int NKP = NCH * 4 * KT / 64;                // 1-KiB DMA pieces of a K tile
    constexpr int NVP = DT * (KT / 16) * 2;               // ... of a V tile
    __shared__ __attribute__((aligned(16))) char lds[2 * (KB + VB)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l32 = lane & 31, h = lane >> 5;

    // XCD-aware order: the query blocks of one (sample, head) run on one XCD, whose L2 then serves K and V
    int id = blockIdx.x;
    const int total = gridDim.x;
    if ((total & 7) == 0) id = (id & 7) * (total >> 3) + (id >> 3);
    const int qb = id % a.qblocks;
    const int nh = id / a.qblocks;
    const int head = nh % a.heads, n = nh / a.heads;
    const int QC = a.heads * NCH;

    // ---- Q fragments of this wave's 32 queries (B operands), kept in registers for the whole key loop ----
    int q = qb * (NW * 32) + wave * 32 + l32;
    const bool q_ok = q < a.Lq;
    if (!q_ok) q = a.Lq - 1;
    h16x8 qh[NCH], ql[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int64_t plane = ((int64_t)(n * QC + head * NCH + c) * 2 + h) * 2;
        qh[c] = *reinterpret_cast<const h16x8 *>(a.qs + ((plane + 0) * a.q_stride + q) * 16);
        if constexpr (!F16) ql[c] = *reinterpret_cast<const h16x8 *>(a.qs + ((plane + 1) * a.q_stride + q) * 16);
    }

    const char *k_base = a.ks + (int64_t)(n * QC + head * NCH) * 4 * a.k_stride * 16;
    const char *v_base = a.vp + (int64_t)(n * a.heads + head) * DT * a.v_groups * 2048;
    auto issue_tile = [&](int t, int buf) {
        char *kd = lds + buf * (KB + VB), *vd = kd + KB;
        const int k0 = t * KT;
#pragma unroll
        for (int p0 = 0; p0 < NKP; p0 += NW) {
            const int p = p0 + wave;
            if (p < NKP) {
                if (KT == 64) {            // piece = (chunk, kg, hl): 64 keys x 16 B, contiguous in the plane
                    if (!F16 || !(p & 1)) glds16b(k_base + ((int64_t)p * a.k_stride + k0 + lane) * 16, kd + p * 1024);
                } else {                   // KT == 32: piece = (chunk, kg), lanes 0..31 -> hi plane, 32..63 -> lo plane
                    if (!F16 || h == 0) glds16b(k_base + ((int64_t)(p * 2 + h) * a.k_stride + k0 + l32) * 16, kd + p * 1024);
                }
            }
        }
#pragma unroll
        for (int p0 = 0; p0 < NVP; p0 += NW) {
            const int p = p0 + wave;
            if (p < NVP) {
                const int dt = p / ((KT / 16) * 2), r = p % ((KT / 16) * 2);
                if (!F16 || !(r & 1)) glds16b(v_base + ((int64_t)dt * a.v_groups * 2 + (k0 / 16) * 2 + r) * 1024 + lane * 16, vd + p * 1024);
            }
        }
    };

    f32x16 o[DT];
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;            // running max (in raw score units) and sum of this lane's 16-key share
    // scores arrive as (sq sk) q.k ; exponent = log2(e)/sqrt(D) * q.k
    const float cexp = a.softmax_scale * 1.4426950408889634f * a.q_scale2[1] * a.k_scale2[1];

    const int ntiles = a.LkP / KT;
    issue_tile(0, 0);
    for (int t = 0; t < ntiles; ++t) {
        __builtin_amdgcn_s_waitcnt(0x0f70);            // vmcnt(0): this wave's pieces of tile t have landed
        __syncthreads();                               // everyone's have; everyone is done with tile t-1
        if (t + 1 < ntiles) issue_tile(t + 1, (t + 1) & 1);
        const char *kb = lds + (t & 1) * (KB + VB), *vb = kb + KB;
#pragma unroll
        for (int sub = 0; sub < KT / 32; ++sub) {
            // ---- S^T = K Q over the head's channel chunks ----
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const char *kp = kb + (((c * 2 + h) * 2) * KT + sub * 32 + l32) * 16;
                const h16x8 kh = *reinterpret_cast<const h16x8 *>(kp);
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh[c], s, 0, 0, 0);
                if constexpr (!F16) {
                    const h16x8 kl = *reinterpret_cast<const h16x8 *>(kp + KT * 16);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql[c], s, 0, 0, 0);
                    s = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh[c], s, 0, 0, 0);
                }
            }
            // ---- keys beyond Lk (prompt padding) never contribute ----
            const int key0 = t * KT + sub * 32 + 4 * h;
            if (key0 + 27 >= a.Lk) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key0 + (r & 3) + 8 * (r >> 2) >= a.Lk) s[r] = -INFINITY;
            }
            // ---- online softmax for the lane's query ----
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * cexp);          // 0 on the first tile (m_run = -inf)
            const float off = P_SHIFT - m_new * cexp;
            m_run = m_new;
            float psum = 0.f;
            unsigned ph[8], pl[8];
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                const float p0 = __builtin_amdgcn_exp2f(fmaf(s[r], cexp, off)), p1 = __builtin_amdgcn_exp2f(fmaf(s[r + 1], cexp, off));
                psum += p0 + p1;
                const h16x2 hi = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(p0, p1));
                const h16x2 lo = __builtin_bit_cast(h16x2, __builtin_amdgcn_cvt_pkrtz(p0 - (float)hi.x, p1 - (float)hi.y));
                ph[r >> 1] = __builtin_bit_cast(unsigned, hi);
                pl[r >> 1] = __builtin_bit_cast(unsigned, lo);
            }
            l_run = l_run * alpha + psum;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.0f) != 0) {
#pragma unroll
                for (int d = 0; d < DT; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            }
            // ---- O^T += V P^T : registers 8 s2 .. 8 s2 + 7 of the tile are the B fragment of 16-key step s2 ----
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const h16x8 bh = __builtin_bit_cast(h16x8, make_uint4(ph[4 * s2], ph[4 * s2 + 1], ph[4 * s2 + 2], ph[4 * s2 + 3]));
                const h16x8 bl = __builtin_bit_cast(h16x8, make_uint4(pl[4 * s2], pl[4 * s2 + 1], pl[4 * s2 + 2], pl[4 * s2 + 3]));
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const char *vq = vb + ((d * (KT / 16) + sub * 2 + s2) * 2) * 1024 + lane * 16;
                    const h16x8 vh = *reinterpret_cast<const h16x8 *>(vq);
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, bh, o[d], 0, 0, 0);
                    if constexpr (!F16) {
                        const h16x8 vl = *reinterpret_cast<const h16x8 *>(vq + 1024);
                        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, bl, o[d], 0, 0, 0);
                        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, bh, o[d], 0, 0, 0);
                    }
                }
            }
        }
    }

    // ---- epilogue: O / (l s_v), fp32 channel-major ----
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (a.op) {
        // operand planes for the output projection: O / l, still carrying V's power-of-two scale
        const float il = 1.0f / l_tot;
        const int n8 = a.heads * a.D / 8;                     // 8-channel blocks of a sample's plane set
        char *pn = a.op + (int64_t)n * (n8 / 2) * 4 * a.LqP * 16;
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = o[d][r] * il;
            // blocks of this head beyond its D channels (head dim padded to DT * 32 rows) are not written
            const int head_blk = head * (a.D / 8), last = head_blk + a.D / 8;
            sink_store_planes(pn, a.LqP, head_blk + d * 4, last < n8 ? last : n8, q, h, v, q_ok, !F16);
        }
        return;
    }
    const float inv = a.v_scale2[1] / l_tot;
    if (q_ok) {
        float *op = a.o + ((int64_t)n * a.heads * a.D + (int64_t)head * a.D) * a.LqP + q;
#pragma unroll
        for (int d = 0; d < DT; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = d * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < a.D) op[(int64_t)row * a.LqP] = o[d][r] * inv;
            }
    }
}

// V [rows][keys] (any strides) -> A fragments in the accumulator-row key order, scaled, hi / lo.
__global__ void __launch_bounds__(256)
attn_pack_v_kernel(const float *__restrict__ v, int heads, int DP, int Lk, int LkP, int DT, int64_t sn, int64_t sr,
                   int64_t sk, const float *__restrict__ scale2, uint4 *__restrict__ out, int64_t total, int prec) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    int64_t r = idx;
    const int lane = (int)(r % 64); r /= 64;
    const int s16 = (int)(r % (LkP / 16)); r /= (LkP / 16);
    const int dt = (int)(r % DT); r /= DT;
    const int head = (int)(r % heads);
    const int64_t n = r / heads;
    const int d = dt * 32 + (lane & 31), hh = lane >> 5;
    const float s = scale2[0];
    unsigned hi[4], lo[4];
#pragma unroll
    for (int jp = 0; jp < 4; ++jp) {
        float x[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int j = 2 * jp + e;
            const int key = 16 * s16 + 8 * (j >> 2) + 4 * hh + (j & 3);
            x[e] = (d < DP && key < Lk) ? v[n * sn + (int64_t)(head * DP + d) * sr + (int64_t)key * sk] * s : 0.f;
        }
        h16x2 a, b;
        a.x = (_Float16)x[0]; a.y = (_Float16)x[1];
        b.x = (_Float16)(x[0] - (float)a.x); b.y = (_Float16)(x[1] - (float)a.y);
        hi[jp] = __builtin_bit_cast(unsigned, a); lo[jp] = __builtin_bit_cast(unsigned, b);
    }
    // [n][head][dt][s16][hl][lane]
    const int64_t base = ((((n * heads + head) * DT + dt) * (LkP / 16) + s16) * 2) * 64 + lane;
    out[base] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    if (prec != 1) out[base + 64] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}

// ---- absolute maxima of several equally long sections at once -> one power-of-two scale per section ----
// x is [outer][sections][len]; section s collects over every outer index.
__global__ void __launch_bounds__(256)
absmax_sections_kernel(const float *__restrict__ x, int64_t outer, int sections, int64_t len, unsigned *__restrict__ bits,
                       float *__restrict__ scale2) {
    const int s = blockIdx.y;
    float m = 0.f;
    auto take = [&](float v) { v = fabsf(v); m = (v == v && v < 3.0e38f) ? fmaxf(m, v) : m; };
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n4 = len >> 2;
    for (int64_t oo = 0; oo < outer; ++oo) {
        const float *xs = x + (oo * sections + s) * len;
        if ((reinterpret_cast<uintptr_t>(xs) & 15) == 0) {
            const float4 *x4 = reinterpret_cast<const float4 *>(xs);
            for (int64_t i = tid; i < n4; i += stride) { const float4 v = x4[i]; take(v.x); take(v.y); take(v.z); take(v.w); }
            for (int64_t i = (n4 << 2) + tid; i < len; i += stride) take(xs[i]);
        } else {
            for (int64_t i = tid; i < len; i += stride) take(xs[i]);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(bits + s, __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
        // the last workgroup of the launch (ticket in bits[63]) turns every section's maximum into its scale and leaves
        // the scratch words zero for the next user: no separate scale launch
        __threadfence();
        const unsigned total = gridDim.x * gridDim.y;
        if (atomicAdd(bits + 63, 1u) == total - 1) {
            atomicExch(bits + 63, 0u);
            for (int q = 0; q < sections; ++q) {
                const float m2 = __uint_as_float(atomicExch(bits + q, 0u));
                float sc = 1.f;
                if (m2 > 0.f && m2 < 3.0e38f) {
                    int e;
                    frexpf(m2, &e);
                    int k = 10 - e;
                    k = k > 60 ? 60 : (k < -60 ? -60 : k);
                    sc = ldexpf(1.f, k);
                }
                scale2[4 * q] = sc;
                scale2[4 * q + 1] = 1.f / sc;
            }
        }
    }
}

// scale2[s] = {2^k, 2^-k, -, -} with |x|max 2^k in [2^9, 2^10)  (same rule as csrc/conv3x3.hip); the maxima were
// collected in bits[s], which is left ZERO for the next user (caller-owned scratch that travels zero between calls)
__global__ void scale_sections_kernel(float *__restrict__ scale2, unsigned *__restrict__ bits, int sections) {
    const int s = threadIdx.x;
    if (s >= sections) return;
    const float m = __uint_as_float(bits[s]);
    bits[s] = 0u;
    float sc = 1.f;
    if (m > 0.f && m < 3.0e38f) {
        int e;
        frexpf(m, &e);
        int k = 10 - e;
        k = k > 60 ? 60 : (k < -60 ? -60 : k);
        sc = ldexpf(1.f, k);
    }
    scale2[4 * s] = sc;
    scale2[4 * s + 1] = 1.f / sc;
}

}  // namespace attn
}  // namespace mvip

using namespace mvip;
using namespace mvip::attn;

extern "C" int mvip_attention_supported(int64_t D) { return D == 40 || D == 80 || D == 160; }

extern "C" int64_t mvip_attention_v_bytes(int64_t N, int64_t heads, int64_t D, int64_t LkP) {
    if (N <= 0 || heads <= 0 || D <= 0 || LkP <= 0 || LkP % 32 != 0) return 0;
    const int64_t DT = (D + 31) / 32;
    return N * heads * DT * (LkP / 16) * 2048;
}

extern "C" int mvip_attention_pack_v(const float *v, int64_t N, int64_t heads, int64_t D, int64_t DP, int64_t Lk,
                                     int64_t LkP, int64_t sn, int64_t sr, int64_t sk, const float *scale2, void *vp,
                                     int prec, void *stream) {
    if ((prec < 0 || prec > 2) || N < 0 || heads <= 0 || D <= 0 || DP < D || Lk <= 0 || LkP < Lk || LkP % 32 != 0) return MVIP_EINVAL;
    if (N == 0) return MVIP_OK;
    if (!v || !vp || !scale2) return MVIP_EINVAL;
    const int DT = (int)((D + 31) / 32);
    const int64_t total = N * heads * DT * (LkP / 16) * 64;
    hipLaunchKernelGGL(attn_pack_v_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), v,
                       (int)heads, (int)DP, (int)Lk, (int)LkP, DT, sn, sr, sk, scale2, (uint4 *)vp, total, prec);
    return check_launch();
}

extern "C" int mvip_absmax_scale_sections(const float *x, int64_t outer, int64_t sections, int64_t len, float *scale2,
                                          void *zero_words64, void *stream) {
    if (outer < 0 || sections <= 0 || sections > 63 || len < 0 || !scale2 || !zero_words64) return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
    if (outer > 0 && len > 0) {
        if (!x) return MVIP_EINVAL;
        int64_t b = (len + 256 * 32 - 1) / (256 * 32);
        b = b < 1 ? 1 : (b > 256 ? 256 : b);
        hipLaunchKernelGGL(absmax_sections_kernel, dim3((unsigned)b, (unsigned)sections), dim3(256), 0, st, x, outer,
                           (int)sections, len, (unsigned *)zero_words64, scale2);
    } else {
        hipLaunchKernelGGL(scale_sections_kernel, dim3(1), dim3(64), 0, st, scale2, (unsigned *)zero_words64, (int)sections);
    }
    return check_launch();
}

static int attention_launch(const void *qs, const void *ks, const void *vp, const float *q_scale2, const float *k_scale2,
                            const float *v_scale2, int64_t N, int64_t heads, int64_t D, int64_t Lq, int64_t LqP, int64_t Lk,
                            int64_t LkP, int64_t q_stride, int64_t k_stride, int64_t v_groups, float softmax_scale, int flags,
                            float *out, void *out_planes, int prec, void *stream) {
    if ((prec < 0 || prec > 2) || N < 0 || heads <= 0 || Lq <= 0 || LqP < Lq || Lk <= 0 || LkP < Lk || LkP % 64 != 0 || Lq % 32 != 0 ||
        q_stride < Lq || k_stride < LkP || v_groups < LkP / 16)
        return MVIP_EINVAL;
    if (!mvip_attention_supported(D)) return MVIP_EUNSUP;
    if (N == 0) return MVIP_OK;
    if (!qs || !ks || !vp || !q_scale2 || !k_scale2 || !v_scale2 || (!out && !out_planes)) return MVIP_EINVAL;
    if (out_planes && (heads * D) % 16 != 0) return MVIP_EINVAL;
    AttnArgs a;
    a.qs = (const char *)qs; a.ks = (const char *)ks; a.vp = (const char *)vp;
    a.q_scale2 = q_scale2; a.k_scale2 = k_scale2; a.v_scale2 = v_scale2; a.o = out; a.op = (char *)out_planes;
    a.N = (int)N; a.heads = (int)heads; a.Lq = (int)Lq; a.LqP = (int)LqP; a.Lk = (int)Lk; a.LkP = (int)LkP; a.D = (int)D;
    a.q_stride = (int)q_stride; a.k_stride = (int)k_stride; a.v_groups = (int)v_groups;
    // 256 queries per workgroup (8 waves) once that still gives every CU a workgroup; flags bit 1 forces 128
    const bool wide = D == 40 && !(flags & 2) && N * heads * ((Lq + 255) / 256) >= 256;
    a.qblocks = (int)(wide ? (Lq + 255) / 256 : (Lq + 127) / 128);
    a.softmax_scale = softmax_scale;
    const int64_t blocks = N * heads * a.qblocks;
    if (blocks > 0x7fffffffLL) return MVIP_EINVAL;
    hipStream_t st = as_stream(stream);
#define MVIP_ATTN(F16_)                                                                                                       \
    do {                                                                                                                     \
        if (wide) hipLaunchKernelGGL((attn_f16x3_kernel<3, 2, 32, 8, F16_>), dim3((unsigned)blocks), dim3(512), 0, st, a);           \
        else if (D == 40 && (flags & 1)) /* tuning switch: 64-key tiles (two workgroups per CU) instead of 32-key tiles (four) */ \
            hipLaunchKernelGGL((attn_f16x3_kernel<3, 2, 64, 4, F16_>), dim3((unsigned)blocks), dim3(256), 0, st, a);                 \
        else if (D == 40) hipLaunchKernelGGL((attn_f16x3_kernel<3, 2, 32, 4, F16_>), dim3((unsigned)blocks), dim3(256), 0, st, a);   \
        else if (D == 80) hipLaunchKernelGGL((attn_f16x3_kernel<5, 3, 32, 4, F16_>), dim3((unsigned)blocks), dim3(256), 0, st, a);   \
        else hipLaunchKernelGGL((attn_f16x3_kernel<10, 5, 32, 4, F16_>), dim3((unsigned)blocks), dim3(256), 0, st, a);               \
    } while (0)
    if (prec == 1) MVIP_ATTN(true); else MVIP_ATTN(false);      // (2 = two-product WEIGHT contractions: no weights here, same as 0)
#undef MVIP_ATTN
    return check_launch();
}

extern "C" int mvip_attention_f16x3(const void *qs, const void *ks, const void *vp, const float *q_scale2,
                                    const float *k_scale2, const float *v_scale2, int64_t N, int64_t heads, int64_t D,
                                    int64_t Lq, int64_t LqP, int64_t Lk, int64_t LkP, float softmax_scale, int flags,
                                    float *out, int prec, void *stream) {
    return attention_launch(qs, ks, vp, q_scale2, k_scale2, v_scale2, N, heads, D, Lq, LqP, Lk, LkP, Lq, LkP, LkP / 16,
                            softmax_scale, flags, out, nullptr, prec, stream);
}

// The same attention between two GEMMs that exchange OPERANDS: qs / ks / vp as written by mvip_gemm_f16x3_sinks (planes of
// q_stride / k_stride tokens, v_groups 16-key groups per V block: the producing GEMM's column count P and P / 16), and
// the result as the output projection's operand planes [N][heads*D/16][2][2][LqP][8 halves], still scaled by V's scale
// v_scale2[0] (pass v_scale2 as that GEMM's x_scale2).  Columns >= Lq of out_planes are not written.
extern "C" int mvip_attention_f16x3_sink(const void *qs, const void *ks, const void *vp, const float *q_scale2,
                                         const float *k_scale2, const float *v_scale2, int64_t N, int64_t heads, int64_t D,
                                         int64_t Lq, int64_t LqP, int64_t Lk, int64_t LkP, int64_t q_stride, int64_t k_stride,
                                         int64_t v_groups, float softmax_scale, int flags, void *out_planes, int prec,
                                         void *stream) {
    if (!out_planes) return MVIP_EINVAL;
    return attention_launch(qs, ks, vp, q_scale2, k_scale2, v_scale2, N, heads, D, Lq, LqP, Lk, LkP, q_stride, k_stride,
                            v_groups, softmax_scale, flags, nullptr, out_planes, prec, stream);
}
