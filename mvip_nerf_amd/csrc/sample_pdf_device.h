// Device side of the hierarchical inverse-CDF sampling + depth merge (DS_NeRF/run_nerf_helpers.py:304-347,
// DS_NeRF/run.py:1809-1816, :1836), shared by csrc/sample_pdf.hip (stand-alone launches) and the fused coarse pass of
// csrc/mlp_fwd16.hip.  One wavefront per ray, element e of a per-ray vector in lane e % 64, register e / 64.
#pragma once
#include "common.h"
#include <math.h>

namespace mvip {

template <int IT>
__device__ __forceinline__ float strided_get(const float (&v)[IT], int e) {
    float out = 0.f;
    const int src = e & 63, item = e >> 6;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const float t = __shfl(v[i], src, 64);
        if (item == i) out = t;
    }
    return out;
}

// value of lane `src` (0 .. 63, NOT masked here) of v: ds_bpermute on the byte address -- HIP's __shfl adds a mask, an or and a
// shift per read for widths and out-of-range lanes that do not occur in this file
__device__ __forceinline__ float lane_read(float v, int src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src << 2, __builtin_bit_cast(int, v)));
}

// #{k < 64 : key[k] < x} (STRICT = true) or <= x, for a lane-sorted key register (lanes beyond the valid count hold
// +inf): six binary-search steps on cross-lane reads + one probe for the count 64.  The running count is kept as a BYTE
// address (4 x count): the probe's lane (count + step - 1) is then that register plus a constant, which rides in the
// instruction's offset field, and a successful probe sets one bit -- compare, select, or: three VALU instructions per step.
template <bool STRICT>
__device__ __forceinline__ int count_below(const float key, const float x) {
    int cnt4 = 0;
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) {
        const float c = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(cnt4 + (step - 1) * 4, __builtin_bit_cast(int, key)));
        cnt4 |= (STRICT ? c < x : c <= x) ? step * 4 : 0;
    }
    const int cnt = cnt4 >> 2;
    const float c63 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, key), 63));   // the steps sum to at most 63
    // (arithmetic, not a branch: a lane-divergent `if` would split the basic block of rays_fast's interleaved chains)
    return cnt + (int)((cnt == 63) & (STRICT ? c63 < x : c63 <= x));
}

// bins/wts in strided registers (nb bins, nb-1 weights).  u strided (nf samples).
// Produces samples (strided) and inds; cdf_out strided (nb entries).
// Up to 64 bins (the reference configuration: 63 midpoints of 64 coarse samples): everything in one register
// per lane.  The normaliser is an fp64 wave sum, the CDF an fp64 wave scan (DPP row operations) rounded to fp32 per
// entry (torch's CPU cumsum accumulates in fp64 in index order; a different summation ORDER changes the fp64
// value by ~1e-16 relative, i.e. the rounded fp32 entry in ~1 of 1e8 cases), and searchsorted(right=True) is a
// six-step binary search on cross-lane reads of the sorted CDF.  ~150 instructions per ray instead of ~2000.
// Returns false (nothing written) if some pdf entry is negative: the CDF is then not sorted and the caller
// takes the index-ordered path.
__device__ __forceinline__ bool inverse_cdf_fast(const float bins, const float wts, int nb, const float u,
                                                 float &sample, int &ind, float &cdf_out) {
    const int l = lane_id();
    const int nw = nb - 1;
    const float w5 = wts + 1e-5f;
    // sums and scans on DPP row operations (common.h): six dependent VALU steps each instead of six ds_bpermute round trips
    // of two registers -- this kernel is bound by the latency of its dependent cross-lane chain, not by bandwidth
    const double tot = dpp_wave_sum(l < nw ? (double)w5 : 0.0);
    const float total = (float)tot;
    const float pdf = l < nw ? w5 / total : 0.f;
    if (__any(pdf < 0.f)) return false;
    const double run = dpp_incl_sum((double)pdf);
    const float incl = (float)run;                              // cdf[l + 1]
    const float cdf = dpp_from_prev(incl, 0.f);                 // cdf[l], valid for l < nb (cdf[0] = 0)
    const float key = l < nb ? cdf : INFINITY;
    const int cnt = count_below<false>(key, u);                 // #{k < nb : cdf[k] <= u}
    const int below = max(0, cnt - 1), above = min(nb - 1, cnt);
    const float cb = lane_read(cdf, below), ca = lane_read(cdf, above);
    const float bb = lane_read(bins, below), ba = lane_read(bins, above);
    float den = ca - cb;
    den = den < 1e-5f ? 1.f : den;
    const float t = (u - cb) / den;
    sample = bb + t * (ba - bb);
    ind = cnt;
    cdf_out = cdf;
    return true;
}

template <int IT>
__device__ __forceinline__ void inverse_cdf(const float (&bins)[IT], const float (&wts)[IT], int nb,
                                            const float (&u)[IT], int nf, float (&samples)[IT],
                                            int (&inds)[IT], float (&cdf)[IT]) {
    if constexpr (IT == 1) {
        if (inverse_cdf_fast(bins[0], wts[0], nb, u[0], samples[0], inds[0], cdf[0])) return;
    }
    const int l = lane_id();
    const int nw = nb - 1;
    // weights + 1e-5, total in index order
    float w5[IT];
    double total_d = 0.0;
#pragma unroll
    for (int i = 0; i < IT; ++i) w5[i] = wts[i] + 1e-5f;
#pragma unroll
    for (int i = 0; i < IT; ++i)
        for (int k = 0; k < 64; ++k) {
            if (i * 64 + k >= nw) break;
            total_d += (double)__shfl(w5[i], k, 64);
        }
    const float total = (float)total_d;
    // cdf[0] = 0; cdf[k+1] = cdf[k] + pdf[k]; count cdf entries <= u on the fly
    double run_d = 0.0;
#pragma unroll
    for (int i = 0; i < IT; ++i) { inds[i] = 0; cdf[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < IT; ++i) inds[i] += (0.f <= u[i]) ? 1 : 0;          // cdf[0] = 0
#pragma unroll
    for (int i = 0; i < IT; ++i)
        for (int k = 0; k < 64; ++k) {
            const int e = i * 64 + k;                  // weight index; writes cdf[e+1]
            if (e >= nw) break;
            const float pdf = __shfl(w5[i], k, 64) / total;
            run_d += (double)pdf;
            const float run = (float)run_d;
            const int dst = e + 1;
#pragma unroll
            for (int ii = 0; ii < IT; ++ii) {
                if ((dst >> 6) == ii && (dst & 63) == l) cdf[ii] = run;
                inds[ii] += (run <= u[ii]) ? 1 : 0;
            }
        }
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int below = max(0, inds[i] - 1);
        const int above = min(nb - 1, inds[i]);
        const float cb = strided_get<IT>(cdf, below), ca = strided_get<IT>(cdf, above);
        const float bb = strided_get<IT>(bins, below), ba = strided_get<IT>(bins, above);
        float den = ca - cb;
        den = den < 1e-5f ? 1.f : den;
        const float t = (u[i] - cb) / den;
        samples[i] = bb + t * (ba - bb);
    }
    (void)nf;
}

// Bitonic sort, ascending, of 64*M values in strided layout (index e = i*64 + lane).
template <int M>
__device__ __forceinline__ void bitonic_sort(float (&v)[M]) {
    const int l = lane_id();
#pragma unroll
    for (int k = 2; k <= 64 * M; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int dj = j >> 6;
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    if ((i & dj) == 0) {
                        const int e = i * 64 + l;
                        const bool up = (e & k) == 0;
                        const float a = v[i], b = v[i | dj];
                        const bool sw = up ? (a > b) : (a < b);
                        v[i] = sw ? b : a;
                        v[i | dj] = sw ? a : b;
                    }
                }
            } else {
#pragma unroll
                for (int i = 0; i < M; ++i) {
                    const int e = i * 64 + l;
                    const bool up = (e & k) == 0;
                    const float other = __shfl_xor(v[i], j, 64);
                    const bool lower = (l & j) == 0;
                    const float mn = fminf(v[i], other), mx = fmaxf(v[i], other);
                    v[i] = (lower == up) ? mn : mx;
                }
            }
        }
    }
}

// Bitonic sort of ONE value per lane (64 values), ascending.  A compare-exchange step is THREE VALU instructions: min and max
// take the partner lane as a DPP operand (v_min_f32_dpp / v_max_f32_dpp: quad permutes for lane distance 1 and 2, a row
// rotation for 8, two bank-masked row shifts for 4) and a v_cndmask picks by a CONSTANT 64-bit lane mask held in scalar
// registers.  Written as inline assembly: through fminf / fmaxf the compiler canonicalises both operands of every min / max
// (v_max_f32 x, x, x: signalling-NaN quieting, ~2 extra instructions per step) and rebuilds the lane predicate from the lane
// id each time -- ten issue slots per step instead of four, on a kernel that is bound by VALU issue.  Distances 16 and 32
// keep the swizzle / half-wave swap of common.h::dpp_xor.  NaNs: v_min / v_max return the other operand, as fminf / fmaxf do.
template <int K, int J>
constexpr unsigned long long bitonic_min_mask() {          // lanes that keep the MINIMUM of (own, partner l ^ J) in block size K
    unsigned long long m = 0;
    for (int l = 0; l < 64; ++l)
        if (((l & J) == 0) == ((l & K) == 0)) m |= 1ull << l;
    return m;
}
template <int K, int J>
__device__ __forceinline__ void bitonic_step64(float &v) {
    constexpr unsigned long long mask = bitonic_min_mask<K, J>();
    float mn, mx;
    if constexpr (J == 1)
        asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_max_f32_dpp %1, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=&v"(mn), "=&v"(mx) : "v"(v));
    else if constexpr (J == 2)
        asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_max_f32_dpp %1, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=&v"(mn), "=&v"(mx) : "v"(v));
    else if constexpr (J == 8)
        asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf\n\t"
                     "v_max_f32_dpp %1, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xf" : "=&v"(mn), "=&v"(mx) : "v"(v));
    else if constexpr (J == 4) {
        mn = v; mx = v;                                     // banks 0, 2 take lane + 4, banks 1, 3 lane - 4; a masked-off bank keeps the old value
        asm volatile("s_nop 1\n\tv_min_f32_dpp %0, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                     "v_max_f32_dpp %1, %2, %2 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
                     "v_min_f32_dpp %0, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
                     "v_max_f32_dpp %1, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa" : "+&v"(mn), "+&v"(mx) : "v"(v));
    } else {
        const float other = dpp_xor<J>(v);
        asm volatile("v_min_f32 %0, %2, %3\n\tv_max_f32 %1, %2, %3" : "=&v"(mn), "=&v"(mx) : "v"(v), "v"(other));
    }
    asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(v) : "v"(mx), "v"(mn), "s"(mask));
}
__device__ __forceinline__ void bitonic_sort64(float &v) {
    bitonic_step64<2, 1>(v);
    bitonic_step64<4, 2>(v); bitonic_step64<4, 1>(v);
    bitonic_step64<8, 4>(v); bitonic_step64<8, 2>(v); bitonic_step64<8, 1>(v);
    bitonic_step64<16, 8>(v); bitonic_step64<16, 4>(v); bitonic_step64<16, 2>(v); bitonic_step64<16, 1>(v);
    bitonic_step64<32, 16>(v); bitonic_step64<32, 8>(v); bitonic_step64<32, 4>(v); bitonic_step64<32, 2>(v);
    bitonic_step64<32, 1>(v);
    bitonic_step64<64, 32>(v); bitonic_step64<64, 16>(v); bitonic_step64<64, 8>(v); bitonic_step64<64, 4>(v);
    bitonic_step64<64, 2>(v); bitonic_step64<64, 1>(v);
}

// inclusive prefix MAXIMUM of non-negative integers over the 64 lanes (the DPP scan of common.h with max instead of add)
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ int dpp_i32(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, BANK_MASK, false); }
__device__ __forceinline__ int dpp_incl_max_nonneg(int v) {
    v = max(v, dpp_i32<0x111>(0, v));
    v = max(v, dpp_i32<0x112>(0, v));
    v = max(v, dpp_i32<0x114>(0, v));
    v = max(v, dpp_i32<0x118>(0, v));
    v = max(v, dpp_i32<0x142, 0xa>(0, v));
    v = max(v, dpp_i32<0x143, 0xc>(0, v));
    return v;
}

// Merge of two SORTED lists of <= 64 values each by rank: element i of a lands at i + #{b < a_i}, element j of b at
// j + #{a <= b_j} (ties: the coarse depth first; the VALUES equal those of sort(cat[a, b]) in every case).  12
// cross-lane reads and two scattered 4-byte stores into the ray's own 512-byte row instead of the 28-stage / 54-shuffle
// bitonic network over 128 values.  Returns false (nothing written) if either list is not sorted.
// below_hint (>= 0, or -1 for none): lane j's sample b_j was interpolated between the midpoints below_hint and below_hint + 1
// (or sits on the last one) of the depths a -- what the inverse CDF knows anyway.  Then a_0 .. a_below <= b_j, so
// #{a <= b_j} is below + 1 or below + 2, decided by ONE cross-lane read instead of a seven-step search; a second read
// checks that the depth after that is strictly greater (it is, unless depths coincide or the interpolation rounded past
// the upper midpoint), and the wave falls back to the search if any lane fails the check.  Only meaningful while b is in
// its original lane order, i.e. when it did not have to be sorted.
__device__ __forceinline__ bool rank_merge64(const float a, int na, float b, int nb, bool b_may_be_unsorted,
                                             float *__restrict__ out_row, int below_hint = -1, int *lds_w = nullptr) {
    const int l = lane_id();
    const float a_key = l < na ? a : INFINITY;
    float b_key = l < nb ? b : INFINITY;
    const float a_next = dpp_from_next(a_key, a_key), b_next = dpp_from_next(b_key, b_key);
    // NaN depths (leave them to the network) or coarse depths out of order: one ballot
    if (__any((a_key != a_key) | (b_key != b_key) | (l < 63 && a_next < a_key))) return false;
    bool b_in_lane_order = true;
    if (__any(l < 63 && b_next < b_key)) {
        if (!b_may_be_unsorted) return false;
        bitonic_sort64(b_key);                                   // random u: sort the 64 new samples (21 stages)
        b_in_lane_order = false;
    }
    // counts clamped to the VALID entries: a valid key equal to +inf (far = inf, non-lindisp) would otherwise count the
    // +inf padding lanes of the other list and land beyond the ray's own row
    int cb = -1;
    if (b_in_lane_order && below_hint >= 0) {
        const int r0 = below_hint + 1;                           // a_0 .. a_{r0 - 1} <= b_j
        const float a0 = lane_read(a_key, r0 & 63), a1 = lane_read(a_key, (r0 + 1) & 63);
        const float e0 = r0 < 64 ? a0 : INFINITY, e1 = r0 + 1 < 64 ? a1 : INFINITY;
        cb = r0 + (e0 <= b_key ? 1 : 0);
        if (__any(l < nb && !(e1 > b_key))) cb = -1;             // wave-uniform: the ballot covers every valid lane
    }
    const int cbj = min(cb >= 0 ? cb : count_below<false>(a_key, b_key), na);        // #{a <= b_j}, non-decreasing in j
    int ca;                                                                         // #{b < a_i}
    if (lds_w) {
        // b_j < a_i  <=>  i >= cb_j (a sorted, cb_j = index of the first a above b_j), and cb_j does not decrease with j: so
        // #{b < a_i} = 1 + the LAST j with cb_j <= i.  Every run of equal cb_j leaves its last j + 1 at word cb_j of a zeroed
        // 65-word row of LDS (one writer per word), lane i reads word i, and an inclusive prefix maximum (six DPP steps)
        // finishes the count: 3 LDS operations + ~15 VALU instructions instead of the seven-step cross-lane search (~45).
        // One wave owns the row and LDS operations of a wave execute in order: no barrier.
        // LDS address space spelled out: through the generic pointer these were flat_store / flat_load with sc0 sc1 and a
        // full s_waitcnt vmcnt(0) after each of the four accesses
        volatile __attribute__((address_space(3))) int *row = (volatile __attribute__((address_space(3))) int *)lds_w;
        row[l] = 0;
        if (l == 0) row[64] = 0;
        const int cb_next = dpp_i32<0x130>(-1, cbj);            // lane l + 1's count (lane 63: -1 = "differs")
        if (l < nb && (l == nb - 1 || cb_next != cbj)) row[cbj] = l + 1;
        ca = dpp_incl_max_nonneg(row[l]);
    } else {
        ca = min(count_below<true>(b_key, a_key), nb);
    }
    const int pa = l + ca, pb = l + cbj;
    if (l < na) out_row[pa] = a_key;
    if (l < nb) out_row[pb] = b_key;
    return true;
}

// One ray's inverse-CDF resampling + merge from REGISTERS (strided layout: element e in lane e % 64, register e / 64):
// zc = the Nc coarse depths, wts = the Nc - 2 interior weights (weight e of the pdf = coarse weight e + 1), uu = the Nf
// uniforms (2.f beyond Nf).  What sample_pdf_merge_kernel does after its loads, as a function so that a fused render
// kernel (csrc/mlp_fwd16.hip) produces the same bits.  z_samples / inds_out / cdf_out may be null.
template <int IT>
__device__ __forceinline__ void sample_merge_ray(const float (&zc)[IT], const float (&wts)[IT], const float (&uu)[IT], int64_t ray,
                                                 int Nc, int Nf, float *__restrict__ z_samples, float *__restrict__ z_merged,
                                                 float *__restrict__ z_std, int64_t *__restrict__ inds_out,
                                                 float *__restrict__ cdf_out, int *lds_w = nullptr) {
    const int l = lane_id();
    const int nb = Nc - 1;                           // midpoints
    float bins[IT], smp[IT], cdf[IT];
    int inds[IT];
    // mids[e] = .5 * (z[e+1] + z[e])
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        float nxt;
        if constexpr (IT == 1) nxt = dpp_from_next(zc[0], 0.f);
        else {
            nxt = __shfl_down(zc[i], 1, 64);
            const float wrap = __shfl(zc[(i + 1) % IT], 0, 64);  // all lanes take part in the shuffle
            if (l == 63) nxt = (i + 1 < IT) ? wrap : 0.f;
        }
        bins[i] = .5f * (nxt + zc[i]);
    }
    inverse_cdf<IT>(bins, wts, nb, uu, Nf, smp, inds, cdf);
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int e = i * 64 + l;
        if (e < Nf) {
            if (z_samples) z_samples[ray * Nf + e] = smp[i];
            if (inds_out) inds_out[ray * Nf + e] = inds[i];
            s1 += smp[i];
        }
        if (cdf_out && e < nb) cdf_out[ray * nb + e] = cdf[i];
    }
    // population std of the new samples (torch.std(unbiased=False)).  1 / Nf by v_rcp_f32 (exact for the powers of two of the
    // reference configurations, 1 ulp otherwise) and v_sqrt_f32 (1 ulp): the sums above are wave reductions, not torch's
    // summation order, so the statistic is compared at a tolerance either way -- and two IEEE division sequences + one square
    // root sequence were ~35 of this kernel's ~400 VALU instructions per ray
    const float inv_nf = __builtin_amdgcn_rcpf((float)Nf);
    const float mean = dpp_wave_sum(s1) * inv_nf;
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int e = i * 64 + l;
        if (e < Nf) { const float d = smp[i] - mean; s2 += d * d; }
    }
    s2 = dpp_wave_sum(s2);
    if (l == 0) z_std[ray] = __builtin_amdgcn_sqrtf(s2 * inv_nf);
    // merge: sort(cat[z, z_samples]).  Both lists are sorted in the reference configuration (stratified coarse depths;
    // the inverse CDF is monotone, so the new samples are sorted whenever u is -- always in deterministic mode): rank merge.
    if constexpr (IT == 1) {
        if (rank_merge64(zc[0], Nc, smp[0], Nf, true, z_merged + ray * (Nc + Nf), max(0, inds[0] - 1), lds_w)) return;
    }
    constexpr int M = 2 * IT;
    float v[M];
#pragma unroll
    for (int i = 0; i < IT; ++i) {
        const int e = i * 64 + l;
        v[i] = e < Nc ? zc[i] : INFINITY;
        v[IT + i] = e < Nf ? smp[i] : INFINITY;
    }
    bitonic_sort<M>(v);
    const int N = Nc + Nf;
#pragma unroll
    for (int i = 0; i < M; ++i) {
        const int e = i * 64 + l;
        if (e < N) z_merged[ray * N + e] = v[i];
    }
}

// ---- N rays per wave, their dependent chains interleaved STEP BY STEP (round 6) ----------------------------------------
// The one-ray kernel is bound by the LATENCY of a ray's dependent chain at the eight waves a SIMD holds (occupancy probe:
// time x1.7 from 2 to 4 waves per SIMD, x1.5 from 4 to 8; fewer instructions or prefetched rows change nothing:
// profiles/r6_sample_merge_occupancy_probe.json, r6_sample_merge_ab.jsonl).  rays_fast is sample_merge_ray<1>'s common route for
// the reference configuration (64 coarse depths, 64 new samples) WITHOUT wave-uniform branches and lane predicates -- the exits
// are collected in `ok`, the results stay in registers, the caller stores them -- written so that every step of every
// primitive is issued for ray 0, ray 1, ... before the next step: the compiler keeps that order (two whole chains written
// one after the other were NOT interleaved by the scheduler), and between two dependent instructions of one ray stands an
// independent one of the other.  A ray whose `ok` is false is redone by sample_merge_ray<1>.  Arithmetic and results are those
// of sample_merge_ray<1>, bit for bit (SORT: the new samples are always sorted -- a no-op for a row that already is -- and
// located by the search; otherwise they must be sorted and are located by the interval hint).
template <int N, class T, class Dpp>
__device__ __forceinline__ void dpp_incl_scan_add_n(T (&v)[N], Dpp) {
#define MVIP_STEP(EXPR) _Pragma("unroll") for (int r = 0; r < N; ++r) v[r] += EXPR;
    MVIP_STEP(Dpp::template z<0x111>(v[r]))
    MVIP_STEP(Dpp::template z<0x112>(v[r]))
    MVIP_STEP(Dpp::template z<0x114>(v[r]))
    MVIP_STEP(Dpp::template z<0x118>(v[r]))
    MVIP_STEP((Dpp::template f<0x142, 0xa>((T)0, v[r])))
    MVIP_STEP((Dpp::template f<0x143, 0xc>((T)0, v[r])))
#undef MVIP_STEP
}
// Total over the wave (only lane 63 of the scan is read): the two cross-row steps run UNMASKED with bound_ctrl -- lane 63 receives
// exactly the additions of the scan, the other rows hold values nobody reads, and no destination has to be zero-filled first
// (two moves per masked step and 32-bit half).
template <int N, class T, class Dpp>
__device__ __forceinline__ void dpp_total_in_lane63_n(T (&v)[N], Dpp) {
#define MVIP_STEP(EXPR) _Pragma("unroll") for (int r = 0; r < N; ++r) v[r] += EXPR;
    MVIP_STEP(Dpp::template z<0x111>(v[r]))
    MVIP_STEP(Dpp::template z<0x112>(v[r]))
    MVIP_STEP(Dpp::template z<0x114>(v[r]))
    MVIP_STEP(Dpp::template z<0x118>(v[r]))
    MVIP_STEP(Dpp::template z<0x142>(v[r]))
    MVIP_STEP(Dpp::template z<0x143>(v[r]))
#undef MVIP_STEP
}
template <int N>
__device__ __forceinline__ void dpp_wave_sum_n(double (&v)[N]) {
    dpp_total_in_lane63_n<N, double>(v, DppF64{});
#pragma unroll
    for (int r = 0; r < N; ++r) {
        const long long t = __builtin_bit_cast(long long, v[r]);
        const int lo = __builtin_amdgcn_readlane((int)t, 63), hi = __builtin_amdgcn_readlane((int)(t >> 32), 63);
        v[r] = __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned)lo);
    }
}
template <int N>
__device__ __forceinline__ void dpp_wave_sum_n(float (&v)[N]) {
    dpp_total_in_lane63_n<N, float>(v, DppF32Z{});
#pragma unroll
    for (int r = 0; r < N; ++r) v[r] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v[r]), 63));
}
// (LAST_IS_INF: the caller's key of lane 63 is +inf, which no x reaches: the six steps already give the count)
template <bool STRICT, int N, bool LAST_IS_INF = false>
__device__ __forceinline__ void count_below_n(const float (&key)[N], const float (&x)[N], int (&cnt)[N]) {
    int cnt4[N];
#pragma unroll
    for (int r = 0; r < N; ++r) cnt4[r] = 0;
#pragma unroll
    for (int step = 32; step > 0; step >>= 1) {
        float c[N];
#pragma unroll
        for (int r = 0; r < N; ++r)
            c[r] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(cnt4[r] + (step - 1) * 4, __builtin_bit_cast(int, key[r])));
#pragma unroll
        for (int r = 0; r < N; ++r) cnt4[r] |= (STRICT ? c[r] < x[r] : c[r] <= x[r]) ? step * 4 : 0;
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
        const int k = cnt4[r] >> 2;
        if constexpr (LAST_IS_INF) { cnt[r] = k; continue; }
        const float c63 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, key[r]), 63));
        cnt[r] = k + (int)((k == 63) & (STRICT ? c63 < x[r] : c63 <= x[r]));
    }
}
// The 21-step network on SIGNED keys, N rays at a time.  A lane that keeps the maximum of (own, partner) in a step holds its
// key NEGATED during that step: partners always have opposite roles, so with t = +-v both compute t' = min(t, -t_partner)
// (min(v, p) on the one side, -max(v, p) on the other) -- ONE v_min_f32_dpp with a negated DPP operand instead of
// v_min_dpp + v_max_dpp + v_cndmask; between steps ONE v_cndmask with a negated source and a constant lane mask moves every
// lane to the next step's sign.  51 vector instructions per ray instead of 83; the values are those of bitonic_sort64 (NaNs are
// excluded by the caller).  One assembly statement per step covers all N rays: the N - 1 instructions of the other rays stand
// between the v_cndmask that writes a key and the DPP read of it (2 wait states: an s_nop 0 for N = 2, nothing for N >= 3).
#define MVIP_FLIP(i) "v_cndmask_b32_e64 %" #i ", %" #i ", -%" #i ", %[m]\n\t"
#define MVIP_MIN(i, CTRL) "v_min_f32_dpp %" #i ", -%" #i ", %" #i " " CTRL "\n\t"
#define MVIP_MINO(o, i, CTRL) "v_min_f32_dpp %" #o ", -%" #i ", %" #i " " CTRL "\n\t"
template <int N, unsigned long long FLIP, int J>
__device__ __forceinline__ void signed_step_n(float (&t)[N]) {
    static_assert(N >= 2 && N <= 4, "");
#define MVIP_DPP_STEP(CTRL)                                                                                                              \
    if constexpr (N == 2)                                                                                                                \
        asm(MVIP_FLIP(0) MVIP_FLIP(1) "s_nop 0\n\t" MVIP_MIN(0, CTRL) MVIP_MIN(1, CTRL) : "+v"(t[0]), "+v"(t[1]) : [m] "s"(FLIP));         \
    else if constexpr (N == 3)                                                                                                           \
        asm(MVIP_FLIP(0) MVIP_FLIP(1) MVIP_FLIP(2) MVIP_MIN(0, CTRL) MVIP_MIN(1, CTRL) MVIP_MIN(2, CTRL)                                  \
            : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]) : [m] "s"(FLIP));                                                                       \
    else                                                                                                                                 \
        asm(MVIP_FLIP(0) MVIP_FLIP(1) MVIP_FLIP(2) MVIP_FLIP(3) MVIP_MIN(0, CTRL) MVIP_MIN(1, CTRL) MVIP_MIN(2, CTRL) MVIP_MIN(3, CTRL)   \
            : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) : [m] "s"(FLIP));
    if constexpr (J == 1) { MVIP_DPP_STEP("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf") }
    else if constexpr (J == 2) { MVIP_DPP_STEP("quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf") }
    else if constexpr (J == 8) { MVIP_DPP_STEP("row_ror:8 row_mask:0xf bank_mask:0xf") }
    else if constexpr (J == 4) {                             // banks 0, 2 take lane + 4, banks 1, 3 lane - 4: every lane of o is written once
        float o[N];
#define MVIP_A "row_shl:4 row_mask:0xf bank_mask:0x5"
#define MVIP_B "row_shr:4 row_mask:0xf bank_mask:0xa"
        if constexpr (N == 2)
            asm(MVIP_FLIP(2) MVIP_FLIP(3) "s_nop 0\n\t" MVIP_MINO(0, 2, MVIP_A) MVIP_MINO(1, 3, MVIP_A) MVIP_MINO(0, 2, MVIP_B) MVIP_MINO(1, 3, MVIP_B)
                : "=&v"(o[0]), "=&v"(o[1]), "+v"(t[0]), "+v"(t[1]) : [m] "s"(FLIP));
        else if constexpr (N == 3)
            asm(MVIP_FLIP(3) MVIP_FLIP(4) MVIP_FLIP(5) MVIP_MINO(0, 3, MVIP_A) MVIP_MINO(1, 4, MVIP_A) MVIP_MINO(2, 5, MVIP_A)
                MVIP_MINO(0, 3, MVIP_B) MVIP_MINO(1, 4, MVIP_B) MVIP_MINO(2, 5, MVIP_B)
                : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]) : [m] "s"(FLIP));
        else
            asm(MVIP_FLIP(4) MVIP_FLIP(5) MVIP_FLIP(6) MVIP_FLIP(7) MVIP_MINO(0, 4, MVIP_A) MVIP_MINO(1, 5, MVIP_A) MVIP_MINO(2, 6, MVIP_A)
                MVIP_MINO(3, 7, MVIP_A) MVIP_MINO(0, 4, MVIP_B) MVIP_MINO(1, 5, MVIP_B) MVIP_MINO(2, 6, MVIP_B) MVIP_MINO(3, 7, MVIP_B)
                : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]) : [m] "s"(FLIP));
#undef MVIP_A
#undef MVIP_B
#pragma unroll
        for (int r = 0; r < N; ++r) t[r] = o[r];
    } else {                                                 // 16: ds_swizzle, 32: the half-wave swap (common.h::dpp_xor); 3 of the 21 steps
#pragma unroll
        for (int r = 0; r < N; ++r) asm("v_cndmask_b32_e64 %0, %0, -%0, %1" : "+v"(t[r]) : "s"(FLIP));
        float other[N];
#pragma unroll
        for (int r = 0; r < N; ++r) other[r] = dpp_xor<J>(t[r]);
#pragma unroll
        for (int r = 0; r < N; ++r) asm("v_min_f32_e64 %0, %0, -%1" : "+v"(t[r]) : "v"(other[r]));
    }
#undef MVIP_DPP_STEP
}
#undef MVIP_FLIP
#undef MVIP_MIN
#undef MVIP_MINO
template <int K, int J>
constexpr unsigned long long bitonic_neg_mask() { return ~bitonic_min_mask<K, J>(); }      // lanes that keep the maximum: key held negated
template <int N>
__device__ __forceinline__ void bitonic_sort64_n(float (&v)[N]) {
    // MVIP_STEP(K, J, PK, PJ): step (K, J) after step (PK, PJ); the flip mask is the XOR of the two sign masks
#define MVIP_STEP(K, J, PK, PJ) signed_step_n<N, bitonic_neg_mask<K, J>() ^ bitonic_neg_mask<PK, PJ>(), J>(v);
    signed_step_n<N, bitonic_neg_mask<2, 1>(), 1>(v);
    MVIP_STEP(4, 2, 2, 1) MVIP_STEP(4, 1, 4, 2)
    MVIP_STEP(8, 4, 4, 1) MVIP_STEP(8, 2, 8, 4) MVIP_STEP(8, 1, 8, 2)
    MVIP_STEP(16, 8, 8, 1) MVIP_STEP(16, 4, 16, 8) MVIP_STEP(16, 2, 16, 4) MVIP_STEP(16, 1, 16, 2)
    MVIP_STEP(32, 16, 16, 1) MVIP_STEP(32, 8, 32, 16) MVIP_STEP(32, 4, 32, 8) MVIP_STEP(32, 2, 32, 4) MVIP_STEP(32, 1, 32, 2)
    MVIP_STEP(64, 32, 32, 1) MVIP_STEP(64, 16, 64, 32) MVIP_STEP(64, 8, 64, 16) MVIP_STEP(64, 4, 64, 8) MVIP_STEP(64, 2, 64, 4)
    MVIP_STEP(64, 1, 64, 2)
#undef MVIP_STEP
#pragma unroll
    for (int r = 0; r < N; ++r) asm("v_cndmask_b32_e64 %0, %0, -%0, %1" : "+v"(v[r]) : "s"(bitonic_neg_mask<64, 1>()));
}
template <int N>
__device__ __forceinline__ void dpp_incl_max_nonneg_n(int (&v)[N]) {
#define MVIP_STEP(EXPR) _Pragma("unroll") for (int r = 0; r < N; ++r) v[r] = max(v[r], EXPR);
    MVIP_STEP(dpp_i32<0x111>(0, v[r]))
    MVIP_STEP(dpp_i32<0x112>(0, v[r]))
    MVIP_STEP(dpp_i32<0x114>(0, v[r]))
    MVIP_STEP(dpp_i32<0x118>(0, v[r]))
    MVIP_STEP((dpp_i32<0x142, 0xa>(0, v[r])))
    MVIP_STEP((dpp_i32<0x143, 0xc>(0, v[r])))
#undef MVIP_STEP
}

template <int N>
struct RaysFast {
    float smp[N], cdf[N], zstd[N], b_key[N];
    int ind[N], pa[N], pb[N];
    bool ok[N];
};
#define MVIP_EACH _Pragma("unroll") for (int r = 0; r < N; ++r)
template <bool SORT, int N>
__device__ __forceinline__ void rays_fast(const float (&zc)[N], const float (&wts)[N], const float (&u)[N], int *lds_w, int row_words,
                                          RaysFast<N> &o) {
    constexpr int Nc = 64, Nf = 64, nb = 63, nw = 62;
    const int l = lane_id();
    float bins[N], w5[N], total[N], pdf[N], cdf[N], key[N], smp[N];
    double acc[N];
    int cnt[N], below[N], above[N];
    bool ok[N];
    MVIP_EACH bins[r] = .5f * (dpp_from_next(zc[r], 0.f) + zc[r]);
    // ---- inverse CDF (inverse_cdf_fast) ----
    MVIP_EACH w5[r] = wts[r] + 1e-5f;
    MVIP_EACH acc[r] = l < nw ? (double)w5[r] : 0.0;
    dpp_wave_sum_n<N>(acc);
    MVIP_EACH total[r] = (float)acc[r];
    MVIP_EACH pdf[r] = (w5[r] / total[r]) * (l < nw ? 1.f : 0.f);      // (x 1 / x 0 exact; a select became a branch around the division)
    MVIP_EACH ok[r] = !__any(pdf[r] < 0.f);
    MVIP_EACH acc[r] = (double)pdf[r];
    dpp_incl_scan_add_n<N, double>(acc, DppF64{});
    MVIP_EACH cdf[r] = dpp_from_prev((float)acc[r], 0.f);
    MVIP_EACH key[r] = l < nb ? cdf[r] : INFINITY;
    count_below_n<false, N, true>(key, u, cnt);                       // key[63] = +inf (63 midpoints)
    MVIP_EACH { below[r] = max(0, cnt[r] - 1); above[r] = min(nb - 1, cnt[r]); }
    float cb_[N], ca_[N], bb[N], ba[N];
    MVIP_EACH { cb_[r] = lane_read(cdf[r], below[r]); ca_[r] = lane_read(cdf[r], above[r]); }
    MVIP_EACH { bb[r] = lane_read(bins[r], below[r]); ba[r] = lane_read(bins[r], above[r]); }
    MVIP_EACH {
        float den = ca_[r] - cb_[r];
        den = den < 1e-5f ? 1.f : den;
        const float t = (u[r] - cb_[r]) / den;
        smp[r] = bb[r] + t * (ba[r] - bb[r]);
    }
    // ---- z_std (sample_merge_ray) ----
    const float inv_nf = __builtin_amdgcn_rcpf((float)Nf);
    float red[N], dev[N];
    MVIP_EACH red[r] = smp[r];
    dpp_wave_sum_n<N>(red);
    MVIP_EACH { dev[r] = smp[r] - red[r] * inv_nf; red[r] = dev[r] * dev[r]; }
    dpp_wave_sum_n<N>(red);
    MVIP_EACH o.zstd[r] = __builtin_amdgcn_sqrtf(red[r] * inv_nf);
    // ---- merge by rank (rank_merge64) ----
    float b_key[N];
    int cbj[N];
    MVIP_EACH b_key[r] = smp[r];
    MVIP_EACH {
        const float a_next = dpp_from_next(zc[r], zc[r]);
        ok[r] = ok[r] & !__any((zc[r] != zc[r]) | (b_key[r] != b_key[r]) | ((l < 63) & (a_next < zc[r])));
    }
    if constexpr (SORT) {
        bitonic_sort64_n<N>(b_key);
        count_below_n<false, N>(zc, b_key, cbj);
        MVIP_EACH cbj[r] = min(cbj[r], Nc);
    } else {
        MVIP_EACH {
            const float b_next = dpp_from_next(b_key[r], b_key[r]);
            ok[r] = ok[r] & !__any((l < 63) & (b_next < b_key[r]));
        }
        float e0[N], e1[N];
        MVIP_EACH {
            const int r0 = below[r] + 1;                         // a_0 .. a_{r0 - 1} <= b_j
            const float a0 = lane_read(zc[r], r0 & 63), a1 = lane_read(zc[r], (r0 + 1) & 63);
            e0[r] = r0 < 64 ? a0 : INFINITY;
            e1[r] = r0 + 1 < 64 ? a1 : INFINITY;
        }
        MVIP_EACH {
            ok[r] = ok[r] & !__any(!(e1[r] > b_key[r]));
            cbj[r] = min(below[r] + 1 + (e0[r] <= b_key[r] ? 1 : 0), Nc);
        }
    }
    int ca[N];
    MVIP_EACH {
        volatile __attribute__((address_space(3))) int *row = (volatile __attribute__((address_space(3))) int *)(lds_w + r * row_words);
        row[l] = 0;
        row[64 + (l & 7)] = 0;                                   // (every lane writes: no lane predicate, no branch)
    }
    MVIP_EACH {
        volatile __attribute__((address_space(3))) int *row = (volatile __attribute__((address_space(3))) int *)(lds_w + r * row_words);
        const int cb_next = dpp_i32<0x130>(-1, cbj[r]);
        const bool last_of_run = (l == Nf - 1) | (cb_next != cbj[r]);
        row[last_of_run ? cbj[r] : 72 + (l & 7)] = l + 1;        // the other lanes write a word nobody reads
    }
    MVIP_EACH {
        volatile __attribute__((address_space(3))) int *row = (volatile __attribute__((address_space(3))) int *)(lds_w + r * row_words);
        ca[r] = row[l];
    }
    dpp_incl_max_nonneg_n<N>(ca);
    MVIP_EACH {
        o.smp[r] = smp[r]; o.cdf[r] = cdf[r]; o.ind[r] = cnt[r]; o.b_key[r] = b_key[r];
        o.pa[r] = l + ca[r]; o.pb[r] = l + cbj[r]; o.ok[r] = ok[r];
    }
}
#undef MVIP_EACH

}  // namespace mvip
