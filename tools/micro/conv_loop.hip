// The 3x3 convolution's stage loop rebuilt piece by piece: which ingredient takes the matrix pipe from the 32.0 cycles per
// MFMA of mfma16_lds.hip to the ~53 the real kernel shows (512 channels at 128 x 128: 219 us against 133 us of matrix issue)?
// Two four-wave workgroups per CU (80 KB of LDS each), every CU busy.  One "stage" = 3 taps x 2 row blocks = 6 groups of
// 6 MFMAs on four accumulators, fragments read from LDS two groups ahead exactly as conv3x3_f16x3_kernel<2> does
// (A: two 1-KB blocks per group; B: four per tap, pixel-indexed with the haloed row stride).
//   F & 1: per-stage workgroup barrier          F & 2: per-stage LDS-DMA of the next stage's operands (19.5 KB)
//   F & 4: B fragments at the kernel's pixel addresses (else wave-linear)
//   F & 16: all six MFMAs of a group take the SAME two operand registers (how much of the power is operand delivery?)
//   F & 32: the MFMAs of a group walk B first (each A fragment stays for ONE MFMA instead of four: the opposite order)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

constexpr int WB = 12288, INB = 21760, PIX = 340, HW = 34;
constexpr int LDS_BYTES = 3 * WB + 2 * INB;            // 80,384: the kernel's footprint, two workgroups per CU

__device__ __forceinline__ void glds(const void *src_lane, void *dst_wave) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src_lane,
                                     (__attribute__((address_space(3))) void *)dst_wave, 16, 0, 0);
}

template <int F>
__global__ void __launch_bounds__(256, 2) k(unsigned long long *out, const float *w, const char *src, int stages) {
    __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l32 = lane & 31, kg = lane >> 5;
    for (int i = threadIdx.x; i < LDS_BYTES / 2; i += 256) reinterpret_cast<_Float16 *>(lds)[i] = (_Float16)(w[i & 16383] * 0.064f);
    __syncthreads();
    f32x16 acc[2][2];
    for (int m = 0; m < 2; ++m) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[m][j][r] = 0.f;
    h16x8 Ah[3], Al[3], Bh[3][2], Bl[3][2];
    char *lds_w = lds, *lds_in = lds + 3 * WB;
    auto w_base = [&](int t) { return lds_w + (t % 3) * WB + lane * 16; };
    auto in_base = [&](int t) { return lds_in + ((t / 3) & 1) * INB + (kg * 2) * (PIX * 16); };
    auto load_a = [&](const char *wb, int g, int set) {
        const int kx = g / 2, m = g % 2;
        Ah[set] = *reinterpret_cast<const h16x8 *>(wb + ((m * 3 + kx) * 2 + 0) * 1024);
        Al[set] = *reinterpret_cast<const h16x8 *>(wb + ((m * 3 + kx) * 2 + 1) * 1024);
    };
    auto load_b = [&](const char *inb, int ky, int kx) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int p = ((2 * wave + j) + ky) * HW + l32 + kx;
            if (!(F & 4)) p = (j * 3 + kx) * 64 + lane - (kg * 2) * PIX;
            Bh[kx][j] = *reinterpret_cast<const h16x8 *>(inb + p * 16);
            Bl[kx][j] = *reinterpret_cast<const h16x8 *>(inb + PIX * 16 + p * 16);
        }
    };
    const char *gsrc = src + ((size_t)blockIdx.x % 64) * 65536 + lane * 16;
    auto issue = [&](int t) {                           // 12 KB of weights + a third of an input chunk, as the kernel per stage
        char *dw = lds_w + (t % 3) * WB;
        for (int b = wave; b < 12; b += 4) glds(gsrc + b * 1024, dw + b * 1024);
        char *di = lds_in + ((t / 3 + 1) & 1) * INB + (t % 3) * 7168;
        for (int b = wave; b < 7; b += 4) glds(gsrc + 16384 + b * 1024, di + b * 1024);
    };
    for (int g = 0; g < 2; ++g) { load_a(w_base(0), g, g % 3); if (g % 2 == 0) load_b(in_base(0), 0, g / 2); }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int t = 0; t < stages; ++t) {
        const int ck = t / 3, ky = t - ck * 3;
        if (t > 0) {
            if (F & 1) __syncthreads();
            if (F & 2) issue(t + 2);
        }
        const char *wb = w_base(t), *inb = in_base(t), *wb_n = w_base(t + 1), *inb_n = in_base(t + 1);
        const int ky_n = ky == 2 ? 0 : ky + 1;
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int kx = g / 2, m = g % 2, set = g % 3;
            const h16x8 a0 = Ah[set], a1 = (F & 16) ? Ah[set] : Al[set];
            const h16x8 b0 = Bh[kx][0], b1 = (F & 16) ? b0 : Bh[kx][1], b2 = (F & 16) ? b0 : Bl[kx][0], b3 = (F & 16) ? b0 : Bl[kx][1];
            if (F & 32) {
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[m][1], 0, 0, 0);
            } else {
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[m][1], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            {
                const int g2 = g + 2;
                if (g2 < 6) { load_a(wb, g2, g2 % 3); if (g2 % 2 == 0) load_b(inb, ky, g2 / 2); }
                else { load_a(wb_n, g2 - 6, g2 % 3); if ((g2 - 6) % 2 == 0) load_b(inb_n, ky_n, (g2 - 6) / 2); }
            }
            __builtin_amdgcn_sched_barrier(0);
            if (F & 32) {
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b2, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b3, acc[m][1], 0, 0, 0);
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[m][1], 0, 0, 0);
            } else {
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b2, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b3, acc[m][1], 0, 0, 0);
                acc[m][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[m][0], 0, 0, 0);
                acc[m][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc[m][1], 0, 0, 0);
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int m = 0; m < 2; ++m) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) s += acc[m][j][r];
    if (s == 12345.678f) out[0] = 1;
    if (lane == 0) out[1 + blockIdx.x * 4 + wave] = t1 - t0;
}

template <int F>
void run(unsigned long long *d, const float *w, const char *src, int reps = 2) {
    const int blocks = 512, stages = 960;
    // `reps` launches back to back (~1.6 ms each); the LAST one is timed: the clock needs tens of milliseconds to settle
    for (int r = 0; r + 1 < reps; ++r) hipLaunchKernelGGL((k<F>), dim3(blocks), dim3(256), 0, 0, d, w, src, stages);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<F>), dim3(blocks), dim3(256), 0, 0, d, w, src, stages);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(1 + blocks * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int i = 0; i < blocks * 4; ++i) c.push_back((double)h[1 + i] / (stages * 36.0) / 2);   // two waves share a SIMD
    std::sort(c.begin(), c.end());
    // wall: 512 workgroups x 4 waves x stages x 36 MFMAs over 1024 SIMDs
    const double mfma_per_simd = 2.0 * stages * 36.0;
    printf("{\"barrier\": %d, \"dma\": %d, \"pixel_b_addresses\": %d, \"simd_cycles_per_mfma_median\": %.2f, \"p90\": %.2f, "
           "\"launch_ms\": %.3f, \"ns_per_mfma_slot\": %.2f, \"implied_MHz\": %.0f, \"same_operands\": %d, \"a_changes_every_mfma\": %d, \"launches_back_to_back\": %d}\n", F & 1, (F >> 1) & 1, (F >> 2) & 1, c[c.size() / 2],
           c[c.size() * 9 / 10], ms, ms * 1e6 / mfma_per_simd, c[c.size() / 2] / (ms * 1e6 / mfma_per_simd) * 1e3, (F >> 4) & 1, (F >> 5) & 1, reps);
}

int main() {
    unsigned long long *d;
    float *w;
    char *src;
    hipMalloc(&d, (1 + 512 * 4) * 8);
    hipMemset(d, 0, (1 + 512 * 4) * 8);
    std::vector<float> hw(1 << 16);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMalloc(&w, hw.size() * 4);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&src, 64 * 65536 + 65536);
    {   // DMA sources: small finite fp16 values like the LDS image (zeros would lower the power and raise the clock)
        std::vector<_Float16> hs((64 * 65536 + 65536) / 2);
        for (size_t i = 0; i < hs.size(); ++i) hs[i] = (_Float16)(((float)((i * 2654435761u) % 1000) / 1000.f - 0.5f) * 1e-3f * 64.f);
        hipMemcpy(src, hs.data(), hs.size() * 2, hipMemcpyHostToDevice);
    }
    run<0>(d, w, src); run<4>(d, w, src); run<1>(d, w, src); run<5>(d, w, src); run<2>(d, w, src); run<3>(d, w, src); run<7>(d, w, src);
    run<16>(d, w, src); run<32>(d, w, src); run<0>(d, w, src); run<16 + 7>(d, w, src); run<32 + 7>(d, w, src); run<7>(d, w, src);
    // sustained: 100 launches (~160 ms) of the bare and of the full loop, twice
    run<0>(d, w, src, 100); run<7>(d, w, src, 100); run<16>(d, w, src, 100); run<0>(d, w, src, 100); run<7>(d, w, src, 100);
    return 0;
}
