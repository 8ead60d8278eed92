// Issue-rate microbenchmark: cycles per v_mfma_f32_32x32x2_f32 (and per v_mfma_f32_32x32x16_f16) for a
// dependent chain and for 2 / 4 independent accumulator chains, one wave per SIMD, every CU busy.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip ; run: ./mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));

template <int CHAINS, bool F16, bool RANDOM = false>
__global__ void __launch_bounds__(256) k(unsigned long long *out, float seed, int iters) {
    f32x16 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = seed * (c + r);
    float a = seed + threadIdx.x, b = seed * 2.f + threadIdx.x;
    h16x8 ah, bh;
    for (int i = 0; i < 8; ++i) { ah[i] = (_Float16)(seed + i); bh[i] = (_Float16)(seed - i); }
    float av[16], bv[16];                     // RANDOM: 16 different full-entropy operand pairs per lane (register-resident)
    unsigned rs = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x * 7919u;
    for (int i = 0; i < 16; ++i) {
        rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; av[i] = __uint_as_float((rs & 0x007fffffu) | 0x3f800000u) - 1.5f;
        rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5; bv[i] = __uint_as_float((rs & 0x007fffffu) | 0x3f800000u) - 1.5f;
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            f32x16 &x = acc[u % CHAINS];
            if constexpr (F16) x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, x, 0, 0, 0);
            else x = __builtin_amdgcn_mfma_f32_32x32x2f32(RANDOM ? av[u] : a, RANDOM ? bv[u] : b, x, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = 1;                         // keep the accumulators alive
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CHAINS, bool F16, bool RANDOM = false>
void run(const char *name, unsigned long long *d, int blocks) {
    const int iters = RANDOM ? 200000 : 2000;     // RANDOM runs ~1 s so that power management reaches steady state
    hipLaunchKernelGGL((k<CHAINS, F16, RANDOM>), dim3(blocks), dim3(256), 0, 0, d, 1.0f, iters);
    hipLaunchKernelGGL((k<CHAINS, F16, RANDOM>), dim3(blocks), dim3(256), 0, 0, d, 1.0f, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1 + blocks * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int i = 0; i < blocks * 4; ++i) c.push_back((double)h[1 + i] / (iters * 16.0));
    std::sort(c.begin(), c.end());
    printf("{\"kernel\": \"%s\", \"chains\": %d, \"cycles_per_mfma_median\": %.3f, \"p10\": %.3f, \"p90\": %.3f}\n", name, CHAINS,
           c[c.size() / 2], c[c.size() / 10], c[c.size() * 9 / 10]);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CHAINS, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k16(unsigned long long *out, float seed, int iters) {
    f32x4 acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) acc[c][r] = seed * (c + r);
    float a = seed + threadIdx.x, b = seed * 2.f + threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[u % CHAINS], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 4; ++r) s += acc[c][r];
    if (s == 12345.678f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int CHAINS, int WAVES>
void run16(unsigned long long *d, int blocks) {
    const int iters = 2000;
    hipLaunchKernelGGL((k16<CHAINS, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, d, 1.0f, iters);
    hipLaunchKernelGGL((k16<CHAINS, WAVES>), dim3(blocks), dim3(WAVES * 64), 0, 0, d, 1.0f, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1 + blocks * 8);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < WAVES; ++w) c.push_back((double)h[1 + b * 8 + w] / (iters * 16.0));
    std::sort(c.begin(), c.end());
    printf("{\"kernel\": \"v_mfma_f32_16x16x4_f32\", \"chains\": %d, \"waves_per_simd\": %d, \"wave_cycles_per_mfma_median\": %.3f, \"pipe_cycles_per_mfma\": %.3f}\n",
           CHAINS, WAVES / 4, c[c.size() / 2], c[c.size() / 2] / (WAVES / 4));
}

int main() {
    unsigned long long *d;
    const int blocks = 256;
    hipMalloc(&d, (1 + blocks * 8) * 8);
    hipMemset(d, 0, (1 + blocks * 8) * 8);
    run<1, false>("v_mfma_f32_32x32x2_f32", d, blocks);
    run<2, false>("v_mfma_f32_32x32x2_f32", d, blocks);
    run<4, false>("v_mfma_f32_32x32x2_f32", d, blocks);
    run<2, false, true>("v_mfma_f32_32x32x2_f32 random operands", d, blocks);
    run<2, false, true>("v_mfma_f32_32x32x2_f32 random operands", d, blocks);
    run<1, true>("v_mfma_f32_32x32x16_f16", d, blocks);
    run<2, true>("v_mfma_f32_32x32x16_f16", d, blocks);
    run<4, true>("v_mfma_f32_32x32x16_f16", d, blocks);
    run16<1, 4>(d, blocks);
    run16<2, 4>(d, blocks);
    run16<1, 8>(d, blocks);
    run16<2, 8>(d, blocks);
    return 0;
}
