// Does straight-line code larger than the instruction cache cost MFMA issue slots?  The same 2-chain MFMA
// stream as a small loop (fits the I-cache) and fully unrolled to N MFMAs per loop trip (8 B each).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int UNROLL>
__global__ void __launch_bounds__(256) k(unsigned long long *out, const float *w, int iters) {
    f32x16 acc0, acc1;
    float b[32];
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
    for (int r = 0; r < 32; ++r) b[r] = w[threadIdx.x * 32 + r];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < UNROLL; u += 2) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[u % 32], b[(u + 7) % 32], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[(u + 1) % 32], b[(u + 9) % 32], acc1, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (s == 12345.678f) out[0] = 1;
    if ((threadIdx.x & 63) == 0) out[1 + blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int UNROLL>
void run(unsigned long long *d, const float *w, int blocks) {
    const int iters = (1 << 20) / UNROLL;
    hipLaunchKernelGGL((k<UNROLL>), dim3(blocks), dim3(256), 0, 0, d, w, iters);
    hipLaunchKernelGGL((k<UNROLL>), dim3(blocks), dim3(256), 0, 0, d, w, iters);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1 + blocks * 4);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int i = 0; i < blocks * 4; ++i) c.push_back((double)h[1 + i] / ((double)iters * UNROLL));
    std::sort(c.begin(), c.end());
    printf("{\"unrolled_mfmas\": %d, \"code_KB\": %d, \"cycles_per_mfma_median\": %.3f, \"p10\": %.3f, \"p90\": %.3f}\n", UNROLL,
           UNROLL * 8 / 1024, c[c.size() / 2], c[c.size() / 10], c[c.size() * 9 / 10]);
}

int main() {
    unsigned long long *d;
    float *w;
    const int blocks = 256;
    hipMalloc(&d, (1 + blocks * 4) * 8);
    hipMemset(d, 0, (1 + blocks * 4) * 8);
    std::vector<float> hw(1 << 16);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
    hipMalloc(&w, hw.size() * 4);
    hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice);
    run<64>(d, w, blocks);
    run<2048>(d, w, blocks);
    run<4096>(d, w, blocks);
    run<8192>(d, w, blocks);
    run<16384>(d, w, blocks);
    return 0;
}
