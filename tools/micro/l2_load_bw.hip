// Operand delivery: how many bytes per clock per CU arrive from L2 through (a) plain global_load_dwordx4 into registers and
// (b) global_load_lds_dwordx4 (LDS-DMA), with all 256 CUs pulling 1-KB wave-contiguous pieces from a working set that
// fits the L2s (2 MB, every workgroup walks all of it)?  NW waves per CU, UN loads in flight per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int NW, int UN>
__global__ void __launch_bounds__(NW * 64) k(unsigned long long *out, const char *buf, int iters, int pieces) {
    __shared__ __attribute__((aligned(16))) char lds[NW * UN * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    int pos = (blockIdx.x * 37 + wave * 11) % pieces;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
            u32x4 v[UN];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                v[u] = *reinterpret_cast<const u32x4 *>(buf + (size_t)pos * 1024 + lane * 16);
                pos = pos + NW >= pieces ? pos + NW - pieces : pos + NW;
            }
#pragma unroll
            for (int u = 0; u < UN; ++u) acc ^= v[u];
        } else {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(buf + (size_t)pos * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(lds + (wave * UN + u) * 1024), 16, 0, 0);
                pos = pos + NW >= pieces ? pos + NW - pieces : pos + NW;
            }
            __builtin_amdgcn_s_waitcnt(0);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    if (MODE == 1) acc[0] ^= *reinterpret_cast<const unsigned *>(lds + lane * 4);
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[0] = 1;
    if (lane == 0) out[1 + blockIdx.x * NW + wave] = c1 - c0;
}
template <int MODE, int NW, int UN>
void run(unsigned long long *d, const char *buf) {
    const int blocks = 256, iters = 2000, pieces = 2048;
    hipLaunchKernelGGL((k<MODE, NW, UN>), dim3(blocks), dim3(NW * 64), 0, 0, d, buf, iters, pieces);
    hipLaunchKernelGGL((k<MODE, NW, UN>), dim3(blocks), dim3(NW * 64), 0, 0, d, buf, iters, pieces);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(1 + blocks * NW);
    hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> c;
    for (int i = 0; i < blocks * NW; ++i) c.push_back((double)h[1 + i]);
    std::sort(c.begin(), c.end());
    const double cyc = c[c.size() / 2];
    printf("{\"path\": \"%s\", \"waves_per_cu\": %d, \"loads_in_flight_per_wave\": %d, \"bytes_per_clk_per_cu\": %.1f}\n",
           MODE == 0 ? "global_load_dwordx4 -> VGPR" : "global_load_lds_dwordx4 -> LDS", NW, UN, (double)iters * UN * 1024.0 * NW / cyc);
}
int main() {
    unsigned long long *d; char *buf;
    hipMalloc(&d, (1 + 256 * 16) * 8); hipMemset(d, 0, (1 + 256 * 16) * 8);
    hipMalloc(&buf, 2048 * 1024); hipMemset(buf, 1, 2048 * 1024);
    run<0, 4, 4>(d, buf); run<0, 4, 8>(d, buf); run<0, 8, 8>(d, buf); run<0, 8, 16>(d, buf); run<0, 16, 8>(d, buf);
    run<1, 4, 4>(d, buf); run<1, 4, 8>(d, buf); run<1, 8, 8>(d, buf); run<1, 8, 16>(d, buf);
    return 0;
}
