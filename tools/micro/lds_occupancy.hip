// How many 256-thread workgroups with B bytes of static LDS does a gfx950 CU hold?  (The split-precision convolution
// kernel uses 80,384 B and counts on two.)  Answer from the runtime's occupancy calculator AND from a timing probe: W
// workgroups per CU each spinning for a fixed number of cycles finish in one round if they are co-resident.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int B>
__global__ void __launch_bounds__(256) k(unsigned long long *out, int spin) {
    __shared__ char lds[B];
    lds[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)spin) { }
    if (threadIdx.x == 0) out[blockIdx.x] = lds[5];
}
template <int B>
void probe(unsigned long long *d) {
    int n = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, k<B>, 256, 0);
    float ms[3];
    for (int w = 1; w <= 3; ++w) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<B>, dim3(256 * w), dim3(256), 0, 0, d, 2000000);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<B>, dim3(256 * w), dim3(256), 0, 0, d, 2000000);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms[w - 1], e0, e1);
    }
    printf("{\"lds_bytes\": %d, \"occupancy_api_blocks_per_cu\": %d, \"ms_1_2_3_wg_per_cu\": [%.2f, %.2f, %.2f]}\n", B, n, ms[0], ms[1], ms[2]);
}
int main() {
    unsigned long long *d; hipMalloc(&d, 8 * 1024);
    probe<61952>(d); probe<73728>(d); probe<80384>(d); probe<81920>(d); probe<83968>(d);
    return 0;
}
