// What v_permlane32_swap_b32 does, observed: prints for lanes 0, 1, 32, 33 the two results of
// __builtin_amdgcn_permlane32_swap(a, b) with a = lane, b = 100 + lane.  (Used by csrc/plane_sink.h.)
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *out) {
    const unsigned lane = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(lane, 100u + lane, false, false);
    out[lane] = r[0];
    out[64 + lane] = r[1];
}
int main() {
    unsigned *d, h[128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l : {0, 1, 31, 32, 33, 63}) printf("lane %2d: r0 = %3u  r1 = %3u\n", l, h[l], h[64 + l]);
    return 0;
}
