// Pack the 24 state-dict tensors of the NeRF MLP into the streaming image (mlp_layout.h), and the
// inverse scatter for gradients.  2.4 MB each way, once per optimizer step: HBM-trivial.
#include "common.h"
#include "mlp_layout.h"

namespace mvip {
using namespace mlp;

struct ParamPtrs { float *p[P_COUNT]; };

// packed float index -> (parameter tensor, element offset), or param = -1 for zero padding
__device__ __forceinline__ void packed_source(int idx, int &param, int &off) {
    param = -1; off = 0;
    if (idx < SEC_A_FLOATS) {
        const int blk = idx / BLOCK_FLOATS, r = idx % BLOCK_FLOATS;
        const int h = r / 128, i = (r % 128) / 4, s = r % 4;
        int local, kgn, ti, kg;
        if (blk < OFF_L1) {
            local = blk; kgn = L0_KG; block_tile(local, kgn, ti, kg);
            const int k = 8 * kg + 4 * h + s, row = 32 * ti + i;
            if (k < 63) { param = P_W0; off = row * 63 + k; }
        } else if (blk < OFF_L5 || (blk >= OFF_L6 && blk < OFF_VIEWS)) {
            int layer;
            if (blk < OFF_L5) { layer = 1 + (blk - OFF_L1) / LH_BLOCKS; local = (blk - OFF_L1) % LH_BLOCKS; }
            else if (blk < OFF_FEAT) { layer = 6 + (blk - OFF_L6) / LH_BLOCKS; local = (blk - OFF_L6) % LH_BLOCKS; }
            else { layer = -1; local = blk - OFF_FEAT; }
            block_tile(local, LH_KG, ti, kg);
            const int k = 8 * kg + 4 * h + s, row = 32 * ti + i;
            param = layer >= 0 ? 2 * layer : P_WF;
            off = row * 256 + k;
        } else if (blk < OFF_L6) {
            local = blk - OFF_L5; block_tile(local, L5_KG, ti, kg);
            const int k = 8 * kg + 4 * h + s, row = 32 * ti + i;
            if (k < 63) { param = 2 * 5; off = row * 319 + k; }
            else if (k >= 64) { param = 2 * 5; off = row * 319 + 63 + (k - 64); }
        } else {
            local = blk - OFF_VIEWS; block_tile(local, LV_KG, ti, kg);
            const int k = 8 * kg + 4 * h + s, row = 32 * ti + i;
            if (k < 283) { param = P_WV; off = row * 283 + k; }
        }
        return;
    }
    const int j = idx - SEC_A_FLOATS;
    if (j < SB_BFEAT) { param = 2 * (j / 256) + 1; off = j % 256; }
    else if (j < SB_BVIEWS) { param = P_BF; off = j - SB_BFEAT; }
    else if (j < SB_WALPHA) { param = P_BV; off = j - SB_BVIEWS; }
    else if (j < SB_BALPHA) { param = P_WA; off = j - SB_WALPHA; }
    else if (j == SB_BALPHA) { param = P_BA; off = 0; }
    else if (j >= SB_WRGB && j < SB_BRGB) { param = P_WR; off = j - SB_WRGB; }
    else if (j >= SB_BRGB && j < SB_BRGB + 3) { param = P_BR; off = j - SB_BRGB; }
}

__global__ void mlp_pack_kernel(ParamPtrs pp, float *__restrict__ packed) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= PACKED_FLOATS) return;
    int param, off;
    packed_source(idx, param, off);
    packed[idx] = param >= 0 ? pp.p[param][off] : 0.f;
}

__global__ void mlp_unpack_grads_kernel(const float *__restrict__ gpacked, ParamPtrs gp, int accumulate) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= PACKED_FLOATS) return;
    int param, off;
    packed_source(idx, param, off);
    if (param < 0) return;
    float *dst = gp.p[param] + off;                 // the mapping is injective: no races
    *dst = accumulate ? (*dst + gpacked[idx]) : gpacked[idx];
}

}  // namespace mvip

using namespace mvip;

extern "C" int64_t mvip_mlp_packed_floats(void) { return mlp::PACKED_FLOATS; }

extern "C" int mvip_mlp_pack(const float *const *params_host, float *packed, void *stream) {
    if (!params_host || !packed) return MVIP_EINVAL;
    ParamPtrs pp;
    for (int i = 0; i < mlp::P_COUNT; ++i) {
        if (!params_host[i]) return MVIP_EINVAL;
        pp.p[i] = const_cast<float *>(params_host[i]);
    }
    hipLaunchKernelGGL(mlp_pack_kernel, dim3((mlp::PACKED_FLOATS + 255) / 256), dim3(256), 0, as_stream(stream),
                       pp, packed);
    return check_launch();
}

extern "C" int mvip_mlp_unpack_grads(const float *grad_packed, float *const *grads_host, int accumulate,
                                     void *stream) {
    if (!grad_packed || !grads_host) return MVIP_EINVAL;
    ParamPtrs gp;
    for (int i = 0; i < mlp::P_COUNT; ++i) {
        if (!grads_host[i]) return MVIP_EINVAL;
        gp.p[i] = grads_host[i];
    }
    hipLaunchKernelGGL(mlp_unpack_grads_kernel, dim3((mlp::PACKED_FLOATS + 255) / 256), dim3(256), 0,
                       as_stream(stream), grad_packed, gp, accumulate);
    return check_launch();
}
