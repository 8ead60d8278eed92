// Timing harness for csrc/mlp_fwd16_f16x3.hip outside the library: the kernel source is compiled INTO this program with one of the
// -DMVIP_EXPERIMENT_F16W16_* macros (no barriers / no sin-cos encoding / no hi-lo conversions: results are then wrong, only the
// time means something) and run on the bench's fine-pass launch shape (190,512 rays x 128 samples, random weights and rays).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DMVIP_EXPERIMENT_F16W16_ONE_RING [-DMVIP_EXPERIMENT_F16W16_NO_BARRIER ...] \
//         -o f16w16_<variant> f16w16_variants.hip
#include "../../mvip_nerf_amd/csrc/mlp_fwd16_f16x3.hip"
#include <cstdio>
#include <vector>
namespace mvip { void set_last_error(hipError_t) {} }

int main(int argc, char **argv) {
    const char *label = argc > 1 ? argv[1] : "full";
    const int64_t B = 190512;
    const int S = 128;
    std::vector<float> h(mlp::PACKED_FLOATS);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) & 0xffffff) / 16777216.f - 0.5f; };
    // an image of plausible magnitudes: fp16 hi fragments ~ +-0.05, lo fragments ~ 2^-12 of that; section B small
    _Float16 *hh = reinterpret_cast<_Float16 *>(h.data());
    for (int64_t b = 0; b < mlp::TOTAL_BLOCKS; ++b)
        for (int e = 0; e < 512; ++e) hh[b * 512 + e] = (_Float16)(rnd() * ((b & 1) ? 2.4e-5f : 0.1f));
    for (int i = mlp::SEC_A_FLOATS; i < mlp::PACKED_FLOATS; ++i) h[i] = rnd() * 0.1f;
    // argv[2] = "zero": an all-zero weight image (same instructions, minimal operand toggling): how much of the launch time is the
    // POWER the data costs rather than the schedule
    if (argc > 2 && argv[2][0] == 'z') for (auto &v : h) v = 0.f;
    std::vector<float> rows(B * 11), z(B * S);
    for (int64_t r = 0; r < B; ++r) {
        for (int c = 0; c < 11; ++c) rows[r * 11 + c] = rnd();
        for (int s = 0; s < S; ++s) z[r * S + s] = 1.2f + 6.f * (s + 0.5f) / S;
    }
    float *dimg, *drows, *dz, *draw;
    hipMalloc(&dimg, h.size() * 4); hipMalloc(&drows, rows.size() * 4); hipMalloc(&dz, z.size() * 4); hipMalloc(&draw, (size_t)B * S * 16);
    hipMemcpy(dimg, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(drows, rows.data(), rows.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dz, z.data(), z.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    mvip_mlp_forward_rays_f16x3_w16(dimg, drows, dz, B, S, draw, nullptr);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int k = 0; k < 3; ++k) mvip_mlp_forward_rays_f16x3_w16(dimg, drows, dz, B, S, draw, nullptr);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 3.f;
        printf("{\"variant\": \"%s\", \"launch_ms\": %.3f, \"fp16_product_TFLOPs\": %.1f}\n", label, ms,
               3.0 * B * S * 1186816.0 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
