// depth -> camera-space points -> per-pixel least-squares plane "normal"
// (DS_NeRF/run.py:1909-1940: depth2xyz_torch, depth2normal_geo).
//
// The reference unfolds a 31x31 zero-padded window per pixel (a [1, 2883, H*W] tensor, 1.6 GB at
// 283x504), forms A^T A with a batched matmul and inverts it.  Algebraically
//        n = (A^T A)^-1 A^T 1,   A^T A = window sums of (xx, xy, xz, yy, yz, zz),  A^T 1 = (x, y, z)
// so this file computes nine zero-padded box sums (separable: row pass, column pass; fp64
// accumulators, fp32 storage) and solves the symmetric 3x3 system by cofactors in fp64.  Memory:
// 9 floats per pixel instead of 2883.  HBM-bound and tiny: 4 B in + 12 B out per pixel plus the
// 2 x 36 B/pixel moment planes.
//
// Backward (the normal map feeds the normal-SDS term, run.py:960-965, so d/d depth is needed):
//        lambda = M^-1 g;   dL/db = lambda;   dL/dM = -(lambda n^T + n lambda^T) (off-diagonal), -lambda_i n_i
// box-filter adjoint = the same zero-padded box filter; then the chain rule through the moments.
#include "common.h"

namespace mvip {

__device__ __forceinline__ void moments9(float x, float y, float z, float m[9]) {
    m[0] = x * x; m[1] = x * y; m[2] = x * z; m[3] = y * y; m[4] = y * z; m[5] = z * z;
    m[6] = x; m[7] = y; m[8] = z;
}

// x = (w - cx) * z / fx ; y = (h - cy) * z / fy   (run.py:1917-1919, same operation order)
__global__ void depth2xyz_kernel(const float *__restrict__ depth, int H, int W, float fx, float fy, float cx,
                                 float cy, float *__restrict__ pts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int h = i / W, w = i % W;
    const float z = depth[i];
    pts[i * 3 + 0] = ((float)w - cx) * z / fx;
    pts[i * 3 + 1] = ((float)h - cy) * z / fy;
    pts[i * 3 + 2] = z;
}

__global__ void depth2xyz_bwd_kernel(const float *__restrict__ g, int H, int W, float fx, float fy, float cx,
                                     float cy, float *__restrict__ d_depth) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int h = i / W, w = i % W;
    d_depth[i] = g[i * 3 + 0] * ((float)w - cx) / fx + g[i * 3 + 1] * ((float)h - cy) / fy + g[i * 3 + 2];
}

// row pass: out[c][h][w] = sum_{|dw|<=r} in_c(h, w+dw), where in_c are either the nine moments of
// the points (MOMENTS=true, `src` = planar points [3,H,W]) or nine given planes (`src` [9,H,W]).
template <bool MOMENTS>
__global__ void box_rows_kernel(const float *__restrict__ src, int H, int W, int r, float *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int h = i / W, w = i % W;
    const int HW = H * W;
    double acc[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) acc[c] = 0.0;
    const int w0 = max(0, w - r), w1 = min(W - 1, w + r);
    for (int ww = w0; ww <= w1; ++ww) {
        const int q = h * W + ww;
        float m[9];
        if (MOMENTS) moments9(src[q], src[HW + q], src[2 * HW + q], m);
        else {
#pragma unroll
            for (int c = 0; c < 9; ++c) m[c] = src[c * HW + q];
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) acc[c] += (double)m[c];
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) out[c * HW + i] = (float)acc[c];
}

__device__ __forceinline__ void col_sums(const float *__restrict__ rows, int H, int W, int r, int h, int w,
                                         double S[9]) {
    const int HW = H * W;
#pragma unroll
    for (int c = 0; c < 9; ++c) S[c] = 0.0;
    const int h0 = max(0, h - r), h1 = min(H - 1, h + r);
    for (int hh = h0; hh <= h1; ++hh)
#pragma unroll
        for (int c = 0; c < 9; ++c) S[c] += (double)rows[c * HW + hh * W + w];
}

// symmetric 3x3 solve M v = b by cofactors (fp64)
__device__ __forceinline__ void solve_sym3(const double S[6], const double b[3], double v[3]) {
    const double sxx = S[0], sxy = S[1], sxz = S[2], syy = S[3], syz = S[4], szz = S[5];
    const double c00 = syy * szz - syz * syz, c01 = sxz * syz - sxy * szz, c02 = sxy * syz - sxz * syy;
    const double c11 = sxx * szz - sxz * sxz, c12 = sxy * sxz - sxx * syz, c22 = sxx * syy - sxy * sxy;
    const double det = sxx * c00 + sxy * c01 + sxz * c02;
    v[0] = (c00 * b[0] + c01 * b[1] + c02 * b[2]) / det;
    v[1] = (c01 * b[0] + c11 * b[1] + c12 * b[2]) / det;
    v[2] = (c02 * b[0] + c12 * b[1] + c22 * b[2]) / det;
}

__global__ void normal_solve_kernel(const float *__restrict__ rows, int H, int W, int r,
                                    float *__restrict__ moments, float *__restrict__ normals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= H * W) return;
    const int HW = H * W;
    double S[9], n[3];
    col_sums(rows, H, W, r, i / W, i % W, S);
#pragma unroll
    for (int c = 0; c < 9; ++c) moments[c * HW + i] = (float)S[c];
    solve_sym3(S, S + 6, n);
#pragma unroll
    for (int c = 0; c < 3; ++c) normals[c * HW + i] = (float)n[c];
}

// gradient w.r.t. the nine box sums at each pixel
__global__ void normal_bwd_sums_kernel(const float *__restrict__ moments, const float *__restrict__ normals,
                                       const float *__restrict__ g, int H, int W, float *__restrict__ gS) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    if (i >= HW) return;
    double S[6], gv[3], lam[3], n[3];
#pragma unroll
    for (int c = 0; c < 6; ++c) S[c] = (double)moments[c * HW + i];
#pragma unroll
    for (int c = 0; c < 3; ++c) { gv[c] = (double)g[c * HW + i]; n[c] = (double)normals[c * HW + i]; }
    solve_sym3(S, gv, lam);
    gS[0 * HW + i] = (float)(-lam[0] * n[0]);
    gS[1 * HW + i] = (float)(-(lam[0] * n[1] + lam[1] * n[0]));
    gS[2 * HW + i] = (float)(-(lam[0] * n[2] + lam[2] * n[0]));
    gS[3 * HW + i] = (float)(-lam[1] * n[1]);
    gS[4 * HW + i] = (float)(-(lam[1] * n[2] + lam[2] * n[1]));
    gS[5 * HW + i] = (float)(-lam[2] * n[2]);
    gS[6 * HW + i] = (float)lam[0];
    gS[7 * HW + i] = (float)lam[1];
    gS[8 * HW + i] = (float)lam[2];
}

__global__ void normal_bwd_points_kernel(const float *__restrict__ rows, const float *__restrict__ pts, int H,
                                         int W, int r, float *__restrict__ d_pts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int HW = H * W;
    if (i >= HW) return;
    double G[9];
    col_sums(rows, H, W, r, i / W, i % W, G);
    const double x = pts[i], y = pts[HW + i], z = pts[2 * HW + i];
    d_pts[i] = (float)(2.0 * x * G[0] + y * G[1] + z * G[2] + G[6]);
    d_pts[HW + i] = (float)(x * G[1] + 2.0 * y * G[3] + z * G[4] + G[7]);
    d_pts[2 * HW + i] = (float)(x * G[2] + y * G[4] + 2.0 * z * G[5] + G[8]);
}

}  // namespace mvip

using namespace mvip;

static inline dim3 grid_for(int n) { return dim3((unsigned)((n + 255) / 256)); }

extern "C" int mvip_depth2xyz(const float *depth, int H, int W, float fx, float fy, float cx, float cy,
                              float *points, void *stream) {
    if (H <= 0 || W <= 0 || !depth || !points) return MVIP_EINVAL;
    hipLaunchKernelGGL(depth2xyz_kernel, grid_for(H * W), dim3(256), 0, as_stream(stream), depth, H, W, fx, fy, cx, cy,
                       points);
    return check_launch();
}

extern "C" int mvip_depth2xyz_backward(const float *g_points, int H, int W, float fx, float fy, float cx, float cy,
                                       float *d_depth, void *stream) {
    if (H <= 0 || W <= 0 || !g_points || !d_depth) return MVIP_EINVAL;
    hipLaunchKernelGGL(depth2xyz_bwd_kernel, grid_for(H * W), dim3(256), 0, as_stream(stream), g_points, H, W, fx, fy,
                       cx, cy, d_depth);
    return check_launch();
}

extern "C" int mvip_normal_fit_forward(const float *points, int H, int W, int k, float *moments, float *scratch,
                                       float *normals, void *stream) {
    if (H <= 0 || W <= 0 || k < 1 || (k & 1) == 0 || !points || !moments || !scratch || !normals) return MVIP_EINVAL;
    hipStream_t s = as_stream(stream);
    hipLaunchKernelGGL(box_rows_kernel<true>, grid_for(H * W), dim3(256), 0, s, points, H, W, (k - 1) / 2, scratch);
    hipLaunchKernelGGL(normal_solve_kernel, grid_for(H * W), dim3(256), 0, s, scratch, H, W, (k - 1) / 2, moments,
                       normals);
    return check_launch();
}

extern "C" int mvip_normal_fit_backward(const float *points, const float *moments, const float *normals,
                                        const float *g_normals, int H, int W, int k, float *scratch, float *d_points,
                                        void *stream) {
    if (H <= 0 || W <= 0 || k < 1 || (k & 1) == 0 || !points || !moments || !normals || !g_normals || !scratch ||
        !d_points) return MVIP_EINVAL;
    hipStream_t s = as_stream(stream);
    float *gS = scratch, *rows = scratch + (int64_t)9 * H * W;
    hipLaunchKernelGGL(normal_bwd_sums_kernel, grid_for(H * W), dim3(256), 0, s, moments, normals, g_normals, H, W, gS);
    hipLaunchKernelGGL(box_rows_kernel<false>, grid_for(H * W), dim3(256), 0, s, gS, H, W, (k - 1) / 2, rows);
    hipLaunchKernelGGL(normal_bwd_points_kernel, grid_for(H * W), dim3(256), 0, s, rows, points, H, W, (k - 1) / 2,
                       d_points);
    return check_launch();
}
