// Standalone (un-fused) forms of the reference's geometry / photometric operators, for callers that
// use model/layers.py piecewise (vo/eval_traj.py, vo/predict.py, MonodepthTrainer.ssim / .project_3d
// attributes).  The training step itself goes through the fused chain in loss_chain.hip; these
// kernels favour clarity: one lane per pixel, coalesced along x, neighbourhoods served by L1/L2.
//
//   dvs_backproject_*  BackprojectDepth.forward   vo/learner_func.py:130-135
//   dvs_project_*      Project3D.forward          vo/learner_func.py:148-159
//   dvs_ssim_*         SSIM.forward               vo/learner_func.py:193-207
//   dvs_smooth_*       get_smooth_loss            vo/learner_func.py:161-174
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr float C1 = 0.0001f, C2 = 0.0009f;

__device__ __forceinline__ int reflect1(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// ------------------------------------------------------------------------------- backproject
__global__ void backproject_fwd_kernel(const float* __restrict__ depth, const float* __restrict__ inv_K,
                                       float* __restrict__ cam, int H, int W) {
    int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, HW = H * W;
    if (i >= HW) return;
    const float* k = inv_K + b * 16;
    float fx = (float)(i % W), fy = (float)(i / W), d = depth[(size_t)b * HW + i];
    float* o = cam + (size_t)b * 4 * HW + i;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float c = fmaf(k[r * 4 + 2], 1.f, fmaf(k[r * 4 + 1], fy, k[r * 4 + 0] * fx));
        o[(size_t)r * HW] = d * c;
    }
    o[(size_t)3 * HW] = 1.f;
}

__global__ void backproject_bwd_kernel(const float* __restrict__ d_cam, const float* __restrict__ inv_K,
                                       float* __restrict__ d_depth, int H, int W) {
    int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, HW = H * W;
    if (i >= HW) return;
    const float* k = inv_K + b * 16;
    float fx = (float)(i % W), fy = (float)(i / W), acc = 0.f;
    const float* g = d_cam + (size_t)b * 4 * HW + i;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float c = fmaf(k[r * 4 + 2], 1.f, fmaf(k[r * 4 + 1], fy, k[r * 4 + 0] * fx));
        acc += g[(size_t)r * HW] * c;
    }
    d_depth[(size_t)b * HW + i] = acc;
}

// ----------------------------------------------------------------------------------- project
__device__ __forceinline__ void make_P(const float* K, const float* T, float P[12]) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = K[i * 4 + 0] * T[j];
            acc = fmaf(K[i * 4 + 1], T[4 + j], acc);
            acc = fmaf(K[i * 4 + 2], T[8 + j], acc);
            acc = fmaf(K[i * 4 + 3], T[12 + j], acc);
            P[i * 4 + j] = acc;
        }
}

__global__ void project_fwd_kernel(const float* __restrict__ pts, const float* __restrict__ K,
                                   const float* __restrict__ T, float* __restrict__ grid, int H, int W, float eps) {
    int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, HW = H * W;
    if (i >= HW) return;
    float P[12];
    make_P(K + b * 16, T + b * 16, P);
    const float* p = pts + (size_t)b * 4 * HW + i;
    float x = p[0], y = p[(size_t)HW], z = p[(size_t)2 * HW], w = p[(size_t)3 * HW];
    float c0 = fmaf(P[3], w, fmaf(P[2], z, fmaf(P[1], y, P[0] * x)));
    float c1 = fmaf(P[7], w, fmaf(P[6], z, fmaf(P[5], y, P[4] * x)));
    float c2 = fmaf(P[11], w, fmaf(P[10], z, fmaf(P[9], y, P[8] * x)));
    float den = c2 + eps;
    float u = c0 / den, v = c1 / den;
    grid[((size_t)b * HW + i) * 2 + 0] = (u / (float)(W - 1) - 0.5f) * 2.f;
    grid[((size_t)b * HW + i) * 2 + 1] = (v / (float)(H - 1) - 0.5f) * 2.f;
}

// d_pts per pixel; dP partial sums per workgroup -> partials[b][block][12]
__global__ __launch_bounds__(NT) void project_bwd_kernel(const float* __restrict__ pts, const float* __restrict__ K,
                                                         const float* __restrict__ T, const float* __restrict__ d_grid,
                                                         float* __restrict__ d_pts, float* __restrict__ partials,
                                                         int H, int W, float eps) {
    __shared__ float sRed[NT / 64][12];
    int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, HW = H * W;
    float P[12], dP[12];
    make_P(K + b * 16, T + b * 16, P);
#pragma unroll
    for (int j = 0; j < 12; ++j) dP[j] = 0.f;
    if (i < HW) {
        const float* p = pts + (size_t)b * 4 * HW + i;
        float q[4] = {p[0], p[(size_t)HW], p[(size_t)2 * HW], p[(size_t)3 * HW]};
        float c0 = fmaf(P[3], q[3], fmaf(P[2], q[2], fmaf(P[1], q[1], P[0] * q[0])));
        float c1 = fmaf(P[7], q[3], fmaf(P[6], q[2], fmaf(P[5], q[1], P[4] * q[0])));
        float c2 = fmaf(P[11], q[3], fmaf(P[10], q[2], fmaf(P[9], q[1], P[8] * q[0])));
        float den = c2 + eps, u = c0 / den, v = c1 / den;
        float du = d_grid[((size_t)b * HW + i) * 2 + 0] * 2.f / (float)(W - 1);
        float dv = d_grid[((size_t)b * HW + i) * 2 + 1] * 2.f / (float)(H - 1);
        float dc[3] = {du / den, dv / den, -(du * u + dv * v) / den};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) dP[r * 4 + j] = dc[r] * q[j];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            d_pts[(size_t)b * 4 * HW + (size_t)j * HW + i] = P[j] * dc[0] + P[4 + j] * dc[1] + P[8 + j] * dc[2];
    }
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        float s = dvs::wave_sum(dP[j]);
        if (lane == 0) sRed[wave][j] = s;
    }
    __syncthreads();
    if (threadIdx.x < 12)
        partials[((size_t)b * gridDim.x + blockIdx.x) * 12 + threadIdx.x] =
            sRed[0][threadIdx.x] + sRed[1][threadIdx.x] + sRed[2][threadIdx.x] + sRed[3][threadIdx.x];
}

// d_T[b][k][j] = sum_{i<3} K[i][k] dP[i][j]   (K gets no gradient: intrinsics are data)
__global__ void project_bwd_reduce_kernel(const float* __restrict__ partials, const float* __restrict__ K,
                                          float* __restrict__ d_T, int nblocks) {
    __shared__ float sP[12];
    int b = blockIdx.x, t = threadIdx.x;
    if (t < 12) {
        float acc = 0.f;
        for (int r = 0; r < nblocks; ++r) acc += partials[((size_t)b * nblocks + r) * 12 + t];
        sP[t] = acc;
    }
    __syncthreads();
    if (t < 16) {
        int k = t >> 2, j = t & 3;
        float acc = 0.f;
        for (int i = 0; i < 3; ++i) acc += K[b * 16 + i * 4 + k] * sP[i * 4 + j];
        d_T[b * 16 + t] = acc;
    }
}

// -------------------------------------------------------------------------------------- SSIM
struct Stats {
    float mux, muy, sigx, sigy, sigxy;
};

__device__ __forceinline__ Stats stats_at(const float* __restrict__ x, const float* __restrict__ y, int H, int W,
                                          int py, int px) {
    float sx = 0.f, sy = 0.f, sxx = 0.f, syy = 0.f, sxy = 0.f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
        int yy = reflect1(py + dy, H);
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            int xx = reflect1(px + dx, W);
            float a = x[yy * W + xx], b = y[yy * W + xx];
            sx += a; sy += b; sxx += a * a; syy += b * b; sxy += a * b;
        }
    }
    Stats s;
    s.mux = sx / 9.f;
    s.muy = sy / 9.f;
    s.sigx = sxx / 9.f - s.mux * s.mux;
    s.sigy = syy / 9.f - s.muy * s.muy;
    s.sigxy = sxy / 9.f - s.mux * s.muy;
    return s;
}

__global__ void ssim_fwd_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                                int H, int W) {
    int plane = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, HW = H * W;
    if (i >= HW) return;
    Stats s = stats_at(x + (size_t)plane * HW, y + (size_t)plane * HW, H, W, i / W, i % W);
    float n = (2.f * s.mux * s.muy + C1) * (2.f * s.sigxy + C2);
    float d = (s.mux * s.mux + s.muy * s.muy + C1) * (s.sigx + s.sigy + C2);
    out[(size_t)plane * HW + i] = fminf(fmaxf((1.f - n / d) * 0.5f, 0.f), 1.f);
}

// gradient wrt x and y at pixel q: sum over window centres p whose (reflect-padded) window contains q
__global__ void ssim_bwd_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ go,
                                float* __restrict__ dx_out, float* __restrict__ dy_out, int H, int W) {
    int plane = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, HW = H * W;
    if (i >= HW) return;
    const float* xp = x + (size_t)plane * HW;
    const float* yp = y + (size_t)plane * HW;
    const float* gp = go + (size_t)plane * HW;
    int qy = i / W, qx = i % W;
    float xq = xp[i], yq = yp[i];
    float gx = 0.f, gy = 0.f;
#pragma unroll 1
    for (int dy = -1; dy <= 1; ++dy) {
        int py = qy + dy;
        if (py < 0 || py >= H) continue;
        float wy = 1.f + ((qy == 1 && dy == -1) ? 1.f : 0.f) + ((qy == H - 2 && dy == 1) ? 1.f : 0.f);
#pragma unroll 1
        for (int dx = -1; dx <= 1; ++dx) {
            int px = qx + dx;
            if (px < 0 || px >= W) continue;
            float wx = 1.f + ((qx == 1 && dx == -1) ? 1.f : 0.f) + ((qx == W - 2 && dx == 1) ? 1.f : 0.f);
            Stats s = stats_at(xp, yp, H, W, py, px);
            float n1 = 2.f * s.mux * s.muy + C1, n2 = 2.f * s.sigxy + C2;
            float d1 = s.mux * s.mux + s.muy * s.muy + C1, d2 = s.sigx + s.sigy + C2;
            float d = d1 * d2, R = (n1 * n2) / d, v = (1.f - R) * 0.5f;
            if (!(v >= 0.f && v <= 1.f)) continue;
            float g = gp[py * W + px] * wx * wy * (-0.5f) / 9.f;
            float dR_dsig = -R / d2, dR_dsxy = 2.f * n1 / d;
            float dR_dmux = 2.f * s.muy * n2 / d - R * 2.f * s.mux / d1;
            float dR_dmuy = 2.f * s.mux * n2 / d - R * 2.f * s.muy / d1;
            gx += g * (dR_dmux + dR_dsig * (2.f * xq - 2.f * s.mux) + dR_dsxy * (yq - s.muy));
            gy += g * (dR_dmuy + dR_dsig * (2.f * yq - 2.f * s.muy) + dR_dsxy * (xq - s.mux));
        }
    }
    if (dx_out) dx_out[(size_t)plane * HW + i] = gx;
    if (dy_out) dy_out[(size_t)plane * HW + i] = gy;
}

// -------------------------------------------------------------------------------- smoothness
// partials[block][2] = {sum |dx disp| e^{-mean_c |dx img|}, sum |dy disp| e^{-mean_c |dy img|}}
__global__ __launch_bounds__(NT) void smooth_fwd_kernel(const float* __restrict__ disp, const float* __restrict__ img,
                                                        float* __restrict__ partials, int C, int H, int W) {
    __shared__ float sRed[NT / 64][2];
    int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, HW = H * W;
    float ax = 0.f, ay = 0.f;
    if (i < HW) {
        int y = i / W, x = i % W;
        const float* d = disp + (size_t)b * HW;
        const float* im = img + (size_t)b * C * HW;
        if (x < W - 1) {
            float gi = 0.f;
            for (int c = 0; c < C; ++c) gi += fabsf(im[(size_t)c * HW + i] - im[(size_t)c * HW + i + 1]);
            ax = fabsf(d[i] - d[i + 1]) * expf(-gi / (float)C);
        }
        if (y < H - 1) {
            float gi = 0.f;
            for (int c = 0; c < C; ++c) gi += fabsf(im[(size_t)c * HW + i] - im[(size_t)c * HW + i + W]);
            ay = fabsf(d[i] - d[i + W]) * expf(-gi / (float)C);
        }
    }
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ax = dvs::wave_sum(ax);
    ay = dvs::wave_sum(ay);
    if (lane == 0) { sRed[wave][0] = ax; sRed[wave][1] = ay; }
    __syncthreads();
    if (threadIdx.x < 2)
        partials[((size_t)b * gridDim.x + blockIdx.x) * 2 + threadIdx.x] =
            sRed[0][threadIdx.x] + sRed[1][threadIdx.x] + sRed[2][threadIdx.x] + sRed[3][threadIdx.x];
}

__global__ __launch_bounds__(NT) void smooth_reduce_kernel(const float* __restrict__ partials, float* __restrict__ out,
                                                           int n, float cx, float cy) {
    __shared__ float sRed[NT / 64][2];
    float ax = 0.f, ay = 0.f;
    for (int r = threadIdx.x; r < n; r += NT) { ax += partials[r * 2]; ay += partials[r * 2 + 1]; }
    int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    ax = dvs::wave_sum(ax);
    ay = dvs::wave_sum(ay);
    if (lane == 0) { sRed[wave][0] = ax; sRed[wave][1] = ay; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sx = sRed[0][0] + sRed[1][0] + sRed[2][0] + sRed[3][0];
        float sy = sRed[0][1] + sRed[1][1] + sRed[2][1] + sRed[3][1];
        out[0] = sx * cx + sy * cy;
    }
}

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__global__ void smooth_bwd_kernel(const float* __restrict__ disp, const float* __restrict__ img,
                                  const float* __restrict__ g_out, float* __restrict__ d_disp, int C, int H, int W,
                                  float cx, float cy) {
    int b = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x, HW = H * W;
    if (i >= HW) return;
    int y = i / W, x = i % W;
    const float* d = disp + (size_t)b * HW;
    const float* im = img + (size_t)b * C * HW;
    auto wgt = [&](int i0, int i1) {
        float gi = 0.f;
        for (int c = 0; c < C; ++c) gi += fabsf(im[(size_t)c * HW + i0] - im[(size_t)c * HW + i1]);
        return expf(-gi / (float)C);
    };
    float g = 0.f;
    if (x < W - 1) g += cx * sgn(d[i] - d[i + 1]) * wgt(i, i + 1);
    if (x > 0) g -= cx * sgn(d[i - 1] - d[i]) * wgt(i - 1, i);
    if (y < H - 1) g += cy * sgn(d[i] - d[i + W]) * wgt(i, i + W);
    if (y > 0) g -= cy * sgn(d[i - W] - d[i]) * wgt(i - W, i);
    d_disp[(size_t)b * HW + i] = g * g_out[0];
}

inline dim3 pix_grid(int HW, int B) { return dim3((HW + NT - 1) / NT, B); }

}  // namespace

extern "C" {

int dvs_backproject_fwd(const float* depth, const float* inv_K, float* cam_points, int B, int H, int W, void* stream) {
    DVS_REQUIRE(depth && inv_K && cam_points, "dvs_backproject_fwd: null pointer");
    DVS_REQUIRE(B > 0 && H > 0 && W > 0, "dvs_backproject_fwd: bad size");
    hipLaunchKernelGGL(backproject_fwd_kernel, pix_grid(H * W, B), dim3(NT), 0, static_cast<hipStream_t>(stream),
                       depth, inv_K, cam_points, H, W);
    return dvs::check_launch("dvs_backproject_fwd");
}

int dvs_backproject_bwd(const float* d_cam_points, const float* inv_K, float* d_depth, int B, int H, int W,
                        void* stream) {
    DVS_REQUIRE(d_cam_points && inv_K && d_depth, "dvs_backproject_bwd: null pointer");
    DVS_REQUIRE(B > 0 && H > 0 && W > 0, "dvs_backproject_bwd: bad size");
    hipLaunchKernelGGL(backproject_bwd_kernel, pix_grid(H * W, B), dim3(NT), 0, static_cast<hipStream_t>(stream),
                       d_cam_points, inv_K, d_depth, H, W);
    return dvs::check_launch("dvs_backproject_bwd");
}

int dvs_project_fwd(const float* points, const float* K, const float* T, float* grid, int B, int H, int W, float eps,
                    void* stream) {
    DVS_REQUIRE(points && K && T && grid, "dvs_project_fwd: null pointer");
    DVS_REQUIRE(B > 0 && H > 1 && W > 1, "dvs_project_fwd: bad size");
    hipLaunchKernelGGL(project_fwd_kernel, pix_grid(H * W, B), dim3(NT), 0, static_cast<hipStream_t>(stream), points,
                       K, T, grid, H, W, eps);
    return dvs::check_launch("dvs_project_fwd");
}

size_t dvs_project_bwd_workspace(int B, int H, int W) {
    return (size_t)B * ((H * W + NT - 1) / NT) * 12 * sizeof(float);
}

int dvs_project_bwd(const float* points, const float* K, const float* T, const float* d_grid, float* d_points,
                    float* d_T, float* workspace, int B, int H, int W, float eps, void* stream) {
    DVS_REQUIRE(points && K && T && d_grid && d_points && d_T && workspace, "dvs_project_bwd: null pointer");
    DVS_REQUIRE(B > 0 && H > 1 && W > 1, "dvs_project_bwd: bad size");
    dim3 g = pix_grid(H * W, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(project_bwd_kernel, g, dim3(NT), 0, st, points, K, T, d_grid, d_points, workspace, H, W, eps);
    hipLaunchKernelGGL(project_bwd_reduce_kernel, dim3(B), dim3(64), 0, st, workspace, K, d_T, (int)g.x);
    return dvs::check_launch("dvs_project_bwd");
}

int dvs_ssim_fwd(const float* x, const float* y, float* out, int planes, int H, int W, void* stream) {
    DVS_REQUIRE(x && y && out, "dvs_ssim_fwd: null pointer");
    DVS_REQUIRE(planes > 0 && H >= 2 && W >= 2, "dvs_ssim_fwd: bad size (reflection pad needs H,W >= 2)");
    hipLaunchKernelGGL(ssim_fwd_kernel, pix_grid(H * W, planes), dim3(NT), 0, static_cast<hipStream_t>(stream), x, y,
                       out, H, W);
    return dvs::check_launch("dvs_ssim_fwd");
}

int dvs_ssim_bwd(const float* x, const float* y, const float* d_out, float* d_x, float* d_y, int planes, int H, int W,
                 void* stream) {
    DVS_REQUIRE(x && y && d_out && (d_x || d_y), "dvs_ssim_bwd: null pointer");
    DVS_REQUIRE(planes > 0 && H >= 2 && W >= 2, "dvs_ssim_bwd: bad size");
    hipLaunchKernelGGL(ssim_bwd_kernel, pix_grid(H * W, planes), dim3(NT), 0, static_cast<hipStream_t>(stream), x, y,
                       d_out, d_x, d_y, H, W);
    return dvs::check_launch("dvs_ssim_bwd");
}

size_t dvs_smooth_workspace(int B, int H, int W) { return (size_t)B * ((H * W + NT - 1) / NT) * 2 * sizeof(float); }

int dvs_smooth_fwd(const float* disp, const float* img, float* out, float* workspace, int B, int C, int H, int W,
                   void* stream) {
    DVS_REQUIRE(disp && img && out && workspace, "dvs_smooth_fwd: null pointer");
    DVS_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1, "dvs_smooth_fwd: bad size");
    dim3 g = pix_grid(H * W, B);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(smooth_fwd_kernel, g, dim3(NT), 0, st, disp, img, workspace, C, H, W);
    float cx = 1.f / ((float)B * (float)H * (float)(W - 1)), cy = 1.f / ((float)B * (float)(H - 1) * (float)W);
    hipLaunchKernelGGL(smooth_reduce_kernel, dim3(1), dim3(NT), 0, st, workspace, out, (int)(g.x * g.y), cx, cy);
    return dvs::check_launch("dvs_smooth_fwd");
}

int dvs_smooth_bwd(const float* disp, const float* img, const float* d_out, float* d_disp, int B, int C, int H, int W,
                   void* stream) {
    DVS_REQUIRE(disp && img && d_out && d_disp, "dvs_smooth_bwd: null pointer");
    DVS_REQUIRE(B > 0 && C > 0 && H > 1 && W > 1, "dvs_smooth_bwd: bad size");
    float cx = 1.f / ((float)B * (float)H * (float)(W - 1)), cy = 1.f / ((float)B * (float)(H - 1) * (float)W);
    hipLaunchKernelGGL(smooth_bwd_kernel, pix_grid(H * W, B), dim3(NT), 0, static_cast<hipStream_t>(stream), disp, img,
                       d_out, d_disp, C, H, W, cx, cy);
    return dvs::check_launch("dvs_smooth_bwd");
}

}  // extern "C"
