// a1: the ResNet stem's MaxPool2d(kernel_size=3, stride=2, padding=1) (torchvision resnet18 through
// model/resnet_encoder.py:104) on NHWC tensors, forward and backward.
//
// One lane per (output pixel, 4 channels): nine coalesced 16-byte loads, first-maximum-wins in (ky, kx) scan order
// (what torch's kernel does, so ties -- frequent after a ReLU -- route gradients identically), padding = -inf.
// The winning tap is stored as one byte per channel; the backward is then a GATHER over the at most four windows
// that contain an input pixel (no atomics, no zero-fill of dx, every byte written once).
// HBM-bound: forward reads x once and writes y + idx (1.3125 x |y| ... |x| = 4 |y|); backward reads dy + idx, writes dx.
#include "common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int PNT = 256;

__global__ __launch_bounds__(PNT) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                          unsigned* __restrict__ idx, int B, int H, int W, int C, int Ho,
                                                          int Wo) {
    const int cv = C >> 2;
    const size_t n = (size_t)B * Ho * Wo * cv;
    for (size_t i = (size_t)blockIdx.x * PNT + threadIdx.x; i < n; i += (size_t)gridDim.x * PNT) {
        const int c4 = (int)(i % cv);
        size_t pix = i / cv;
        const int ox = (int)(pix % Wo);
        pix /= Wo;
        const int oy = (int)(pix % Ho), b = (int)(pix / Ho);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned bi = 0;                                            // 4 x 8-bit tap numbers
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int iy = 2 * oy - 1 + t / 3, ix = 2 * ox - 1 + t % 3;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)b * H + iy) * W + ix) * C + c4 * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (v[j] > best[j] || v[j] != v[j]) {            // NaN propagates, as in torch
                        best[j] = v[j];
                        bi = (bi & ~(0xffu << (8 * j))) | ((unsigned)t << (8 * j));
                    }
                }
            }
        }
        *reinterpret_cast<f32x4*>(y + i * 4) = best;
        idx[i] = bi;
    }
}

// res (optional): another gradient of the pooled tensor (DepthNet's finest skip connection), added here
__global__ __launch_bounds__(PNT) void maxpool_bwd_kernel(const float* __restrict__ dy, const unsigned* __restrict__ idx,
                                                          float* __restrict__ dx, const float* __restrict__ res, int B, int H,
                                                          int W, int C, int Ho, int Wo) {
    const int cv = C >> 2;
    const size_t n = (size_t)B * H * W * cv;
    for (size_t i = (size_t)blockIdx.x * PNT + threadIdx.x; i < n; i += (size_t)gridDim.x * PNT) {
        const int c4 = (int)(i % cv);
        size_t pix = i / cv;
        const int ix = (int)(pix % W);
        pix /= W;
        const int iy = (int)(pix % H), b = (int)(pix / H);
        f32x4 g = {0.f, 0.f, 0.f, 0.f};
        if (res) g = *reinterpret_cast<const f32x4*>(res + i * 4);
        // windows (oy, ox) with 2*oy - 1 + ky == iy, ky in 0..2
        const int oy_hi = (iy + 1) >> 1, ox_hi = (ix + 1) >> 1;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const int oy = oy_hi - a, ky = iy + 1 - 2 * oy;
            if (oy < 0 || oy >= Ho || ky > 2) continue;
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const int ox = ox_hi - c, kx = ix + 1 - 2 * ox;
                if (ox < 0 || ox >= Wo || kx > 2) continue;
                const size_t o = (((size_t)b * Ho + oy) * Wo + ox) * cv + c4;
                const unsigned w = idx[o];
                const f32x4 d = *reinterpret_cast<const f32x4*>(dy + o * 4);
                const unsigned t = (unsigned)(ky * 3 + kx);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (((w >> (8 * j)) & 0xffu) == t) g[j] += d[j];
            }
        }
        *reinterpret_cast<f32x4*>(dx + i * 4) = g;
    }
}

inline unsigned pool_grid(size_t n) {
    size_t b = (n + PNT - 1) / PNT;
    return (unsigned)(b > 8192 ? 8192 : (b == 0 ? 1 : b));
}


// model/layers.py:196-199 `upsample` as a standalone operator: y[b, 2y+dy, 2x+dx, c] = x[b, y, x, c]; backward = 2x2 sum.
// One lane per (INPUT pixel, 4 channels): one 16-byte load, four 16-byte stores (forward) or the reverse.
__global__ __launch_bounds__(PNT) void upsample2x_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int H,
                                                         int W, int C, int backward) {
    const int cv = C >> 2;
    const size_t n = (size_t)B * H * W * cv;
    for (size_t i = (size_t)blockIdx.x * PNT + threadIdx.x; i < n; i += (size_t)gridDim.x * PNT) {
        const int c4 = (int)(i % cv);
        size_t pix = i / cv;
        const int x = (int)(pix % W);
        pix /= W;
        const int y = (int)(pix % H), b = (int)(pix / H);
        const size_t lo = (((size_t)b * H + y) * W + x) * C + c4 * 4;
        const size_t hi = (((size_t)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C + c4 * 4;
        const size_t row = (size_t)2 * W * C;
        if (!backward) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(src + lo);
            *reinterpret_cast<f32x4*>(dst + hi) = v;
            *reinterpret_cast<f32x4*>(dst + hi + C) = v;
            *reinterpret_cast<f32x4*>(dst + hi + row) = v;
            *reinterpret_cast<f32x4*>(dst + hi + row + C) = v;
        } else {
            const f32x4 a = *reinterpret_cast<const f32x4*>(src + hi), bq = *reinterpret_cast<const f32x4*>(src + hi + C);
            const f32x4 c = *reinterpret_cast<const f32x4*>(src + hi + row), d = *reinterpret_cast<const f32x4*>(src + hi + row + C);
            *reinterpret_cast<f32x4*>(dst + lo) = (a + bq) + (c + d);
        }
    }
}
}  // namespace

extern "C" {

int dvs_maxpool3x3s2_fwd(const float* x, float* y, unsigned char* idx, int B, int H, int W, int C, void* stream) {
    DVS_REQUIRE(x && y && idx && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "dvs_maxpool3x3s2_fwd: bad argument");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t n = (size_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(pool_grid(n)), dim3(PNT), 0, static_cast<hipStream_t>(stream), x, y,
                       reinterpret_cast<unsigned*>(idx), B, H, W, C, Ho, Wo);
    return dvs::check_launch("dvs_maxpool3x3s2_fwd");
}

int dvs_maxpool3x3s2_bwd(const float* dy, const unsigned char* idx, float* dx, int B, int H, int W, int C, void* stream) {
    return dvs_maxpool3x3s2_bwd_res(dy, idx, dx, nullptr, B, H, W, C, stream);
}

int dvs_maxpool3x3s2_bwd_res(const float* dy, const unsigned char* idx, float* dx, const float* residual, int B, int H, int W,
                             int C, void* stream) {
    DVS_REQUIRE(dy && dx && idx && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "dvs_maxpool3x3s2_bwd: bad argument");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t n = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(pool_grid(n)), dim3(PNT), 0, static_cast<hipStream_t>(stream), dy,
                       reinterpret_cast<const unsigned*>(idx), dx, residual, B, H, W, C, Ho, Wo);
    return dvs::check_launch("dvs_maxpool3x3s2_bwd");
}

int dvs_upsample2x_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    DVS_REQUIRE(x && y && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "dvs_upsample2x_fwd: bad argument");
    const size_t n = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(pool_grid(n)), dim3(PNT), 0, static_cast<hipStream_t>(stream), x, y, B, H, W, C, 0);
    return dvs::check_launch("dvs_upsample2x_fwd");
}

int dvs_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream) {
    DVS_REQUIRE(dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "dvs_upsample2x_bwd: bad argument");
    const size_t n = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(upsample2x_kernel, dim3(pool_grid(n)), dim3(PNT), 0, static_cast<hipStream_t>(stream), dy, dx, B, H, W, C, 1);
    return dvs::check_launch("dvs_upsample2x_bwd");
}

}  // extern "C"
