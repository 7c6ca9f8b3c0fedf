// a13: Adam over one flat fp32 arena (all trainable tensors of DepthNet + PoseNet live back to back
// in one buffer, their gradients in a second one).  Replaces the ~250 per-tensor launches of
// torch.optim.Adam(lr=1e-4) used by the reference trainer (vo/train.py:114-117,192) with one
// HBM-bound pass: 16 B read + 12 B written per parameter (+4 B when the gradient is zeroed in the
// same pass, replacing optimizer.zero_grad of vo/train.py:175).
#include "common.h"

#include <cmath>

namespace {

using f4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, size_t n,
                                                   float lr, float b1, float b2, float eps, float bc1,
                                                   float bc2_sqrt, float grad_scale, int zero_grad) {
    size_t n4 = n / 4;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    const float step_size = lr / bc1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        f4 pp = reinterpret_cast<f4*>(p)[i], gg = reinterpret_cast<f4*>(g)[i];
        f4 mm = reinterpret_cast<f4*>(m)[i], vv = reinterpret_cast<f4*>(v)[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gk = gg[k] * grad_scale;
            mm[k] = b1 * mm[k] + (1.f - b1) * gk;
            vv[k] = b2 * vv[k] + (1.f - b2) * gk * gk;
            float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
            pp[k] -= step_size * (mm[k] / denom);
        }
        reinterpret_cast<f4*>(p)[i] = pp;
        reinterpret_cast<f4*>(m)[i] = mm;
        reinterpret_cast<f4*>(v)[i] = vv;
        if (zero_grad) reinterpret_cast<f4*>(g)[i] = f4{0.f, 0.f, 0.f, 0.f};
    }
    // tail (n not a multiple of 4)
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float gk = g[i] * grad_scale;
        float mk = b1 * m[i] + (1.f - b1) * gk;
        float vk = b2 * v[i] + (1.f - b2) * gk * gk;
        p[i] -= step_size * (mk / (sqrtf(vk) / bc2_sqrt + eps));
        m[i] = mk;
        v[i] = vk;
        if (zero_grad) g[i] = 0.f;
    }
}

}  // namespace

extern "C" {

int dvs_adam_step(float* param, float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float lr,
                  float beta1, float beta2, float eps, int step, float grad_scale, int zero_grad, void* stream) {
    DVS_REQUIRE(param && grad && exp_avg && exp_avg_sq, "dvs_adam_step: null pointer");
    DVS_REQUIRE(n > 0 && step >= 1, "dvs_adam_step: n=%zu step=%d", n, step);
    DVS_REQUIRE((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                "dvs_adam_step: buffers must be 16-byte aligned");
    // bias corrections in double like torch.optim.Adam's Python scalars
    float bc1 = (float)(1.0 - pow((double)beta1, (double)step));
    float bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
    size_t blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;   // 256 CUs x 8 blocks, grid-stride the rest
    if (blocks == 0) blocks = 1;
    dvs::ProfScope prof(dvs::SLOT_ADAM, static_cast<hipStream_t>(stream));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), param,
                       grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps, bc1, bc2_sqrt, grad_scale, zero_grad);
    return dvs::check_launch("dvs_adam_step");
}

}  // extern "C"
