// a4: (axis-angle, translation) -> 4x4 camera motion, forward and backward.
// Restates transformation_from_parameters / rot_from_axisangle / get_translation_matrix
// (reference vo/learner_func.py:29-104) with the reference's operation order.  One lane per batch
// element: the op is 60 flops per element, it exists as a kernel only to remove ~30 tiny eager
// launches from the step.
#include "common.h"

namespace {

__global__ void pose_to_mat_fwd_kernel(const float* __restrict__ aa, const float* __restrict__ tr,
                                       int invert, float* __restrict__ M, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float vx = aa[b * 3 + 0], vy = aa[b * 3 + 1], vz = aa[b * 3 + 2];
    float angle = sqrtf(vx * vx + vy * vy + vz * vz);
    float inv = angle + 1e-7f;
    float x = vx / inv, y = vy / inv, z = vz / inv;
    float ca = cosf(angle), sa = sinf(angle);
    float C = 1.0f - ca;
    float xs = x * sa, ys = y * sa, zs = z * sa;
    float xC = x * C, yC = y * C, zC = z * C;
    float xyC = x * yC, yzC = y * zC, zxC = z * xC;
    float R[3][3] = {{x * xC + ca, xyC - zs, zxC + ys},
                     {xyC + zs, y * yC + ca, yzC - xs},
                     {zxC - ys, yzC + xs, z * zC + ca}};
    float t[3] = {tr[b * 3 + 0], tr[b * 3 + 1], tr[b * 3 + 2]};
    float* m = M + b * 16;
    if (invert) {
        // M = R^T . T(-t)
        for (int i = 0; i < 3; ++i) {
            float acc = 0.f;
            for (int k = 0; k < 3; ++k) {
                m[i * 4 + k] = R[k][i];
                acc += R[k][i] * (-t[k]);
            }
            m[i * 4 + 3] = acc;
        }
    } else {
        // M = T(t) . R
        for (int i = 0; i < 3; ++i) {
            for (int k = 0; k < 3; ++k) m[i * 4 + k] = R[i][k];
            m[i * 4 + 3] = t[i];
        }
    }
    m[12] = 0.f;
    m[13] = 0.f;
    m[14] = 0.f;
    m[15] = 1.f;
}

__global__ void pose_to_mat_bwd_kernel(const float* __restrict__ aa, const float* __restrict__ tr,
                                       int invert, const float* __restrict__ dM,
                                       float* __restrict__ d_aa, float* __restrict__ d_tr, int B) {
    int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float vx = aa[b * 3 + 0], vy = aa[b * 3 + 1], vz = aa[b * 3 + 2];
    float angle = sqrtf(vx * vx + vy * vy + vz * vz);
    float inv = angle + 1e-7f;
    float x = vx / inv, y = vy / inv, z = vz / inv;
    float ca = cosf(angle), sa = sinf(angle);
    float C = 1.0f - ca;
    float t[3] = {tr[b * 3 + 0], tr[b * 3 + 1], tr[b * 3 + 2]};
    const float* g = dM + b * 16;
    float dR[3][3], dt[3];
    if (invert) {
        float xC = x * C, yC = y * C, zC = z * C;
        float xs = x * sa, ys = y * sa, zs = z * sa;
        float R[3][3] = {{x * xC + ca, x * yC - zs, z * xC + ys},
                         {x * yC + zs, y * yC + ca, y * zC - xs},
                         {z * xC - ys, y * zC + xs, z * zC + ca}};
        for (int k = 0; k < 3; ++k) {
            float acc = 0.f;
            for (int i = 0; i < 3; ++i) {
                dR[k][i] = g[i * 4 + k] + g[i * 4 + 3] * (-t[k]);
                acc += R[k][i] * g[i * 4 + 3];
            }
            dt[k] = -acc;
        }
    } else {
        for (int i = 0; i < 3; ++i) {
            for (int k = 0; k < 3; ++k) dR[i][k] = g[i * 4 + k];
            dt[i] = g[i * 4 + 3];
        }
    }
    float s01 = dR[0][1] + dR[1][0], s02 = dR[0][2] + dR[2][0], s12 = dR[1][2] + dR[2][1];
    float a21 = dR[2][1] - dR[1][2], a02 = dR[0][2] - dR[2][0], a10 = dR[1][0] - dR[0][1];
    float dx = dR[0][0] * 2.f * x * C + s01 * y * C + s02 * z * C + a21 * sa;
    float dy = dR[1][1] * 2.f * y * C + s01 * x * C + s12 * z * C + a02 * sa;
    float dz = dR[2][2] * 2.f * z * C + s02 * x * C + s12 * y * C + a10 * sa;
    float dC = dR[0][0] * x * x + dR[1][1] * y * y + dR[2][2] * z * z + s01 * x * y + s02 * z * x + s12 * y * z;
    float dca = dR[0][0] + dR[1][1] + dR[2][2] - dC;
    float dsa = a21 * x + a02 * y + a10 * z;
    float dangle = -sa * dca + ca * dsa - (dx * vx + dy * vy + dz * vz) / (inv * inv);
    // torch.norm backward: v/|v|, with the subgradient 0 at |v| = 0
    float rn = angle > 0.f ? dangle / angle : 0.f;
    d_aa[b * 3 + 0] = dx / inv + rn * vx;
    d_aa[b * 3 + 1] = dy / inv + rn * vy;
    d_aa[b * 3 + 2] = dz / inv + rn * vz;
    d_tr[b * 3 + 0] = dt[0];
    d_tr[b * 3 + 1] = dt[1];
    d_tr[b * 3 + 2] = dt[2];
}

}  // namespace

extern "C" {

int dvs_pose_to_mat_fwd(const float* axisangle, const float* translation, int invert, float* M,
                        int B, void* stream) {
    DVS_REQUIRE(axisangle && translation && M, "dvs_pose_to_mat_fwd: null pointer");
    DVS_REQUIRE(B > 0, "dvs_pose_to_mat_fwd: B=%d", B);
    hipLaunchKernelGGL(pose_to_mat_fwd_kernel, dim3((B + 63) / 64), dim3(64), 0,
                       static_cast<hipStream_t>(stream), axisangle, translation, invert, M, B);
    return dvs::check_launch("dvs_pose_to_mat_fwd");
}

int dvs_pose_to_mat_bwd(const float* axisangle, const float* translation, int invert,
                        const float* dM, float* d_axisangle, float* d_translation, int B,
                        void* stream) {
    DVS_REQUIRE(axisangle && translation && dM && d_axisangle && d_translation,
                "dvs_pose_to_mat_bwd: null pointer");
    DVS_REQUIRE(B > 0, "dvs_pose_to_mat_bwd: B=%d", B);
    hipLaunchKernelGGL(pose_to_mat_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0,
                       static_cast<hipStream_t>(stream), axisangle, translation, invert, dM,
                       d_axisangle, d_translation, B);
    return dvs::check_launch("dvs_pose_to_mat_bwd");
}

}  // extern "C"
