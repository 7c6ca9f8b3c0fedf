// the supervised depth learner's multi-scale loss (depth/depth_learner.py:51-117) for gfx950.
//
//   for each scale s:  pred = F.interpolate(pred_depth_s, (H, W), bilinear, align_corners=False)     (:107)
//                      smooth_s = get_smooth_loss(pred, rgb)     mean-normalised, edge-aware          (:51-73)
//                      silog_s  = sqrt(mean(d^2) - 0.85 mean(d)^2),  d = log(clamp(pred, 1e-6)) - log(gt) over valid   (:75-95)
//
// One forward launch covers every scale of a batch: a 256-thread workgroup walks a 64x4 pixel patch per step, re-derives the
// bilinear upsample from the low-resolution map (never materialised), takes the stencil differences from the three samples
// it needs and reduces {sum pred, Gx, Gy, sum d, sum d^2, n_valid} by wavefront butterfly -> LDS -> one partial row per
// workgroup; a finalize kernel adds the rows in double precision.  HBM-bound by bytes (gt + mask + rgb once per scale);
// the backward recomputes the same samples, forms d loss / d pred per pixel and scatters it through the transposed
// upsample (atomics on the low-resolution map; plain stores at full resolution).
#include "common.h"

namespace {

constexpr int NT = 256;
constexpr int NACC = 6;            // per (scale, image, workgroup): sum pred, Gx, Gy, sum d, sum d^2, n_valid
constexpr int NFIN = 8;            // per (scale, image): mean_clamped, Gx, Gy, raw mean ; per scale (row B): dmean, silog, n, -

struct DLParams {
    dvs_depth_loss_cfg cfg;
    const float* pred[DVS_MAX_SCALES];
    const float* gt;
    const unsigned char* mask;
    const float* rgb;
    float* ws;
    float* out;
    const float* d_out;
    float* d_pred[DVS_MAX_SCALES];
    int blocks_per_image;
};

__device__ __forceinline__ float up_at(const float* __restrict__ d, int hs, int ws, bool same, int W, int X, int Y, float ry,
                                       float rx) {
    if (same) return d[Y * W + X];
    float sy = fmaxf(ry * (Y + 0.5f) - 0.5f, 0.f), sx = fmaxf(rx * (X + 0.5f) - 0.5f, 0.f);
    int y0 = min((int)sy, hs - 1), x0 = min((int)sx, ws - 1);
    int y1 = y0 + (y0 < hs - 1), x1 = x0 + (x0 < ws - 1);
    float ly = sy - y0, lx = sx - x0;
    float v00 = d[y0 * ws + x0], v01 = d[y0 * ws + x1], v10 = d[y1 * ws + x0], v11 = d[y1 * ws + x1];
    return (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

__device__ __forceinline__ float edge_w(const float* __restrict__ img, int HW, int o0, int o1) {
    float g = fabsf(img[o0] - img[o1]) + fabsf(img[HW + o0] - img[HW + o1]) + fabsf(img[2 * HW + o0] - img[2 * HW + o1]);
    return __expf(-g * (1.f / 3.f));
}

// workspace layout: partial rows [S][B][blocks][NACC] floats, then the finalized table [S][B + 1][NFIN] floats
__device__ __forceinline__ float* fin_table(const DLParams& p) {
    return p.ws + (size_t)p.cfg.num_scales * p.cfg.B * p.blocks_per_image * NACC;
}

__global__ __launch_bounds__(NT) void depth_loss_fwd_kernel(DLParams p) {
    __shared__ float sRed[NT / 64][NACC];
    const dvs_depth_loss_cfg& c = p.cfg;
    const int H = c.H, W = c.W, HW = H * W;
    const int b = blockIdx.z, s = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const int hs = c.hs[s], ws = c.ws[s];
    const bool same = hs == H && ws == W;
    const float ry = (float)hs / (float)H, rx = (float)ws / (float)W;
    const float* d = p.pred[s] + (size_t)b * hs * ws;
    const float* gt = p.gt + (size_t)b * HW;
    const unsigned char* mk = p.mask + (size_t)b * HW;
    const float* img = p.rgb + (size_t)b * 3 * HW;
    float a[NACC] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = blk * NT + tid; i < HW; i += gridDim.x * NT) {
        const int Y = i / W, X = i - Y * W;
        const float v = up_at(d, hs, ws, same, W, X, Y, ry, rx);
        a[0] += v;
        if (X < W - 1) a[1] += fabsf(up_at(d, hs, ws, same, W, X + 1, Y, ry, rx) - v) * edge_w(img, HW, i, i + 1);
        if (Y < H - 1) a[2] += fabsf(up_at(d, hs, ws, same, W, X, Y + 1, ry, rx) - v) * edge_w(img, HW, i, i + W);
        if (mk[i]) {
            const float dl = logf(fmaxf(v, 1e-6f)) - logf(gt[i]);
            a[3] += dl;
            a[4] += dl * dl;
            a[5] += 1.f;
        }
    }
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
        float v = dvs::wave_sum(a[j]);
        if ((tid & 63) == 0) sRed[tid >> 6][j] = v;
    }
    __syncthreads();
    if (tid < NACC)
        p.ws[(((size_t)s * c.B + b) * p.blocks_per_image + blk) * NACC + tid] = sRed[0][tid] + sRed[1][tid] + sRed[2][tid] + sRed[3][tid];
}

// one workgroup per scale: double-precision sums of the partial rows, then the two losses of the scale
__global__ __launch_bounds__(NT) void depth_loss_finalize_kernel(DLParams p) {
    __shared__ double sAcc[NT];
    __shared__ double sImg[3];
    const dvs_depth_loss_cfg& c = p.cfg;
    const int s = blockIdx.x, tid = threadIdx.x, nb = p.blocks_per_image;
    const double hw = (double)c.H * (double)c.W;
    float* fin = fin_table(p) + (size_t)s * (c.B + 1) * NFIN;
    double tot[3] = {0.0, 0.0, 0.0}, sx = 0.0, sy = 0.0;
    for (int b = 0; b <= c.B; ++b) {            // b == B: the scale-wide silog sums
        for (int j = 0; j < 3; ++j) {
            double v = 0.0;
            if (b < c.B) {
                for (int r = tid; r < nb; r += NT) v += p.ws[(((size_t)s * c.B + b) * nb + r) * NACC + j];
            } else {
                for (int r = tid; r < nb * c.B; r += NT) v += p.ws[((size_t)s * c.B * nb + r) * NACC + 3 + j];
            }
            sAcc[tid] = v;
            __syncthreads();
            for (int o = NT / 2; o > 0; o >>= 1) {
                if (tid < o) sAcc[tid] += sAcc[tid + o];
                __syncthreads();
            }
            if (tid == 0) sImg[j] = sAcc[0];
            __syncthreads();
        }
        if (b < c.B) {
            const double raw = sImg[0] / hw, mean = raw > 1e-7 ? raw : 1e-7;      // disp.mean().clamp(min=1e-7), :57
            sx += sImg[1] / mean;
            sy += sImg[2] / mean;
            if (tid == 0) {
                fin[b * NFIN + 0] = (float)mean;
                fin[b * NFIN + 1] = (float)sImg[1];
                fin[b * NFIN + 2] = (float)sImg[2];
                fin[b * NFIN + 3] = (float)raw;
            }
        } else {
            tot[0] = sImg[0];
            tot[1] = sImg[1];
            tot[2] = sImg[2];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double n = tot[2], dmean = tot[0] / n, d2 = tot[1] / n;
        const double silog = sqrt(d2 - (double)c.variance_focus * dmean * dmean);
        const double smooth = sx / ((double)c.B * c.H * (c.W - 1)) + sy / ((double)c.B * (c.H - 1) * c.W);
        fin[c.B * NFIN + 0] = (float)dmean;
        fin[c.B * NFIN + 1] = (float)silog;
        fin[c.B * NFIN + 2] = (float)n;
        p.out[s] = (float)silog;
        p.out[c.num_scales + s] = (float)smooth;
    }
}

__device__ __forceinline__ float sgnf(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(NT) void depth_loss_bwd_kernel(DLParams p) {
    const dvs_depth_loss_cfg& c = p.cfg;
    const int H = c.H, W = c.W, HW = H * W;
    const int b = blockIdx.z, s = blockIdx.y, tid = threadIdx.x;
    const int hs = c.hs[s], ws = c.ws[s];
    const bool same = hs == H && ws == W;
    const float ry = (float)hs / (float)H, rx = (float)ws / (float)W;
    const float* d = p.pred[s] + (size_t)b * hs * ws;
    float* dd = p.d_pred[s] + (size_t)b * hs * ws;
    const float* gt = p.gt + (size_t)b * HW;
    const unsigned char* mk = p.mask + (size_t)b * HW;
    const float* img = p.rgb + (size_t)b * 3 * HW;
    const float* fin = fin_table(p) + (size_t)s * (c.B + 1) * NFIN;
    const float g_silog = p.d_out[s], g_smooth = p.d_out[c.num_scales + s];
    const float mean = fin[b * NFIN + 0], Gx = fin[b * NFIN + 1], Gy = fin[b * NFIN + 2], raw = fin[b * NFIN + 3];
    const float dmean = fin[c.B * NFIN + 0], silog = fin[c.B * NFIN + 1], n = fin[c.B * NFIN + 2];
    const float cx = 1.f / ((float)c.B * (float)H * (float)(W - 1)), cy = 1.f / ((float)c.B * (float)(H - 1) * (float)W);
    const float k_grad = g_smooth / mean;
    const float k_mean = (raw > 1e-7f) ? -g_smooth * (cx * Gx + cy * Gy) / (mean * mean) / (float)HW : 0.f;
    // d silog / d dl(q) = (dl / n - vf * dmean / n) / silog
    const float k_sil = g_silog / (silog * n);
    for (int i = blockIdx.x * NT + tid; i < HW; i += gridDim.x * NT) {
        const int Y = i / W, X = i - Y * W;
        const float v = up_at(d, hs, ws, same, W, X, Y, ry, rx);
        float g = 0.f;
        if (X < W - 1) g += cx * sgnf(v - up_at(d, hs, ws, same, W, X + 1, Y, ry, rx)) * edge_w(img, HW, i, i + 1);
        if (X > 0) g -= cx * sgnf(up_at(d, hs, ws, same, W, X - 1, Y, ry, rx) - v) * edge_w(img, HW, i - 1, i);
        if (Y < H - 1) g += cy * sgnf(v - up_at(d, hs, ws, same, W, X, Y + 1, ry, rx)) * edge_w(img, HW, i, i + W);
        if (Y > 0) g -= cy * sgnf(up_at(d, hs, ws, same, W, X, Y - 1, ry, rx) - v) * edge_w(img, HW, i - W, i);
        g = k_grad * g + k_mean;
        if (mk[i] && v > 1e-6f) {
            const float dl = logf(v) - logf(gt[i]);
            g += k_sil * (dl - c.variance_focus * dmean) / v;
        }
        if (same) {
            dd[i] = g;
        } else {
            float sy = fmaxf(ry * (Y + 0.5f) - 0.5f, 0.f), sx = fmaxf(rx * (X + 0.5f) - 0.5f, 0.f);
            int y0 = min((int)sy, hs - 1), x0 = min((int)sx, ws - 1);
            int y1 = y0 + (y0 < hs - 1), x1 = x0 + (x0 < ws - 1);
            float ly = sy - y0, lx = sx - x0;
            atomicAdd(&dd[y0 * ws + x0], g * (1.f - ly) * (1.f - lx));
            atomicAdd(&dd[y0 * ws + x1], g * (1.f - ly) * lx);
            atomicAdd(&dd[y1 * ws + x0], g * ly * (1.f - lx));
            atomicAdd(&dd[y1 * ws + x1], g * ly * lx);
        }
    }
}

int validate(const dvs_depth_loss_cfg* c, const char* who) {
    DVS_REQUIRE(c, "%s: null cfg", who);
    DVS_REQUIRE(c->B > 0 && c->H >= 2 && c->W >= 2, "%s: bad size B=%d H=%d W=%d", who, c->B, c->H, c->W);
    DVS_REQUIRE(c->num_scales >= 1 && c->num_scales <= DVS_MAX_SCALES, "%s: num_scales=%d", who, c->num_scales);
    for (int s = 0; s < c->num_scales; ++s) {
        DVS_REQUIRE(c->hs[s] > 0 && c->ws[s] > 0 && c->hs[s] <= c->H && c->ws[s] <= c->W, "%s: scale %d is %dx%d", who, s,
                    c->hs[s], c->ws[s]);
    }
    return DVS_OK;
}

int blocks_per_image(const dvs_depth_loss_cfg* c) {
    long n = ((long)c->H * c->W + NT * 4 - 1) / (NT * 4);      // ~4 pixels per thread
    return (int)(n < 1 ? 1 : (n > 512 ? 512 : n));
}

DLParams make_params(const dvs_depth_loss_cfg* cfg, const float* const* pred, const float* gt, const unsigned char* mask,
                     const float* rgb, float* ws) {
    DLParams p{};
    p.cfg = *cfg;
    for (int s = 0; s < cfg->num_scales; ++s) p.pred[s] = pred[s];
    p.gt = gt;
    p.mask = mask;
    p.rgb = rgb;
    p.ws = ws;
    p.blocks_per_image = blocks_per_image(cfg);
    return p;
}

}  // namespace

extern "C" {

size_t dvs_depth_loss_workspace(const dvs_depth_loss_cfg* cfg) {
    if (validate(cfg, "dvs_depth_loss_workspace")) return 0;
    return ((size_t)cfg->num_scales * cfg->B * blocks_per_image(cfg) * NACC + (size_t)cfg->num_scales * (cfg->B + 1) * NFIN) * sizeof(float);
}

int dvs_depth_loss_fwd(const dvs_depth_loss_cfg* cfg, const float* const* pred_depth, const float* gt_depth,
                       const unsigned char* valid_mask, const float* rgb, float* workspace, float* out, void* stream) {
    int rc = validate(cfg, "dvs_depth_loss_fwd");
    if (rc) return rc;
    DVS_REQUIRE(pred_depth && gt_depth && valid_mask && rgb && workspace && out, "dvs_depth_loss_fwd: null argument");
    for (int s = 0; s < cfg->num_scales; ++s) DVS_REQUIRE(pred_depth[s], "dvs_depth_loss_fwd: null pred_depth[%d]", s);
    DLParams p = make_params(cfg, pred_depth, gt_depth, valid_mask, rgb, workspace);
    p.out = out;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(depth_loss_fwd_kernel, dim3(p.blocks_per_image, cfg->num_scales, cfg->B), dim3(NT), 0, st, p);
    hipLaunchKernelGGL(depth_loss_finalize_kernel, dim3(cfg->num_scales), dim3(NT), 0, st, p);
    return dvs::check_launch("dvs_depth_loss_fwd");
}

int dvs_depth_loss_bwd(const dvs_depth_loss_cfg* cfg, const float* const* pred_depth, const float* gt_depth,
                       const unsigned char* valid_mask, const float* rgb, float* workspace, const float* d_out,
                       float* const* d_pred_depth, void* stream) {
    int rc = validate(cfg, "dvs_depth_loss_bwd");
    if (rc) return rc;
    DVS_REQUIRE(pred_depth && gt_depth && valid_mask && rgb && workspace && d_out && d_pred_depth, "dvs_depth_loss_bwd: null argument");
    DLParams p = make_params(cfg, pred_depth, gt_depth, valid_mask, rgb, workspace);
    p.d_out = d_out;
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int s = 0; s < cfg->num_scales; ++s) {
        DVS_REQUIRE(pred_depth[s] && d_pred_depth[s], "dvs_depth_loss_bwd: null pred_depth / d_pred_depth[%d]", s);
        p.d_pred[s] = d_pred_depth[s];
        if (!(cfg->hs[s] == cfg->H && cfg->ws[s] == cfg->W)) {
            hipError_t e = hipMemsetAsync(d_pred_depth[s], 0, (size_t)cfg->B * cfg->hs[s] * cfg->ws[s] * sizeof(float), st);
            if (e != hipSuccess) return dvs::fail(DVS_ERR_LAUNCH, "dvs_depth_loss_bwd: memset: %s", hipGetErrorString(e));
        }
    }
    hipLaunchKernelGGL(depth_loss_bwd_kernel, dim3(p.blocks_per_image, cfg->num_scales, cfg->B), dim3(NT), 0, st, p);
    return dvs::check_launch("dvs_depth_loss_bwd");
}

}  // extern "C"
