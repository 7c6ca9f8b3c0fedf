// a5-a12: the fused view-synthesis loss chain (forward + backward) for gfx950.
//
// One launch covers every scale and both source frames of a batch.  A 256-thread workgroup owns a
// 64x16 pixel tile of one image: 64 lanes run along x so every row access is one coalesced 256-B
// segment, and each lane owns a 4-row strip so the 3x3 SSIM window sums are separable (6 row sums
// serve 4 pixels).  The 3x3 windows and the smoothness stencil are served from LDS tiles that carry
// a reflect-padded halo (1 px forward, 2 px backward).  Warped colours are never written to HBM
// unless the caller asks for the reference's `outputs` tensors; the backward recomputes them.
// Reductions: wavefront butterfly (64 lanes) -> LDS -> one partial row per workgroup -> a small
// finalize kernel (deterministic, no float atomics on the loss).
// Arithmetic: fp32; reciprocals use v_rcp_f32 (1 ulp) instead of IEEE division -- the coordinate
// noise this adds (~3e-5 px) is the size of the reference's own fp32 rounding.
//
// Reference arithmetic restated here (file:line under the reference repo):
//   F.interpolate bilinear/align_corners=False   vo/learner_new.py:136-140
//   disp_to_depth                                vo/learner_func.py:16-26
//   BackprojectDepth / Project3D                 vo/learner_func.py:106-159
//   F.grid_sample border / align_corners=True    vo/learner_new.py:165-170
//   SSIM, reprojection loss                      vo/learner_func.py:177-207, vo/learner_new.py:60-74
//   auto-mask min, smoothness, loss assembly     vo/learner_new.py:199-257
#include "common.h"

namespace {

constexpr int TW = 64, TH = 16, NT = 256, PX = 4;  // tile, threads, rows per lane
constexpr int FH = TH + 2, FW = TW + 2;            // forward tile + 1-px halo
constexpr int BH = TH + 4, BW = TW + 4;            // backward tile + 2-px halo
constexpr int ACC_W = TW / 2 + 4, ACC_H = TH / 2 + 4;  // low-res d_disp footprint of a tile (ratio >= 2)
constexpr int NPART = 16;                          // forward partial row: 4 scales x {min, sum disp, Gx, Gy}
constexpr int NDP = 12;                            // dP = d loss / d (K.T)[:3,:4]
constexpr float C1 = 0.0001f, C2 = 0.0009f;        // SSIM constants, learner_func.py:190-191
constexpr float K9 = 1.0f / 9.0f;
#ifndef FWD_WAVES
#define FWD_WAVES 3   // waves per SIMD the register allocator targets (4 and 3 spill: checked with -Rpass-analysis)
#endif
#ifndef BWD_WAVES
#define BWD_WAVES 2
#endif
#ifndef CHAIN_DBG
#define CHAIN_DBG 0   // timing cuts of chain_bwd_kernel (tools/chain_cuts.sh): 1 no gathers in staging, 2 no field phase, 4 no 3x3 gather,
#endif                // 8 no chain phase, 16 no wave reductions -- results are wrong with any bit set
#ifndef BWD_NT
#define BWD_NT 256    // threads per workgroup of chain_bwd_kernel: 256 (4 rows per lane, the default) or 512 (2 rows per lane, four waves per SIMD:
#endif                // measured equal before the owner-thread staging, slower after it -- 128 registers spill)
constexpr int FIELD_ROWS = 6, FIELD_THREADS = FW * (FH / FIELD_ROWS);  // 66 columns x 3 row groups
static_assert(FH % FIELD_ROWS == 0 && FIELD_THREADS <= NT, "field strips must tile the 1-px-halo tile");
static_assert(ACC_H * ACC_W <= 3 * FH * FW, "sAcc aliases sF");

struct ChainParams {
    dvs_chain_cfg cfg;
    dvs_chain_fwd_io io;
    int tiles_x, tiles_y;
};

struct CamMats {
    float iK[9];   // inv_K[:3,:3]
    float P[12];   // (K @ T)[:3,:]
};

__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * n - 2 - i : i;
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ void load_cam(const ChainParams& p, int b, int f, CamMats& m) {
    const float* K = p.io.K + b * 16;
    const float* T = p.io.T[f] + b * 16;
    const float* iK = p.io.inv_K + b * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float acc = K[i * 4 + 0] * T[0 * 4 + j];
            acc = fmaf(K[i * 4 + 1], T[1 * 4 + j], acc);
            acc = fmaf(K[i * 4 + 2], T[2 * 4 + j], acc);
            acc = fmaf(K[i * 4 + 3], T[3 * 4 + j], acc);
            m.P[i * 4 + j] = acc;
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) m.iK[i * 3 + j] = iK[i * 4 + j];
    }
}

// Bilinear upsample of a [hs,ws] disparity map to (X,Y) of the [H,W] image, align_corners=False.
__device__ __forceinline__ float disp_up_at(const float* __restrict__ d, int hs, int ws, bool same_res, int W,
                                            int X, int Y, float ry, float rx) {
    if (same_res) return d[Y * W + X];
    float sy = fmaxf(ry * (Y + 0.5f) - 0.5f, 0.f);
    float sx = fmaxf(rx * (X + 0.5f) - 0.5f, 0.f);
    int y0 = min((int)sy, hs - 1), x0 = min((int)sx, ws - 1);
    int y1 = y0 + (y0 < hs - 1), x1 = x0 + (x0 < ws - 1);
    float ly = sy - y0, lx = sx - x0;
    float v00 = d[y0 * ws + x0], v01 = d[y0 * ws + x1];
    float v10 = d[y1 * ws + x0], v11 = d[y1 * ws + x1];
    return (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
}

struct Geo {            // per-launch constants of the warp
    float min_disp, disp_range;
    float inv_wm1, inv_hm1, wm1, hm1;
    int H, W;
};

struct Warp {
    float depth, c0, c1, c2;      // depth and the pixel ray inv_K.[X,Y,1]
    float rden, u, v;             // projective divide (rden = 1/(z + eps))
    float gx, gy;                 // normalised grid (Project3D output)
    float mx, my;                 // 1 where the coordinate was NOT clipped (gradient passes)
    int x0, y0;
    float tx, ty;
    bool in_x1, in_y1;
};

// Pixel ray inv_K[:3,:3].[X,Y,1] (BackprojectDepth's pix_coords product): scale- and frame-invariant.
__device__ __forceinline__ void pixel_ray(const float* __restrict__ iK, int X, int Y, float& c0, float& c1, float& c2) {
    float fx = (float)X, fy = (float)Y;
    c0 = fmaf(iK[1], fy, iK[0] * fx) + iK[2];
    c1 = fmaf(iK[4], fy, iK[3] * fx) + iK[5];
    c2 = fmaf(iK[7], fy, iK[6] * fx) + iK[8];
}

// depth * ray -> Project3D -> grid_sample coordinate, with the reference's operation order.  P = (K @ T)[:3,:].
__device__ __forceinline__ void warp_project(const float* __restrict__ P, const Geo& g, float depth, float c0, float c1,
                                             float c2, Warp& w) {
    w.depth = depth;
    w.c0 = c0;
    w.c1 = c1;
    w.c2 = c2;
    float X3 = depth * c0, Y3 = depth * c1, Z3 = depth * c2;
    float p0 = fmaf(P[2], Z3, fmaf(P[1], Y3, P[0] * X3)) + P[3];
    float p1 = fmaf(P[6], Z3, fmaf(P[5], Y3, P[4] * X3)) + P[7];
    float p2 = fmaf(P[10], Z3, fmaf(P[9], Y3, P[8] * X3)) + P[11];
    w.rden = frcp(p2 + 1e-7f);
    w.u = p0 * w.rden;
    w.v = p1 * w.rden;
    w.gx = (w.u * g.inv_wm1 - 0.5f) * 2.f;
    w.gy = (w.v * g.inv_hm1 - 0.5f) * 2.f;
    float ix = ((w.gx + 1.f) * 0.5f) * g.wm1;
    float iy = ((w.gy + 1.f) * 0.5f) * g.hm1;
    w.mx = (ix > 0.f && ix < g.wm1) ? 1.f : 0.f;
    w.my = (iy > 0.f && iy < g.hm1) ? 1.f : 0.f;
    ix = fminf(fmaxf(ix, 0.f), g.wm1);
    iy = fminf(fmaxf(iy, 0.f), g.hm1);
    float x0f = floorf(ix), y0f = floorf(iy);
    w.x0 = (int)x0f;
    w.y0 = (int)y0f;
    w.tx = ix - x0f;
    w.ty = iy - y0f;
    w.in_x1 = (w.x0 + 1) <= g.W - 1;
    w.in_y1 = (w.y0 + 1) <= g.H - 1;
}

// BackprojectDepth -> Project3D -> grid_sample coordinate for one pixel and one frame.
__device__ __forceinline__ void warp_geom(const CamMats& m, const Geo& g, int X, int Y, float disp_up, Warp& w) {
    float c0, c1, c2;
    pixel_ray(m.iK, X, Y, c0, c1, c2);
    warp_project(m.P, g, frcp(g.min_disp + g.disp_range * disp_up), c0, c1, c2, w);
}

// Gather the 4 neighbours of one channel plane (out-of-range upper neighbours contribute 0).
__device__ __forceinline__ void gather4(const float* __restrict__ plane, int W, const Warp& w, float& nw,
                                        float& ne, float& sw, float& se) {
    const float* r0 = plane + w.y0 * W + w.x0;
    int dx = w.in_x1 ? 1 : 0;
    int dy = w.in_y1 ? W : 0;
    nw = r0[0];
    float a = r0[dx], b = r0[dy], c = r0[dy + dx];
    ne = w.in_x1 ? a : 0.f;
    sw = w.in_y1 ? b : 0.f;
    se = (w.in_x1 && w.in_y1) ? c : 0.f;
}

__device__ __forceinline__ float bilerp(const Warp& w, float nw, float ne, float sw, float se) {
    float wx1 = w.tx, wx0 = 1.f - w.tx, wy1 = w.ty, wy0 = 1.f - w.ty;
    return nw * (wx0 * wy0) + ne * (wx1 * wy0) + sw * (wx0 * wy1) + se * (wx1 * wy1);
}

// The two source frames repacked as RGBA pixels (chain_pack_kernel, once per forward call, kept for the backward): the four
// taps of a bilinear sample are four 16-byte loads with one address computation instead of twelve 4-byte loads on three
// planes -- a third of the gather instructions through the CU's texture-address unit and a third of the address arithmetic.
// A neighbour beyond the last column / row is read at the clamped position instead: its bilinear weight is exactly 0 there
// (the coordinate was clamped to W-1 / H-1) and, in the backward, the coordinate mask mx / my is 0.
using f4 = __attribute__((ext_vector_type(4))) float;
__device__ __forceinline__ void gather_rgba(const f4* __restrict__ img, int W, const Warp& w, f4& nw, f4& ne, f4& sw, f4& se) {
    const f4* r0 = img + (w.y0 * W + w.x0);
    const int dx = w.in_x1 ? 1 : 0, dy = w.in_y1 ? W : 0;
    nw = r0[0];
    ne = r0[dx];
    sw = r0[dy];
    se = r0[dy + dx];
}

__device__ __forceinline__ f4 bilerp4(const Warp& w, f4 nw, f4 ne, f4 sw, f4 se) {
    const float wx1 = w.tx, wx0 = 1.f - w.tx, wy1 = w.ty, wy0 = 1.f - w.ty;
    return nw * (wx0 * wy0) + ne * (wx1 * wy0) + sw * (wx0 * wy1) + se * (wx1 * wy1);
}

// Horizontal 3-sums of one window row: x, y, x^2, y^2, xy.
struct RowSums {
    float x, y, xx, yy, xy;
};
__device__ __forceinline__ RowSums row_sums(const float* __restrict__ xr, const float* __restrict__ yr) {
    float x0 = xr[0], x1 = xr[1], x2 = xr[2], y0 = yr[0], y1 = yr[1], y2 = yr[2];
    RowSums r;
    r.x = x0 + x1 + x2;
    r.y = y0 + y1 + y2;
    r.xx = fmaf(x2, x2, fmaf(x1, x1, x0 * x0));
    r.yy = fmaf(y2, y2, fmaf(y1, y1, y0 * y0));
    r.xy = fmaf(x2, y2, fmaf(x1, y1, x0 * y0));
    return r;
}

struct Stats {
    float mux, muy, sigx, sigy, sigxy;
};
__device__ __forceinline__ Stats stats_of(const RowSums& a, const RowSums& b, const RowSums& c) {
    Stats s;
    s.mux = (a.x + b.x + c.x) * K9;
    s.muy = (a.y + b.y + c.y) * K9;
    s.sigx = (a.xx + b.xx + c.xx) * K9 - s.mux * s.mux;
    s.sigy = (a.yy + b.yy + c.yy) * K9 - s.muy * s.muy;
    s.sigxy = (a.xy + b.xy + c.xy) * K9 - s.mux * s.muy;
    return s;
}

// ssim_ratio * mean_c SSIM + (1 - ssim_ratio) * mean_c |t - p| (learner_new.py:60-74) for the 4-row strip
// whose first 3x3 window starts at (ly0, lx) of the halo tiles.
template <int LD, int PLANE>
__device__ __forceinline__ void reproj_strip(const float* __restrict__ sX, const float* __restrict__ sT, int ly0,
                                             int lx, float ssim_ratio, float out[PX]) {
    float ssim_sum[PX] = {0.f, 0.f, 0.f, 0.f}, l1_sum[PX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        const float* x = sX + c * PLANE + ly0 * LD + lx;
        const float* y = sT + c * PLANE + ly0 * LD + lx;
        RowSums r[PX + 2];
#pragma unroll
        for (int j = 0; j < PX + 2; ++j) {
            r[j] = row_sums(x + j * LD, y + j * LD);
            if (j >= 1 && j <= PX) l1_sum[j - 1] += fabsf(y[j * LD + 1] - x[j * LD + 1]);
        }
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            Stats s = stats_of(r[k], r[k + 1], r[k + 2]);
            float n = (2.f * s.mux * s.muy + C1) * (2.f * s.sigxy + C2);
            float d = (s.mux * s.mux + s.muy * s.muy + C1) * (s.sigx + s.sigy + C2);
            float v = (1.f - n * frcp(d)) * 0.5f;
            ssim_sum[k] += fminf(fmaxf(v, 0.f), 1.f);
        }
    }
#pragma unroll
    for (int k = 0; k < PX; ++k)
        out[k] = ssim_ratio * (ssim_sum[k] * (1.f / 3.f)) + (1.f - ssim_ratio) * (l1_sum[k] * (1.f / 3.f));
}

// The same for TWO predictions of one target at once (the two source frames of a scale, or the two identity candidates):
// the target-side row sums (y, y^2) and statistics (mu_y, sigma_y) are formed once and shared.
template <int LD, int PLANE>
__device__ __forceinline__ void reproj_pair(const float* __restrict__ sXa, const float* __restrict__ sXb,
                                            const float* __restrict__ sT, int ly0, int lx, float ssim_ratio, float outa[PX],
                                            float outb[PX]) {
    float ssa[PX] = {0.f, 0.f, 0.f, 0.f}, ssb[PX] = {0.f, 0.f, 0.f, 0.f};
    float l1a[PX] = {0.f, 0.f, 0.f, 0.f}, l1b[PX] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < 3; ++c) {
        const int o = c * PLANE + ly0 * LD + lx;
        const float* xa = sXa + o;
        const float* xb = sXb + o;
        const float* y = sT + o;
        float ry[PX + 2], ryy[PX + 2], ax[PX + 2], axx[PX + 2], axy[PX + 2], bx[PX + 2], bxx[PX + 2], bxy[PX + 2];
#pragma unroll
        for (int j = 0; j < PX + 2; ++j) {
            const float y0 = y[j * LD], y1 = y[j * LD + 1], y2 = y[j * LD + 2];
            const float a0 = xa[j * LD], a1 = xa[j * LD + 1], a2 = xa[j * LD + 2];
            const float b0 = xb[j * LD], b1 = xb[j * LD + 1], b2 = xb[j * LD + 2];
            ry[j] = y0 + y1 + y2;
            ryy[j] = fmaf(y2, y2, fmaf(y1, y1, y0 * y0));
            ax[j] = a0 + a1 + a2;
            axx[j] = fmaf(a2, a2, fmaf(a1, a1, a0 * a0));
            axy[j] = fmaf(a2, y2, fmaf(a1, y1, a0 * y0));
            bx[j] = b0 + b1 + b2;
            bxx[j] = fmaf(b2, b2, fmaf(b1, b1, b0 * b0));
            bxy[j] = fmaf(b2, y2, fmaf(b1, y1, b0 * y0));
            if (j >= 1 && j <= PX) {
                l1a[j - 1] += fabsf(y1 - a1);
                l1b[j - 1] += fabsf(y1 - b1);
            }
        }
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            const float muy = (ry[k] + ry[k + 1] + ry[k + 2]) * K9;
            const float sigy = (ryy[k] + ryy[k + 1] + ryy[k + 2]) * K9 - muy * muy;
            const float my2 = fmaf(muy, muy, C1), sy2 = sigy + C2;
            {
                const float mux = (ax[k] + ax[k + 1] + ax[k + 2]) * K9;
                const float sigx = (axx[k] + axx[k + 1] + axx[k + 2]) * K9 - mux * mux;
                const float sigxy = (axy[k] + axy[k + 1] + axy[k + 2]) * K9 - mux * muy;
                const float n = (2.f * mux * muy + C1) * (2.f * sigxy + C2);
                const float d = fmaf(mux, mux, my2) * (sigx + sy2);
                ssa[k] += fminf(fmaxf((1.f - n * frcp(d)) * 0.5f, 0.f), 1.f);
            }
            {
                const float mux = (bx[k] + bx[k + 1] + bx[k + 2]) * K9;
                const float sigx = (bxx[k] + bxx[k + 1] + bxx[k + 2]) * K9 - mux * mux;
                const float sigxy = (bxy[k] + bxy[k + 1] + bxy[k + 2]) * K9 - mux * muy;
                const float n = (2.f * mux * muy + C1) * (2.f * sigxy + C2);
                const float d = fmaf(mux, mux, my2) * (sigx + sy2);
                ssb[k] += fminf(fmaxf((1.f - n * frcp(d)) * 0.5f, 0.f), 1.f);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < PX; ++k) {
        outa[k] = ssim_ratio * (ssa[k] * (1.f / 3.f)) + (1.f - ssim_ratio) * (l1a[k] * (1.f / 3.f));
        outb[k] = ssim_ratio * (ssb[k] * (1.f / 3.f)) + (1.f - ssim_ratio) * (l1b[k] * (1.f / 3.f));
    }
}

// ------------------------------------------------------------------------------ counter-based noise
// Philox4x32-10 (Salmon et al.); stands in for the reference's torch.randn tie-break noise
// (learner_new.py:226-229) when the caller does not inject a noise tensor.
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0,
                                           uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
    float u1 = ((float)a + 0.5f) * 2.3283064365386963e-10f;  // (0,1)
    float u2 = ((float)b + 0.5f) * 2.3283064365386963e-10f;
    float r = sqrtf(-2.f * __logf(u1));
    float s, c;
    __sincosf(6.283185307179586f * u2, &s, &c);
    n0 = r * c;
    n1 = r * s;
}

// A workgroup-uniform float that the vector ALU computed (divisions, int -> float conversions) lives in a VECTOR register for the
// rest of the kernel unless told otherwise; in chain_bwd_kernel, which runs at its register limit, two dozen of them were spilled
// to scratch in the prologue and re-read in every loop (+ 0.5 GB of HBM traffic per launch).  v_readfirstlane moves them to SGPRs.
__device__ __forceinline__ float uni(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }

__device__ __forceinline__ Geo make_geo(const dvs_chain_cfg& c) {
    Geo g;
    g.min_disp = uni(1.0f / c.max_depth);
    g.disp_range = uni(1.0f / c.min_depth - 1.0f / c.max_depth);
    g.wm1 = uni((float)(c.W - 1));
    g.hm1 = uni((float)(c.H - 1));
    g.inv_wm1 = uni(1.0f / g.wm1);
    g.inv_hm1 = uni(1.0f / g.hm1);
    g.H = c.H;
    g.W = c.W;
    return g;
}

// exp(-mean_c |a - b|) of two pixels of the LDS target tile
template <int PLANE>
__device__ __forceinline__ float edge_weight(const float* __restrict__ sT, int o0, int o1) {
    float gi = fabsf(sT[o0] - sT[o1]) + fabsf(sT[PLANE + o0] - sT[PLANE + o1]) +
               fabsf(sT[2 * PLANE + o0] - sT[2 * PLANE + o1]);
    return __expf(-gi * (1.f / 3.f));
}

// ------------------------------------------------------------------------------------- forward
// Per-image camera table behind the [B][S][4] statistics of the `stats` workspace: CAM_STRIDE floats per image =
// inv_K[:3,:3] (9, padded to 12), (K @ T_-1)[:3,:] (12), (K @ T_+1)[:3,:] (12).  Written by one tiny kernel per forward
// call; the main kernels read it with scalar loads (uniform per workgroup) instead of re-deriving K @ T on the vector
// ALUs for every scale and frame, and the backward finds it where the forward left it.
constexpr int CAM_STRIDE = 36;
__host__ __device__ __forceinline__ size_t stats_floats(int B, int S) {       // statistics + camera table, padded to 16 bytes
    return ((size_t)B * S * 4 + (size_t)B * CAM_STRIDE + 3) / 4 * 4;
}
__device__ __forceinline__ const float* cam_table(const ChainParams& p, int b) {
    return p.io.stats + (size_t)p.cfg.B * p.cfg.num_scales * 4 + (size_t)b * CAM_STRIDE;
}
// RGBA copies of the source frames behind that: [2 frames][B][H*W] pixels of 16 bytes
__device__ __forceinline__ const f4* packed_source(const ChainParams& p, int f, int b) {
    const f4* base = reinterpret_cast<const f4*>(p.io.stats + stats_floats(p.cfg.B, p.cfg.num_scales));
    return base + ((size_t)f * p.cfg.B + b) * ((size_t)p.cfg.H * p.cfg.W);
}

__global__ __launch_bounds__(NT) void chain_pack_kernel(ChainParams p) {
    const int HW = p.cfg.H * p.cfg.W;
    const size_t n = (size_t)2 * p.cfg.B * HW;
    f4* out = reinterpret_cast<f4*>(p.io.stats + stats_floats(p.cfg.B, p.cfg.num_scales));
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const int pix = (int)(i % HW);
        const size_t fb = i / HW;
        const int b = (int)(fb % p.cfg.B), f = (int)(fb / p.cfg.B);
        const float* src = p.io.source[f] + (size_t)b * 3 * HW + pix;
        out[i] = f4{src[0], src[HW], src[2 * HW], 0.f};
    }
}

__global__ void chain_cam_kernel(ChainParams p) {
    const int b = blockIdx.x, t = threadIdx.x;
    float* out = p.io.stats + (size_t)p.cfg.B * p.cfg.num_scales * 4 + (size_t)b * CAM_STRIDE;
    if (t < 12) {
        const float* iK = p.io.inv_K + b * 16;
        out[t] = (t < 9) ? iK[(t / 3) * 4 + (t % 3)] : 0.f;
    } else if (t < 36) {
        const int f = (t - 12) / 12, e = (t - 12) % 12, i = e / 4, j = e % 4;
        const float* K = p.io.K + b * 16;
        const float* T = p.io.T[f] + b * 16;
        float acc = K[i * 4 + 0] * T[0 * 4 + j];
        acc = fmaf(K[i * 4 + 1], T[1 * 4 + j], acc);
        acc = fmaf(K[i * 4 + 2], T[2 * 4 + j], acc);
        acc = fmaf(K[i * 4 + 3], T[3 * 4 + j], acc);
        out[t] = acc;
    }
}

// Halo positions sH[i] of the (TH+2) x (TW+2) tile, packed as gx | gy << 16 | interior << 31 (gx, gy = reflected image
// coordinates; interior = a pixel of the tile proper that lies inside the image).  The division by the tile width and the
// two reflections are done once per workgroup; every staging loop (S scales x 2 frames) reads one LDS word instead.

__global__ __launch_bounds__(NT, FWD_WAVES) void chain_fwd_kernel(ChainParams p) {
    __shared__ float sT[3 * FH * FW];
    __shared__ float sX[2][3 * FH * FW];       // the two predictions compared with the target tile (both frames at once)
    __shared__ float sD[FH * FW];
    __shared__ uint32_t sH[FH * FW];            // halo positions, see below
    __shared__ float sRed[NT / 64][NPART];

    const dvs_chain_cfg& c = p.cfg;
    const int H = c.H, W = c.W, HW = H * W, S = c.num_scales;
    const int b = blockIdx.z, X0 = blockIdx.x * TW, Y0 = blockIdx.y * TH;
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const Geo geo = make_geo(c);
    constexpr int PL = FH * FW;
    const int X = X0 + tx, Yb = Y0 + PX * ty;      // my strip: column X, rows Yb .. Yb+3

    const float* tgt = p.io.target + (size_t)b * 3 * HW;
    const float* src0 = p.io.source[0] + (size_t)b * 3 * HW;
    const float* src1 = p.io.source[1] + (size_t)b * 3 * HW;
    for (int i = tid; i < PL; i += NT) {
        const int hy = i / FW, hx = i - hy * FW;
        const int px = X0 - 1 + hx, py = Y0 - 1 + hy;
        const int gx = reflect_idx(px, W), gy = reflect_idx(py, H);
        const bool interior = hx >= 1 && hx <= TW && hy >= 1 && hy <= TH && px < W && py < H;
        sH[i] = (uint32_t)gx | ((uint32_t)gy << 16) | (interior ? 0x80000000u : 0u);
        const int o = gy * W + gx;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) sT[ch * PL + i] = tgt[ch * HW + o];
        if (c.auto_mask) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                sX[0][ch * PL + i] = src0[ch * HW + o];
                sX[1][ch * PL + i] = src1[ch * HW + o];
            }
        }
    }
    __syncthreads();
    float ident[2][PX] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (c.auto_mask) {
        reproj_pair<FW, PL>(sX[0], sX[1], sT, PX * ty, tx, c.ssim_ratio, ident[0], ident[1]);
        __syncthreads();
    }

    uint32_t selbits[PX] = {0, 0, 0, 0};
    uint32_t rnd[PX][2] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};      // second half of a Philox block, kept for the odd scale
    const int wave = tid >> 6, lane = tid & 63;
    if (tid < NPART) sRed[0][tid] = sRed[1][tid] = sRed[2][tid] = sRed[3][tid] = 0.f;
    const float* cam = cam_table(p, b);        // uniform: scalar loads

#pragma unroll 1
    for (int s = 0; s < S; ++s) {
        const int hs = c.hs[s], ws = c.ws[s];
        const bool same_res = (hs == H && ws == W);
        const float* dsp = p.io.disp[s] + (size_t)b * hs * ws;
        const float ry = (float)hs / (float)H, rx = (float)ws / (float)W;
        float rp[2][PX];
        float a_min = 0.f, a_disp = 0.f, a_gx = 0.f, a_gy = 0.f;   // this scale's partial sums
#pragma unroll 1
        for (int f = 0; f < 2; ++f) {
            const float* P = cam + 12 + 12 * f;
            const f4* src4 = packed_source(p, f, b);
            float* sXf = sX[f];
            float* o_color = p.io.color[s][f];
            float* o_grid = p.io.grid[s][f];
            float* o_dup = (f == 0) ? p.io.disp_up[s] : nullptr;
            float* o_depth = (f == 0) ? p.io.depth[s] : nullptr;
            const bool materialize = o_color || o_grid || o_dup || o_depth;
            for (int i = tid; i < PL; i += NT) {
                const uint32_t h = sH[i];
                const int gx = (int)(h & 0xffffu), gy = (int)((h >> 16) & 0x7fffu);
                // the upsampled disparity is formed for frame -1 and re-read from the LDS tile for frame +1
                float du;
                if (f == 0) {
                    du = disp_up_at(dsp, hs, ws, same_res, W, gx, gy, ry, rx);
                    sD[i] = du;
                } else {
                    du = sD[i];
                }
                const float depth = frcp(geo.min_disp + geo.disp_range * du);
                float c0, c1, c2;
                pixel_ray(cam, gx, gy, c0, c1, c2);
                Warp w;
                warp_project(P, geo, depth, c0, c1, c2, w);
                f4 nw, ne, sw, se;
                gather_rgba(src4, W, w, nw, ne, sw, se);
                const f4 c4 = bilerp4(w, nw, ne, sw, se);
                const float col[3] = {c4[0], c4[1], c4[2]};
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) sXf[ch * PL + i] = col[ch];
                // optional materialisation of the reference's `outputs` tensors (interior pixels only)
                if (materialize && (h >> 31)) {
                    const size_t o = (size_t)b * HW + (size_t)gy * W + gx;        // interior: (gx, gy) is the pixel itself
                    if (o_color) {
#pragma unroll
                        for (int ch = 0; ch < 3; ++ch)
                            o_color[(size_t)b * 3 * HW + (size_t)ch * HW + (size_t)gy * W + gx] = col[ch];
                    }
                    if (o_grid) {
                        o_grid[o * 2 + 0] = w.gx;
                        o_grid[o * 2 + 1] = w.gy;
                    }
                    if (o_dup) o_dup[o] = du;
                    if (o_depth) o_depth[o] = depth;
                }
            }
        }
        __syncthreads();
        reproj_pair<FW, PL>(sX[0], sX[1], sT, PX * ty, tx, c.ssim_ratio, rp[0], rp[1]);

#pragma unroll
        for (int k = 0; k < PX; ++k) {
            int Y = Yb + k;
            if (X >= W || Y >= H) continue;
            // 4-way min with first-minimum-wins ties; candidates 0,1 = identity(-1,+1), 2,3 = reprojection(-1,+1)
            float best = rp[0][k];
            uint32_t idx = 2;
            if (c.auto_mask) {
                float n0, n1;
                if (p.io.noise) {
                    const float* nz = p.io.noise + ((size_t)s * c.B + b) * 2 * HW + (size_t)Y * W + X;
                    n0 = nz[0];
                    n1 = nz[HW];
                } else if (s & 1) {
                    box_muller(rnd[k][0], rnd[k][1], n0, n1);            // second half of the block drawn at scale s - 1
                } else {
                    uint32_t r[4];
                    philox4x32((uint32_t)(Y * W + X), (uint32_t)b, (uint32_t)(s >> 1), 0u,
                               (uint32_t)p.io.seed, (uint32_t)(p.io.seed >> 32), r);
                    box_muller(r[0], r[1], n0, n1);
                    rnd[k][0] = r[2];
                    rnd[k][1] = r[3];
                }
                float i0 = ident[0][k] + n0 * 0.00001f, i1 = ident[1][k] + n1 * 0.00001f;
                best = i0;
                idx = 0;
                if (i1 < best) { best = i1; idx = 1; }
                if (rp[0][k] < best) { best = rp[0][k]; idx = 2; }
            }
            if (rp[1][k] < best) { best = rp[1][k]; idx = 3; }
            a_min += best;
            selbits[k] |= idx << (2 * s);

            // smoothness partial sums on the un-normalised disparity (the per-image mean divides out
            // in the finalize kernel): learner_new.py:246-250, learner_func.py:161-174
            int o = (PX * ty + k + 1) * FW + tx + 1;
            float d0 = sD[o];
            a_disp += d0;
            if (X < W - 1) a_gx += fabsf(d0 - sD[o + 1]) * edge_weight<PL>(sT, o, o + 1);
            if (Y < H - 1) a_gy += fabsf(d0 - sD[o + FW]) * edge_weight<PL>(sT, o, o + FW);
        }
        // wave butterfly now, workgroup sum after the scale loop
        a_min = dvs::wave_sum(a_min);
        a_disp = dvs::wave_sum(a_disp);
        a_gx = dvs::wave_sum(a_gx);
        a_gy = dvs::wave_sum(a_gy);
        if (lane == 0) {
            sRed[wave][s * 4 + 0] = a_min;
            sRed[wave][s * 4 + 1] = a_disp;
            sRed[wave][s * 4 + 2] = a_gx;
            sRed[wave][s * 4 + 3] = a_gy;
        }
        __syncthreads();  // sD / sX are rewritten by the next scale
    }

#pragma unroll
    for (int k = 0; k < PX; ++k) {
        int Y = Yb + k;
        if (X < W && Y < H) p.io.sel[(size_t)b * HW + (size_t)Y * W + X] = (uint8_t)selbits[k];
    }

    // workgroup reduction: per-wave sums sit in LDS -> 16 lanes
    if (tid < NPART) {
        float v = sRed[0][tid] + sRed[1][tid] + sRed[2][tid] + sRed[3][tid];
        int tile = blockIdx.y * p.tiles_x + blockIdx.x;
        p.io.partials[((size_t)b * p.tiles_x * p.tiles_y + tile) * NPART + tid] = v;
    }
}

// Stage 1: per image, sum the tile partials -> stats[b][s] = {sum min-loss, sum disp_up, Gx, Gy}.
__global__ __launch_bounds__(NT) void chain_fwd_reduce_kernel(ChainParams p) {
    __shared__ float sRed[NT / 64][NPART];
    const int b = blockIdx.x, ntiles = p.tiles_x * p.tiles_y, tid = threadIdx.x;
    // thread t handles column (t & 15) of rows (t >> 4), (t >> 4) + 16, ...
    const int col = tid & 15;
    float v = 0.f;
    for (int r = tid >> 4; r < ntiles; r += NT / 16) v += p.io.partials[((size_t)b * ntiles + r) * NPART + col];
    // lanes with equal `col` inside a wave: xor-reduce over lane bits 4,5
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if ((tid & 63) < 16) sRed[tid >> 6][col] = v;
    __syncthreads();
    if (tid < NPART) {
        float t = sRed[0][tid] + sRed[1][tid] + sRed[2][tid] + sRed[3][tid];
        int s = tid >> 2, j = tid & 3;
        if (s < p.cfg.num_scales) p.io.stats[((size_t)b * p.cfg.num_scales + s) * 4 + j] = t;
    }
}

// Stage 2: losses[s] = mean(min) + ratio/2^s * (mean_x + mean_y) of the normalised smoothness terms.
__global__ void chain_fwd_losses_kernel(ChainParams p) {
    const dvs_chain_cfg& c = p.cfg;
    int s = threadIdx.x;
    if (s >= c.num_scales) return;
    float minsum = 0.f, gx = 0.f, gy = 0.f;
    float hw = (float)c.H * (float)c.W;
    for (int b = 0; b < c.B; ++b) {
        const float* st = p.io.stats + ((size_t)b * c.num_scales + s) * 4;
        float mean = fmaxf(st[1] / hw, 0.001f) + 1e-7f;
        minsum += st[0];
        gx += st[2] / mean;
        gy += st[3] / mean;
    }
    float nx = (float)c.B * (float)c.H * (float)(c.W - 1), ny = (float)c.B * (float)(c.H - 1) * (float)c.W;
    float smooth = gx / nx + gy / ny;
    p.io.losses[s] = minsum / ((float)c.B * hw) + c.smoothness_ratio * smooth / (float)(1 << s);
}

// ------------------------------------------------------------------------------------ backward
struct BwdParams {
    dvs_chain_bwd_io g;
};

__device__ __forceinline__ float sgn(float v) { return (v > 0.f) ? 1.f : ((v < 0.f) ? -1.f : 0.f); }

// NTB threads own PXB rows of a column each (NTB / 64 waves x PXB = the TH rows of the tile): 256 x 4 at two waves per SIMD, or
// 512 x 2 at four -- same tile and LDS, half the per-thread state, twice the waves to hide LDS / gather latency and barriers behind.
template <int NTB, int PXB>
__global__ __launch_bounds__(NTB, NTB == 512 ? 4 : BWD_WAVES) void chain_bwd_kernel(ChainParams p, BwdParams q) {
    static_assert((NTB / 64) * PXB == TH, "waves x rows per lane = tile height");
    constexpr int NWV = NTB / 64;
    constexpr int RING = BH * BW - TH * TW;                                      // pixels of the 2-px halo ring
    constexpr int FR = NTB == 512 ? 3 : FIELD_ROWS, FT = FW * (FH / FR);      // field strips: 66 columns x (18 / FR) row groups
    static_assert(FH % FR == 0 && FT <= NTB, "field strips must tile the 1-px-halo tile");
    __shared__ float sT[3 * BH * BW];
    __shared__ float sX[3 * BH * BW];
    __shared__ float sD[BH * BW];
    __shared__ float sF[3 * FH * FW];     // SSIM derivative fields of one channel; reused as sAcc
    __shared__ uint8_t sSel[FH * FW];
    __shared__ uint32_t sH[BH * BW];      // halo positions gx | gy << 16 of the 2-px-halo tile (see the forward kernel)
    __shared__ float sRed[NTB / 64][NDP];
    float* sAcc = sF;

    const dvs_chain_cfg& c = p.cfg;
    const int H = c.H, W = c.W, HW = H * W, S = c.num_scales;
    const int b = blockIdx.z, X0 = blockIdx.x * TW, Y0 = blockIdx.y * TH;
    const int tid0 = threadIdx.x;
    const int wave = tid0 >> 6, lane = tid0 & 63;
    const Geo geo = make_geo(c);
    constexpr int PLB = BH * BW, PLF = FH * FW;
    const int tile = blockIdx.y * p.tiles_x + blockIdx.x, ntiles = p.tiles_x * p.tiles_y;
    int tid = tid0;

    const float* tgt = p.io.target + (size_t)b * 3 * HW;
    for (int i = tid; i < PLB; i += NTB) {
        int hy = i / BW, hx = i - hy * BW;
        int gx = reflect_idx(X0 - 2 + hx, W), gy = reflect_idx(Y0 - 2 + hy, H);
        sH[i] = (uint32_t)gx | ((uint32_t)gy << 16);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) sT[ch * PLB + i] = tgt[ch * HW + gy * W + gx];
    }
    const float* cam = cam_table(p, b);        // uniform: scalar loads (written by the forward call)
    for (int i = tid; i < PLF; i += NTB) {
        int hy = i / FW, hx = i - hy * FW;
        int px = X0 - 1 + hx, py = Y0 - 1 + hy;
        bool in = px >= 0 && px < W && py >= 0 && py < H;
        // pixels outside the image never match a frame id (0xFF -> every 2-bit field is 3 only if stored so;
        // use a separate validity test below instead)
        sSel[i] = in ? p.io.sel[(size_t)b * HW + (size_t)py * W + px] : (uint8_t)0;
    }
    __syncthreads();

    const float w_pix = uni(1.0f / ((float)c.B * (float)HW));
    const float cx = uni(1.0f / ((float)c.B * (float)H * (float)(W - 1)));
    const float cy = uni(1.0f / ((float)c.B * (float)(H - 1) * (float)W));
    const int s_lo = (q.g.scale_end > q.g.scale_begin) ? q.g.scale_begin : 0;
    const int s_hi = (q.g.scale_end > q.g.scale_begin) ? min(q.g.scale_end, S) : S;
#pragma unroll 1
    for (int s = s_lo; s < s_hi; ++s) {
        // Everything a lane derives from its thread index -- strip coordinates, LDS addresses with their constants folded in, the
        // 64-bit output pointers -- is loop invariant, so the compiler hoists it all in front of this loop and, at the register
        // limit, spills it there and reloads it in every phase (two dozen dwords, + 0.5 GB of scratch traffic per launch).  An opaque
        // copy of the thread index per scale makes those few adds live where they are used.
        asm volatile("" : "+v"(tid));
        const int tx = tid & 63, ty = tid >> 6;
        const int X = X0 + tx, Yb = Y0 + PXB * ty;      // my strip: column X, rows Yb .. Yb + PXB - 1
        // ReflectionPad2d(1): the window of border pixel 0 (W-1) reads column 1 (W-2) twice
        const float wxm = (X == 1) ? 2.f : 1.f, wxp = (X == W - 2) ? 2.f : 1.f;
        // field strip owned by this thread: column fcol, rows frow0 .. frow0 + FR - 1 of the 1-px-halo tile
        const int fcol = tid % FW, frow0 = (tid / FW) * FR;
        const bool field_thread = tid < FT;
        const int fpx = X0 - 1 + fcol;
        const int hs = c.hs[s], ws = c.ws[s];
        const bool same_res = (hs == H && ws == W);
        const float* dsp = p.io.disp[s] + (size_t)b * hs * ws;
        const float ry = uni((float)hs / (float)H), rx = uni((float)ws / (float)W);
        const float gl = q.g.d_losses[s];
        const float w_ssim = uni(gl * w_pix * c.ssim_ratio * (1.f / 3.f) * (-0.5f) * K9);
        const float w_l1 = uni(gl * w_pix * (1.f - c.ssim_ratio) * (1.f / 3.f));
        float gd[PXB];                        // d loss / d disp_up at my pixels
#pragma unroll
        for (int k = 0; k < PXB; ++k) gd[k] = 0.f;

#pragma unroll 1
        for (int f = 0; f < 2; ++f) {
            const float* P = cam + 12 + 12 * f;
            const f4* src4 = packed_source(p, f, b);
            const uint32_t want = 2u + (uint32_t)f;
            // Staging: warp every pixel of the tile + 2-px halo into sX.  A thread stages ITS OWN pixels first and keeps the x / y
            // differences of their four taps (times the coordinate masks and the projective 1 / z) in registers -- the chain phase
            // below then needs neither the projection's gather nor the taps again (it was a fifth of the kernel: four dependent
            // gathers per pixel at the end of every frame) -- and the 336 pixels of the halo ring are spread over the threads.
            auto stage = [&](int i, float (*keep)[2]) __attribute__((always_inline)) {
                const uint32_t h = sH[i];
                const int gx = (int)(h & 0xffffu), gy = (int)(h >> 16);
                float du;
                if (f == 0) {
                    du = disp_up_at(dsp, hs, ws, same_res, W, gx, gy, ry, rx);
                    sD[i] = du;
                } else {
                    du = sD[i];
                }
                float c0, c1, c2;
                pixel_ray(cam, gx, gy, c0, c1, c2);
                Warp w;
                warp_project(P, geo, frcp(geo.min_disp + geo.disp_range * du), c0, c1, c2, w);
                f4 nw, ne, sw, se;
                if (CHAIN_DBG & 1) nw = ne = sw = se = f4{w.u, w.v, w.gx, w.gy};
                else gather_rgba(src4, W, w, nw, ne, sw, se);
                const f4 c4 = bilerp4(w, nw, ne, sw, se);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) sX[ch * PLB + i] = c4[ch];
                if (keep) {
                    // d colour / d ix, d colour / d iy (grid_sample's bilinear derivative) times d ix / d u = mask, and 1 / z
                    const f4 ddx = ((ne - nw) * (1.f - w.ty) + (se - sw) * w.ty) * (w.mx * w.rden);
                    const f4 ddy = ((sw - nw) * (1.f - w.tx) + (se - ne) * w.tx) * (w.my * w.rden);
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch) {
                        keep[ch][0] = ddx[ch];
                        keep[ch][1] = ddy[ch];
                    }
                }
            };
            float tapd[PXB][3][2];
#pragma unroll
            for (int k = 0; k < PXB; ++k) stage((PXB * ty + k + 2) * BW + tx + 2, tapd[k]);
            for (int j = tid; j < RING; j += NTB) {
                // ring pixel j -> tile index: rows 0, 1, then rows BH-2, BH-1, then columns 0, 1, BW-2, BW-1 of the rows between
                int i;
                if (j < 2 * BW) i = j;
                else if (j < 4 * BW) i = (BH - 4) * BW + j;                    // (BH - 2) * BW + (j - 2 * BW)
                else {
                    const int r = j - 4 * BW, c = r & 3;
                    i = (2 + (r >> 2)) * BW + (c < 2 ? c : BW - 4 + c);
                }
                stage(i, nullptr);
            }
            __syncthreads();

            float dcol[PXB][3];
#pragma unroll 1
            for (int ch = 0; ch < 3; ++ch) {
                // SSIM derivative coefficient fields at every pixel p of tile + 1-px halo:
                // d out / d x(r) = -1/2 * 1/9 * (alpha(p) + beta(p) x(r) + gamma(p) y(r)) for r in window(p)
                if (field_thread && !(CHAIN_DBG & 2)) {
                    const float* xw = sX + ch * PLB + frow0 * BW + fcol;
                    const float* yw = sT + ch * PLB + frow0 * BW + fcol;
                    RowSums r0 = row_sums(xw, yw), r1 = row_sums(xw + BW, yw + BW);
#pragma unroll
                    for (int j = 0; j < FR; ++j) {
                        RowSums r2 = row_sums(xw + (j + 2) * BW, yw + (j + 2) * BW);
                        int hy = frow0 + j, i = hy * FW + fcol, py = Y0 - 1 + hy;
                        float fa = 0.f, fb = 0.f, fc = 0.f;
                        bool in = fpx >= 0 && fpx < W && py >= 0 && py < H;
                        if (in && ((sSel[i] >> (2 * s)) & 3u) == want) {
                            Stats st = stats_of(r0, r1, r2);
                            float n1 = 2.f * st.mux * st.muy + C1, n2 = 2.f * st.sigxy + C2;
                            float d1 = st.mux * st.mux + st.muy * st.muy + C1, d2 = st.sigx + st.sigy + C2;
                            float rd1 = frcp(d1), rd2 = frcp(d2), rd = rd1 * rd2;
                            float R = n1 * n2 * rd, v = (1.f - R) * 0.5f;
                            if (v >= 0.f && v <= 1.f) {
                                float dR_dmux = 2.f * st.muy * n2 * rd - R * 2.f * st.mux * rd1;
                                float dR_dsx = -R * rd2;
                                float dR_dsxy = 2.f * n1 * rd;
                                fa = w_ssim * (dR_dmux - 2.f * st.mux * dR_dsx - st.muy * dR_dsxy);
                                fb = w_ssim * 2.f * dR_dsx;
                                fc = w_ssim * dR_dsxy;
                            }
                        }
                        sF[0 * PLF + i] = fa;
                        sF[1 * PLF + i] = fb;
                        sF[2 * PLF + i] = fc;
                        r0 = r1;
                        r1 = r2;
                    }
                }
                __syncthreads();
                if (CHAIN_DBG & 4) {
#pragma unroll
                    for (int k = 0; k < PXB; ++k) dcol[k][ch] = sX[ch * PLB + (PXB * ty + k + 2) * BW + tx + 2];
                } else {
                    // separable 3x3 gather of the three fields for my 4-row strip (rows PXB*ty .. +5 of sF)
                    float ha[PXB + 2], hb[PXB + 2], hc[PXB + 2];
#pragma unroll
                    for (int j = 0; j < PXB + 2; ++j) {
                        int o = (PXB * ty + j) * FW + tx;
                        ha[j] = wxm * sF[o] + sF[o + 1] + wxp * sF[o + 2];
                        hb[j] = wxm * sF[PLF + o] + sF[PLF + o + 1] + wxp * sF[PLF + o + 2];
                        hc[j] = wxm * sF[2 * PLF + o] + sF[2 * PLF + o + 1] + wxp * sF[2 * PLF + o + 2];
                    }
#pragma unroll
                    for (int k = 0; k < PXB; ++k) {
                        int Y = Yb + k;
                        float wym = (Y == 1) ? 2.f : 1.f, wyp = (Y == H - 2) ? 2.f : 1.f;
                        float Sa = wym * ha[k] + ha[k + 1] + wyp * ha[k + 2];
                        float Sb = wym * hb[k] + hb[k + 1] + wyp * hb[k + 2];
                        float Sc = wym * hc[k] + hc[k + 1] + wyp * hc[k + 2];
                        int o2 = (PXB * ty + k + 2) * BW + tx + 2;
                        float xq = sX[ch * PLB + o2], yq = sT[ch * PLB + o2];
                        float g = Sa + Sb * xq + Sc * yq;
                        if (((sSel[(PXB * ty + k + 1) * FW + tx + 1] >> (2 * s)) & 3u) == want) g += w_l1 * sgn(xq - yq);
                        dcol[k][ch] = g;
                    }
                }
                __syncthreads();
            }

            // chain d colour -> grid_sample -> projection -> depth -> disp_up, and dP = d/d (K.T)[:3,:]
            float dP[NDP];
#pragma unroll
            for (int j = 0; j < NDP; ++j) dP[j] = 0.f;
#pragma unroll
            for (int k = 0; k < PXB; ++k) {
                int Y = Yb + k;
                if (X >= W || Y >= H) continue;
                if (CHAIN_DBG & 8) {
                    dP[k] += dcol[k][0] + dcol[k][1] + dcol[k][2];
                    continue;
                }
                float du = sD[(PXB * ty + k + 2) * BW + tx + 2];
                float c0, c1, c2;
                pixel_ray(cam, X, Y, c0, c1, c2);
                Warp w;
                warp_project(P, geo, frcp(geo.min_disp + geo.disp_range * du), c0, c1, c2, w);     // (arithmetic only: no gather)
                // d ix / d u = mask (the (W-1)/2 and 2/(W-1) factors of unnormalise/normalise cancel); masks and 1 / z ride in tapd
                float dp0 = 0.f, dp1 = 0.f;
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    dp0 += dcol[k][ch] * tapd[k][ch][0];
                    dp1 += dcol[k][ch] * tapd[k][ch][1];
                }
                float dp2 = -(dp0 * w.u + dp1 * w.v);
                float X3 = w.depth * w.c0, Y3 = w.depth * w.c1, Z3 = w.depth * w.c2;
                dP[0] += dp0 * X3; dP[1] += dp0 * Y3; dP[2] += dp0 * Z3; dP[3] += dp0;
                dP[4] += dp1 * X3; dP[5] += dp1 * Y3; dP[6] += dp1 * Z3; dP[7] += dp1;
                dP[8] += dp2 * X3; dP[9] += dp2 * Y3; dP[10] += dp2 * Z3; dP[11] += dp2;
                float dc0 = P[0] * dp0 + P[4] * dp1 + P[8] * dp2;
                float dc1 = P[1] * dp0 + P[5] * dp1 + P[9] * dp2;
                float dc2 = P[2] * dp0 + P[6] * dp1 + P[10] * dp2;
                float d_depth = dc0 * w.c0 + dc1 * w.c1 + dc2 * w.c2;
                gd[k] += d_depth * (-w.depth * w.depth * geo.disp_range);
            }
#pragma unroll
            for (int j = 0; j < NDP; ++j) {
                float v = (CHAIN_DBG & 16) ? dP[j] : dvs::wave_sum_lane63(dP[j]);
                if (lane == 63) sRed[wave][j] = v;
            }
            __syncthreads();
            if (tid < NDP) {
                float v = 0.f;
#pragma unroll
                for (int wv = 0; wv < NWV; ++wv) v += sRed[wv][tid];
                q.g.bwd_partials[((((size_t)b * ntiles + tile) * S + s) * 2 + f) * NDP + tid] = v;
            }
            __syncthreads();  // sX, sF, sRed reused by the next frame
        }

        // smoothness gradient (learner_new.py:246-252): loss_s += ratio/2^s * sum_b (Gx_b cx + Gy_b cy) / (m_b + eps)
        {
            const float* st = p.io.stats + ((size_t)b * S + s) * 4;
            float hw = (float)HW;
            float mean_raw = st[1] / hw;
            float mean = fmaxf(mean_raw, 0.001f) + 1e-7f;
            float gs = gl * c.smoothness_ratio / (float)(1 << s);
            float k_grad = gs / mean;
            float k_mean = (mean_raw >= 0.001f) ? -gs * (cx * st[2] + cy * st[3]) / (mean * mean) / hw : 0.f;
#pragma unroll
            for (int k = 0; k < PXB; ++k) {
                int Y = Yb + k;
                if (X >= W || Y >= H) continue;
                int o = (PXB * ty + k + 2) * BW + tx + 2;
                float d0 = sD[o];
                float g = 0.f;
                if (X < W - 1) g += cx * sgn(d0 - sD[o + 1]) * edge_weight<PLB>(sT, o, o + 1);
                if (X > 0) g -= cx * sgn(sD[o - 1] - d0) * edge_weight<PLB>(sT, o - 1, o);
                if (Y < H - 1) g += cy * sgn(d0 - sD[o + BW]) * edge_weight<PLB>(sT, o, o + BW);
                if (Y > 0) g -= cy * sgn(sD[o - BW] - d0) * edge_weight<PLB>(sT, o - BW, o);
                gd[k] += k_grad * g + k_mean;
            }
        }

        // transpose of the bilinear upsample: d disp_up -> d disp_s
        float* dd = q.g.d_disp[s] + (size_t)b * hs * ws;
        if (same_res) {
#pragma unroll
            for (int k = 0; k < PXB; ++k) {
                int Y = Yb + k;
                if (X < W && Y < H) dd[Y * W + X] = gd[k];
            }
        } else {
            // accumulate the tile's low-res footprint in LDS (sAcc aliases sF: the frame loop has ended),
            // then one global atomic per touched low-res pixel
            for (int i = tid; i < ACC_H * ACC_W; i += NTB) sAcc[i] = 0.f;
            int ox = (int)fmaxf(rx * (X0 + 0.5f) - 0.5f, 0.f), oy = (int)fmaxf(ry * (Y0 + 0.5f) - 0.5f, 0.f);
            __syncthreads();
#pragma unroll
            for (int k = 0; k < PXB; ++k) {
                int Y = Yb + k;
                if (X >= W || Y >= H) continue;
                float sy = fmaxf(ry * (Y + 0.5f) - 0.5f, 0.f), sx = fmaxf(rx * (X + 0.5f) - 0.5f, 0.f);
                int y0 = min((int)sy, hs - 1), x0 = min((int)sx, ws - 1);
                int y1 = y0 + (y0 < hs - 1), x1 = x0 + (x0 < ws - 1);
                float ly = sy - y0, lx = sx - x0;
                float g = gd[k];
                atomicAdd(&sAcc[(y0 - oy) * ACC_W + (x0 - ox)], g * (1.f - ly) * (1.f - lx));
                atomicAdd(&sAcc[(y0 - oy) * ACC_W + (x1 - ox)], g * (1.f - ly) * lx);
                atomicAdd(&sAcc[(y1 - oy) * ACC_W + (x0 - ox)], g * ly * (1.f - lx));
                atomicAdd(&sAcc[(y1 - oy) * ACC_W + (x1 - ox)], g * ly * lx);
            }
            __syncthreads();
            for (int i = tid; i < ACC_H * ACC_W; i += NTB) {
                int ay = i / ACC_W, ax = i - ay * ACC_W;
                float v = sAcc[i];
                int yy = oy + ay, xx = ox + ax;
                if (v != 0.f && yy < hs && xx < ws) atomicAdd(&dd[yy * ws + xx], v);
            }
        }
        __syncthreads();  // sD, sF/sAcc reused by the next scale
    }
}

// d_T[f][b] = K[:3,:]^T . sum_{tiles,scales} dP      (P = (K.T)[:3,:], learner_func.py:149)
__global__ __launch_bounds__(NT) void chain_bwd_reduce_kernel(ChainParams p, BwdParams q) {
    __shared__ float sRed[NT / 64][16];
    __shared__ float sP[NDP];
    const int b = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
    const int S = p.cfg.num_scales, ntiles = p.tiles_x * p.tiles_y;
    const int col = tid & 15;
    float v = 0.f;
    if (col < NDP) {
        for (int r = tid >> 4; r < ntiles * S; r += NT / 16) {
            int t = r / S, s = r - t * S;
            v += q.g.bwd_partials[((((size_t)b * ntiles + t) * S + s) * 2 + f) * NDP + col];
        }
    }
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if ((tid & 63) < 16) sRed[tid >> 6][col] = v;
    __syncthreads();
    if (tid < NDP) sP[tid] = sRed[0][tid] + sRed[1][tid] + sRed[2][tid] + sRed[3][tid];
    __syncthreads();
    if (tid < 16) {
        int k = tid >> 2, j = tid & 3;
        const float* K = p.io.K + b * 16;
        float acc = 0.f;
        for (int i = 0; i < 3; ++i) acc += K[i * 4 + k] * sP[i * 4 + j];
        q.g.d_T[f][b * 16 + tid] = acc;
    }
}

int validate(const dvs_chain_cfg* c, const char* who) {
    DVS_REQUIRE(c, "%s: null cfg", who);
    DVS_REQUIRE(c->B > 0 && c->H >= 4 && c->W >= 4, "%s: bad size B=%d H=%d W=%d", who, c->B, c->H, c->W);
    DVS_REQUIRE(c->num_scales >= 1 && c->num_scales <= DVS_MAX_SCALES, "%s: num_scales=%d", who, c->num_scales);
    for (int s = 0; s < c->num_scales; ++s) {
        DVS_REQUIRE(c->hs[s] > 0 && c->ws[s] > 0, "%s: scale %d is %dx%d", who, s, c->hs[s], c->ws[s]);
        bool same = c->hs[s] == c->H && c->ws[s] == c->W;
        DVS_REQUIRE(same || (c->hs[s] * 2 <= c->H && c->ws[s] * 2 <= c->W),
                    "%s: scale %d (%dx%d) must be full resolution or at most half of %dx%d", who, s, c->hs[s],
                    c->ws[s], c->H, c->W);
    }
    DVS_REQUIRE(c->min_depth > 0.f && c->max_depth > c->min_depth, "%s: depth range", who);
    return DVS_OK;
}

ChainParams make_params(const dvs_chain_cfg* cfg, const dvs_chain_fwd_io* io) {
    ChainParams p;
    p.cfg = *cfg;
    p.io = *io;
    p.tiles_x = (cfg->W + TW - 1) / TW;
    p.tiles_y = (cfg->H + TH - 1) / TH;
    return p;
}

}  // namespace

extern "C" {

int dvs_chain_workspace(const dvs_chain_cfg* cfg, size_t* partials_bytes, size_t* sel_bytes,
                        size_t* stats_bytes, size_t* bwd_partials_bytes) {
    int rc = validate(cfg, "dvs_chain_workspace");
    if (rc) return rc;
    size_t ntiles = (size_t)((cfg->W + TW - 1) / TW) * ((cfg->H + TH - 1) / TH);
    if (partials_bytes) *partials_bytes = (size_t)cfg->B * ntiles * NPART * sizeof(float);
    if (sel_bytes) *sel_bytes = (size_t)cfg->B * cfg->H * cfg->W;
    if (stats_bytes)
        *stats_bytes = (stats_floats(cfg->B, cfg->num_scales) + (size_t)2 * cfg->B * cfg->H * cfg->W * 4) * sizeof(float);
    if (bwd_partials_bytes) *bwd_partials_bytes = (size_t)cfg->B * ntiles * cfg->num_scales * 2 * NDP * sizeof(float);
    return DVS_OK;
}

int dvs_chain_fwd(const dvs_chain_cfg* cfg, const dvs_chain_fwd_io* io, void* stream) {
    int rc = validate(cfg, "dvs_chain_fwd");
    if (rc) return rc;
    DVS_REQUIRE(io, "dvs_chain_fwd: null io");
    DVS_REQUIRE(io->target && io->source[0] && io->source[1] && io->K && io->inv_K && io->T[0] && io->T[1],
                "dvs_chain_fwd: null input");
    DVS_REQUIRE(io->partials && io->sel && io->stats && io->losses, "dvs_chain_fwd: null workspace/output");
    for (int s = 0; s < cfg->num_scales; ++s) DVS_REQUIRE(io->disp[s], "dvs_chain_fwd: null disp[%d]", s);
    ChainParams p = make_params(cfg, io);
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(chain_cam_kernel, dim3(cfg->B), dim3(64), 0, st, p);
    hipLaunchKernelGGL(chain_pack_kernel, dim3(2048), dim3(NT), 0, st, p);
    {
        dvs::ProfScope prof(dvs::SLOT_CHAIN_FWD, st);
        hipLaunchKernelGGL(chain_fwd_kernel, dim3(p.tiles_x, p.tiles_y, cfg->B), dim3(NT), 0, st, p);
    }
    hipLaunchKernelGGL(chain_fwd_reduce_kernel, dim3(cfg->B), dim3(NT), 0, st, p);
    hipLaunchKernelGGL(chain_fwd_losses_kernel, dim3(1), dim3(64), 0, st, p);
    return dvs::check_launch("dvs_chain_fwd");
}

int dvs_chain_bwd(const dvs_chain_cfg* cfg, const dvs_chain_fwd_io* io, const dvs_chain_bwd_io* g,
                  void* stream) {
    int rc = validate(cfg, "dvs_chain_bwd");
    if (rc) return rc;
    DVS_REQUIRE(io && g, "dvs_chain_bwd: null io");
    DVS_REQUIRE(io->target && io->source[0] && io->source[1] && io->K && io->inv_K && io->T[0] && io->T[1],
                "dvs_chain_bwd: null input");
    DVS_REQUIRE(io->sel && io->stats, "dvs_chain_bwd: null forward state");
    DVS_REQUIRE(g->d_losses && g->d_T[0] && g->d_T[1] && g->bwd_partials, "dvs_chain_bwd: null gradient buffer");
    hipStream_t st = static_cast<hipStream_t>(stream);
    DVS_REQUIRE(g->phase >= 0 && g->phase <= 2 && g->scale_begin >= 0 && g->scale_end <= cfg->num_scales, "dvs_chain_bwd: bad phase / scale range");
    const int s_lo = (g->scale_end > g->scale_begin) ? g->scale_begin : 0;
    const int s_hi = (g->scale_end > g->scale_begin) ? g->scale_end : cfg->num_scales;
    for (int s = (g->phase == 2 ? cfg->num_scales : s_lo); s < s_hi; ++s) {
        DVS_REQUIRE(io->disp[s] && g->d_disp[s], "dvs_chain_bwd: null disp/d_disp[%d]", s);
        if (!(cfg->hs[s] == cfg->H && cfg->ws[s] == cfg->W)) {
            hipError_t e = hipMemsetAsync(g->d_disp[s], 0, (size_t)cfg->B * cfg->hs[s] * cfg->ws[s] * sizeof(float), st);
            if (e != hipSuccess) return dvs::fail(DVS_ERR_LAUNCH, "dvs_chain_bwd: memset: %s", hipGetErrorString(e));
        }
    }
    ChainParams p = make_params(cfg, io);
    BwdParams q;
    q.g = *g;
    if (g->phase != 2) {
        dvs::ProfScope prof(dvs::SLOT_CHAIN_BWD, st);
        hipLaunchKernelGGL((chain_bwd_kernel<BWD_NT, TH / (BWD_NT / 64)>), dim3(p.tiles_x, p.tiles_y, cfg->B), dim3(BWD_NT), 0, st, p, q);
    }
    if (g->phase != 1) hipLaunchKernelGGL(chain_bwd_reduce_kernel, dim3(cfg->B, 2), dim3(NT), 0, st, p, q);
    return dvs::check_launch("dvs_chain_bwd");
}

}  // extern "C"
