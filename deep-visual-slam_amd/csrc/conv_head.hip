// a2/a3: the narrow output heads -- DepthNet's four 3x3 "dispconv" layers (Cin 16..128 -> 1 channel, reflection
// pad, sigmoid; model/depthnet.py:57-58,86-88) and PoseNet's final 1x1 (256 -> 6; model/posenet_single.py:165).
//
// With 1..8 output channels a 32-wide MFMA tile would idle 75-97 % of the matrix core and these layers are
// pure bandwidth anyway (one pass over a [B,H,W,Cin] map), so they run on the vector ALUs:
//   forward : one lane per output pixel, weights in LDS, 16-byte channel gathers (neighbouring lanes walk
//             neighbouring pixels, so every cache line fetched is fully used by the wave);
//   dgrad   : one lane per (pixel, 4 input channels);
//   wgrad   : lanes own 4-wide k-slices, workgroups own pixel ranges, register accumulation, one atomic
//             per (workgroup, weight) at the end.
// Same NHWC layouts and [Cout][kh][kw][Cin] weights as the MFMA kernels (conv_common.h).
#include "conv_common.h"

#include <cstdlib>

namespace {
using namespace dvsconv;

constexpr int HNT = 256;
constexpr int MAXCO = 8;

struct HeadParams {
    const float* x;      // [B,H,W,Cin]
    const float* w;      // [Cout][kh][kw][Cin]
    const float* bias;
    float* y;            // [B,H,W,Cout]   (stride 1, "same" size)
    const float* dy;     // backward: gradient of y (post-activation)
    float* dx;           // [B,H,W,Cin]
    const float* dx_res; // optional [B,H,W,Cin]: another gradient of x, added to dx (dvs_conv2d_head_bwd_res)
    float* dw;           // [Cout][Ktot], atomics
    float* dbias;        // [Cout], atomics
    int B, H, W, Cin, Cout, k, pad, reflect, act;
    int pix_per_block;
};

__device__ __forceinline__ int src_index(int i, int n, int reflect, bool& ok) {
    if (reflect) return reflect_i(i, n);
    ok = ok && (unsigned)i < (unsigned)n;
    return clampi(i, n);
}

template <int COUT, int KS>
__global__ __launch_bounds__(HNT) void head_fwd_kernel(HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) float sw[];       // [COUT][Ktot]
    const int Ktot = KS * KS * p.Cin, M = p.B * p.H * p.W;
    for (int i = threadIdx.x; i < COUT * Ktot; i += HNT) sw[i] = p.w[i];
    __syncthreads();
    const int m = blockIdx.x * HNT + threadIdx.x;
    if (m >= M) return;
    const int b = m / (p.H * p.W), rem = m - b * (p.H * p.W), oy = rem / p.W, ox = rem - oy * p.W;
    // the KS*KS source pixels of this output pixel (padding resolved once); all taps of a channel slice are
    // loaded before any is consumed, so KS*KS independent 16-byte loads are in flight per lane
    const float* src[KS * KS];
    bool ok[KS * KS];
#pragma unroll
    for (int t = 0; t < KS * KS; ++t) {
        bool v = true;
        int iy = src_index(oy - p.pad + t / KS, p.H, p.reflect, v), ix = src_index(ox - p.pad + t % KS, p.W, p.reflect, v);
        ok[t] = v;
        src[t] = p.x + (((size_t)b * p.H + iy) * p.W + ix) * p.Cin;
    }
    float acc[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[c] = p.bias ? p.bias[c] : 0.f;
    for (int ci = 0; ci < p.Cin; ci += 4) {
        f32x4 v[KS * KS];
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) v[t] = *reinterpret_cast<const f32x4*>(src[t] + ci);
#pragma unroll
        for (int t = 0; t < KS * KS; ++t) {
            if (!ok[t]) continue;
#pragma unroll
            for (int c = 0; c < COUT; ++c) {
                f32x4 wv = *reinterpret_cast<const f32x4*>(sw + c * Ktot + t * p.Cin + ci);
                acc[c] = fmaf(v[t][0], wv[0], fmaf(v[t][1], wv[1], fmaf(v[t][2], wv[2], fmaf(v[t][3], wv[3], acc[c]))));
            }
        }
    }
#pragma unroll
    for (int c = 0; c < COUT; ++c) p.y[(size_t)m * COUT + c] = apply_act(acc[c], p.act);
}

// dY' = dY * act'(Y) at pixel m, all COUT channels
template <int COUT>
__device__ __forceinline__ void load_dyp(const HeadParams& p, size_t m, float (&g)[COUT]) {
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        float v = p.dy[m * COUT + c];
        if (p.act) v *= act_grad_from_out(p.y[m * COUT + c], p.act);
        g[c] = v;
    }
}

// dx[m][ci..ci+3] = sum_{tap,co} dY'[src(m,tap)][co] * w[co][tap][ci..]; reflection fold as in gather_dgrad_raw
template <int COUT>
__global__ __launch_bounds__(HNT) void head_dgrad_kernel(HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) float sw[];
    const int Ktot = p.k * p.k * p.Cin, M = p.B * p.H * p.W, cv = p.Cin / 4;
    for (int i = threadIdx.x; i < COUT * Ktot; i += HNT) sw[i] = p.w[i];
    __syncthreads();
    const size_t gid = (size_t)blockIdx.x * HNT + threadIdx.x;
    if (gid >= (size_t)M * cv) return;
    const int m = (int)(gid / cv), ci = (int)(gid % cv) * 4;
    const int b = m / (p.H * p.W), rem = m - b * (p.H * p.W), y = rem / p.W, x = rem - y * p.W;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto add_from = [&](int ty, int tx, int ky, int kx) {
        float g[COUT];
        load_dyp<COUT>(p, ((size_t)b * p.H + ty) * p.W + tx, g);
        const float* wp = sw + (ky * p.k + kx) * p.Cin + ci;
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
            f32x4 wv = *reinterpret_cast<const f32x4*>(wp + c * Ktot);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(g[c], wv[j], acc[j]);
        }
    };
    for (int ky = 0; ky < p.k; ++ky) {
        for (int kx = 0; kx < p.k; ++kx) {
            int ty = y + p.pad - ky, tx = x + p.pad - kx;
            bool in_y = (unsigned)ty < (unsigned)p.H, in_x = (unsigned)tx < (unsigned)p.W;
            if (in_y && in_x) add_from(ty, tx, ky, kx);
            if (p.reflect) {
                int ey = (y == 1 && ky == 0) ? 0 : ((y == p.H - 2 && ky == p.k - 1) ? p.H - 1 : -1);
                int ex = (x == 1 && kx == 0) ? 0 : ((x == p.W - 2 && kx == p.k - 1) ? p.W - 1 : -1);
                if (ey >= 0 && in_x) add_from(ey, tx, ky, kx);
                if (ex >= 0 && in_y) add_from(ty, ex, ky, kx);
                if (ey >= 0 && ex >= 0) add_from(ey, ex, ky, kx);
            }
        }
    }
    if (p.dx_res) acc += *reinterpret_cast<const f32x4*>(p.dx_res + (size_t)m * p.Cin + ci);
    *reinterpret_cast<f32x4*>(p.dx + (size_t)m * p.Cin + ci) = acc;
}

// Row form of the 3x3 data gradient: a workgroup owns a segment of one image row (256 / (Cin/4) pixels) and stages the
// three dY' rows it needs (activation derivative applied, zeros outside the image) in LDS once -- the per-pixel kernel
// above issues 18 scalar global loads per lane (9 taps x (dY, Y), the Cin/4 lanes of a pixel all fetching the same
// values) and was ~4x off the HBM time of its one write pass.  Same arithmetic, reflection fold included.
template <int COUT>
__global__ __launch_bounds__(HNT) void head_dgrad_rows_kernel(HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int Ktot = 9 * p.Cin, cv = p.Cin / 4, SEGW = HNT / cv, GW = SEGW + 2;
    float* sw = sm;                       // [COUT][Ktot]
    float* sg = sm + COUT * Ktot;         // [3][GW][COUT]
    for (int i = threadIdx.x; i < COUT * Ktot; i += HNT) sw[i] = p.w[i];
    const int nseg = (p.W + SEGW - 1) / SEGW;
    const int seg = blockIdx.x % nseg, row = blockIdx.x / nseg;
    const int b = row / p.H, y = row - b * p.H, x0 = seg * SEGW;
    for (int i = threadIdx.x; i < 3 * GW; i += HNT) {
        const int r = i / GW, e = i - r * GW;
        const int ty = y - 1 + r, tx = x0 - 1 + e;
        const bool in = (unsigned)ty < (unsigned)p.H && (unsigned)tx < (unsigned)p.W;
        float g[COUT];
        load_dyp<COUT>(p, ((size_t)b * p.H + clampi(ty, p.H)) * p.W + clampi(tx, p.W), g);
#pragma unroll
        for (int c = 0; c < COUT; ++c) sg[i * COUT + c] = in ? g[c] : 0.f;
    }
    __syncthreads();
    const int px = threadIdx.x / cv, ci = (threadIdx.x % cv) * 4, x = x0 + px;
    if (x >= p.W) return;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto add_from = [&](int ty, int tx, int ky, int kx) {
        const float* g = sg + ((ty - (y - 1)) * GW + (tx - (x0 - 1))) * COUT;
        const float* wp = sw + (ky * 3 + kx) * p.Cin + ci;
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
            f32x4 wv = *reinterpret_cast<const f32x4*>(wp + c * Ktot);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(g[c], wv[j], acc[j]);
        }
    };
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ty = y + 1 - ky, tx = x + 1 - kx;
            const bool in_y = (unsigned)ty < (unsigned)p.H, in_x = (unsigned)tx < (unsigned)p.W;
            if (in_y && in_x) add_from(ty, tx, ky, kx);
            if (p.reflect) {
                const int ey = (y == 1 && ky == 0) ? 0 : ((y == p.H - 2 && ky == 2) ? p.H - 1 : -1);
                const int ex = (x == 1 && kx == 0) ? 0 : ((x == p.W - 2 && kx == 2) ? p.W - 1 : -1);
                if (ey >= 0 && in_x) add_from(ey, tx, ky, kx);
                if (ex >= 0 && in_y) add_from(ty, ex, ky, kx);
                if (ey >= 0 && ex >= 0) add_from(ey, ex, ky, kx);
            }
        }
    }
    if (p.dx_res) acc += *reinterpret_cast<const f32x4*>(p.dx_res + (((size_t)b * p.H + y) * p.W + x) * p.Cin + ci);
    *reinterpret_cast<f32x4*>(p.dx + (((size_t)b * p.H + y) * p.W + x) * p.Cin + ci) = acc;
}

// dW[co][k..k+3] += sum_pixels dY'[m][co] * x[src(m, tap(k))][ci(k)..]; lanes own k-slices, a workgroup owns a
// pixel range; several pixel lanes per k-slice when Ktot/4 < 256.
template <int COUT>
__global__ __launch_bounds__(HNT) void head_wgrad_kernel(HeadParams p) {
    extern __shared__ __attribute__((aligned(16))) float sred[];     // [HNT][COUT*4] for the pixel-lane reduction
    const int Ktot = p.k * p.k * p.Cin, KV = Ktot / 4, M = p.B * p.H * p.W;
    const int plane_count = max(1, HNT / KV);                        // pixel lanes per k-slice
    const int kv_stride = (KV + HNT - 1) / HNT;                      // k-slices per lane when KV > 256
    const int tid = threadIdx.x;
    const int pl = (KV >= HNT) ? 0 : tid / KV, kv0 = (KV >= HNT) ? tid : tid % KV;
    const bool lane_active = (KV >= HNT) || (pl < plane_count);
    const int m0 = blockIdx.x * p.pix_per_block, m1 = min(M, m0 + p.pix_per_block);
    float bsum[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) bsum[c] = 0.f;
    for (int rep = 0; rep < kv_stride; ++rep) {
        const int kv = kv0 + rep * HNT;
        f32x4 acc[COUT];
#pragma unroll
        for (int c = 0; c < COUT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (lane_active && kv < KV) {
            const int k = kv * 4, tap = k / p.Cin, ci = k - tap * p.Cin, ky = tap / p.k, kx = tap - ky * p.k;
            constexpr int U = 4;                                     // pixels in flight per lane
            for (int mb = m0 + pl; mb < m1; mb += plane_count * U) {
                f32x4 v[U];
                float g[U][COUT];
                bool okv[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {                        // issue every load first
                    int m = mb + u * plane_count;
                    bool in = m < m1;
                    m = min(m, m1 - 1);
                    const int b = m / (p.H * p.W), rem = m - b * (p.H * p.W), oy = rem / p.W, ox = rem - oy * p.W;
                    load_dyp<COUT>(p, (size_t)m, g[u]);
                    bool ok = in;
                    int iy = src_index(oy - p.pad + ky, p.H, p.reflect, ok), ix = src_index(ox - p.pad + kx, p.W, p.reflect, ok);
                    okv[u] = ok;
                    v[u] = *reinterpret_cast<const f32x4*>(p.x + (((size_t)b * p.H + iy) * p.W + ix) * p.Cin + ci);
                    if (!in) {
#pragma unroll
                        for (int c = 0; c < COUT; ++c) g[u][c] = 0.f;
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (rep == 0 && kv0 == 0) {
#pragma unroll
                        for (int c = 0; c < COUT; ++c) bsum[c] += g[u][c];
                    }
                    if (!okv[u]) continue;
#pragma unroll
                    for (int c = 0; c < COUT; ++c)
#pragma unroll
                        for (int j = 0; j < 4; ++j) acc[c][j] = fmaf(g[u][c], v[u][j], acc[c][j]);
                }
            }
        }
        // reduce the pixel lanes of each k-slice through LDS, then one atomic per weight
        if (KV < HNT) {
            __syncthreads();
#pragma unroll
            for (int c = 0; c < COUT; ++c) *reinterpret_cast<f32x4*>(sred + (tid * COUT + c) * 4) = acc[c];
            __syncthreads();
            if (pl == 0 && kv < KV) {
                for (int q = 1; q < plane_count; ++q)
#pragma unroll
                    for (int c = 0; c < COUT; ++c) acc[c] += *reinterpret_cast<const f32x4*>(sred + ((tid + q * KV) * COUT + c) * 4);
            }
        }
        if (pl == 0 && kv < KV) {
#pragma unroll
            for (int c = 0; c < COUT; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(p.dw + (size_t)c * Ktot + kv * 4 + j, acc[c][j]);
        }
    }
    if (p.dbias && kv0 == 0 && lane_active) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) atomicAdd(p.dbias + c, bsum[c]);
    }
}

// ---------------------------------------------------------------------------------------------
// Row-streaming forms (Cin/4 a power of two <= 64, COUT * k * k <= 18): lanes run along the channel dimension --
// lane = (pixel slot, 4-channel chunk), a wave covers 64 / (Cin/4) neighbouring pixels -- so every tap is ONE fully
// coalesced 16-byte-per-lane load of a contiguous 1 KB run of the NHWC row, the lane's weight slices (forward) or
// gradient accumulators (weight gradient) live in registers for the whole kernel, and nothing is divided per pixel:
// a workgroup walks whole image rows.  The forward reduces a pixel's chunks with log2(Cin/4) butterfly steps; the
// weight gradient reduces pixel slots the same way, then the four waves through LDS, then one atomic per weight
// and workgroup (<= 512 workgroups: same-address float atomics serialise at ~85 ns each).
// ---------------------------------------------------------------------------------------------
template <int COUT, int KS>
__global__ __launch_bounds__(HNT) void head_fwd_rows_kernel(HeadParams p) {
    constexpr int T = KS * KS;
    const int cpl = p.Cin >> 2, ppb = HNT / cpl;                   // lanes per pixel, pixels per workgroup pass
    const int kc = (threadIdx.x % cpl) * 4, slot = threadIdx.x / cpl;
    f32x4 wr[COUT][T];
#pragma unroll
    for (int c = 0; c < COUT; ++c)
#pragma unroll
        for (int t = 0; t < T; ++t) wr[c][t] = *reinterpret_cast<const f32x4*>(p.w + ((size_t)c * T + t) * p.Cin + kc);
    float bv[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) bv[c] = p.bias ? p.bias[c] : 0.f;
    // a workgroup walks a run of CONSECUTIVE rows: the two rows it shares with the next step come from its own L1 / L2 (rows dealt
    // round-robin were fetched by three workgroups, i.e. three XCDs: 3x the HBM reads)
    const int rows = p.B * p.H, per = (rows + (int)gridDim.x - 1) / (int)gridDim.x;
    const int row_end = min(rows, ((int)blockIdx.x + 1) * per);
    for (int row = blockIdx.x * per; row < row_end; ++row) {
        const int b = row / p.H, oy = row - b * p.H;
        const float* rp[KS];
        bool rok[KS];
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            bool v = true;
            int iy = src_index(oy - p.pad + ky, p.H, p.reflect, v);
            rok[ky] = v;
            rp[ky] = p.x + ((size_t)b * p.H + iy) * p.W * p.Cin + kc;
        }
        for (int ox0 = 0; ox0 < p.W; ox0 += ppb) {
            const int ox = ox0 + slot;
            const bool px_ok = ox < p.W;
            f32x4 v[T];
            bool ok[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {                              // all taps in flight before any is used
                bool vx = px_ok && rok[t / KS];
                int ix = src_index(min(ox, p.W - 1) - p.pad + t % KS, p.W, p.reflect, vx);
                ok[t] = vx;
                v[t] = *reinterpret_cast<const f32x4*>(rp[t / KS] + (size_t)ix * p.Cin);
            }
            float acc[COUT];
#pragma unroll
            for (int c = 0; c < COUT; ++c) acc[c] = 0.f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float m = ok[t] ? 1.f : 0.f;
#pragma unroll
                for (int c = 0; c < COUT; ++c)
                    acc[c] = fmaf(m * v[t][0], wr[c][t][0], fmaf(m * v[t][1], wr[c][t][1],
                             fmaf(m * v[t][2], wr[c][t][2], fmaf(m * v[t][3], wr[c][t][3], acc[c]))));
            }
            for (int off = cpl >> 1; off >= 1; off >>= 1)
#pragma unroll
                for (int c = 0; c < COUT; ++c) acc[c] += __shfl_xor(acc[c], off, 64);
            if (kc == 0 && px_ok) {
                float* yp = p.y + ((size_t)row * p.W + ox) * COUT;
#pragma unroll
                for (int c = 0; c < COUT; ++c) yp[c] = apply_act(acc[c] + bv[c], p.act);
            }
        }
    }
}

template <int COUT, int KS>
__global__ __launch_bounds__(HNT) void head_wgrad_rows_kernel(HeadParams p) {
    constexpr int T = KS * KS;
    extern __shared__ __attribute__((aligned(16))) float sred[];       // [4 waves][cpl][COUT*T] float4 (+ bias sums)
    const int cpl = p.Cin >> 2, ppb = HNT / cpl;
    const int kq = threadIdx.x % cpl, kc = kq * 4, slot = threadIdx.x / cpl;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x4 acc[COUT][T];
    float bsum[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) {
        bsum[c] = 0.f;
#pragma unroll
        for (int t = 0; t < T; ++t) acc[c][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int rows = p.B * p.H;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int b = row / p.H, oy = row - b * p.H;
        const float* rp[KS];
        bool rok[KS];
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
            bool v = true;
            int iy = src_index(oy - p.pad + ky, p.H, p.reflect, v);
            rok[ky] = v;
            rp[ky] = p.x + ((size_t)b * p.H + iy) * p.W * p.Cin + kc;
        }
        for (int ox0 = 0; ox0 < p.W; ox0 += ppb) {
            const int ox = ox0 + slot;
            const bool px_ok = ox < p.W;
            const int oxc = min(ox, p.W - 1);
            f32x4 v[T];
            float m[T];
#pragma unroll
            for (int t = 0; t < T; ++t) {
                bool vx = px_ok && rok[t / KS];
                int ix = src_index(oxc - p.pad + t % KS, p.W, p.reflect, vx);
                m[t] = vx ? 1.f : 0.f;
                v[t] = *reinterpret_cast<const f32x4*>(rp[t / KS] + (size_t)ix * p.Cin);
            }
            float g[COUT];
            load_dyp<COUT>(p, (size_t)row * p.W + oxc, g);
#pragma unroll
            for (int c = 0; c < COUT; ++c) {
                g[c] = px_ok ? g[c] : 0.f;
                bsum[c] += g[c];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float gm = g[c] * m[t];
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[c][t][j] = fmaf(gm, v[t][j], acc[c][t][j]);
                }
            }
        }
    }
    // pixel slots of one wave: butterfly over the lane bits above the chunk index
    for (int off = cpl; off < 64; off <<= 1) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
            bsum[c] += __shfl_xor(bsum[c], off, 64);
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[c][t][j] += __shfl_xor(acc[c][t][j], off, 64);
        }
    }
    // the four waves through LDS (when a wave holds less than one pixel slot per chunk, cpl == 64, lanes map 1:1)
    const int per_wave = cpl * COUT * T;                               // float4 entries
    if (lane < cpl) {
#pragma unroll
        for (int c = 0; c < COUT; ++c)
#pragma unroll
            for (int t = 0; t < T; ++t)
                *reinterpret_cast<f32x4*>(sred + ((size_t)wave * per_wave + (c * T + t) * cpl + lane) * 4) = acc[c][t];
    }
    float* sb = sred + (size_t)4 * per_wave * 4;                       // [4][COUT] bias partials
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < COUT; ++c) sb[wave * COUT + c] = bsum[c];
    }
    __syncthreads();
    const int Ktot = T * p.Cin;
    for (int e = threadIdx.x; e < per_wave; e += HNT) {                // e = (c*T + t)*cpl + chunk
        f32x4 sum = *reinterpret_cast<const f32x4*>(sred + (size_t)e * 4);
#pragma unroll
        for (int w = 1; w < 4; ++w) sum += *reinterpret_cast<const f32x4*>(sred + ((size_t)w * per_wave + e) * 4);
        const int chunk = e % cpl, ct = e / cpl, t = ct % T, c = ct / T;
        float* dst = p.dw + (size_t)c * Ktot + t * p.Cin + chunk * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) atomicAdd(dst + j, sum[j]);
    }
    if (p.dbias && threadIdx.x < COUT)
        atomicAdd(p.dbias + threadIdx.x, sb[threadIdx.x] + sb[COUT + threadIdx.x] + sb[2 * COUT + threadIdx.x] + sb[3 * COUT + threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------
// Input-indexed weight gradient of the one-channel 3x3 heads (round 3).  The row form above is indexed by the OUTPUT pixel: nine
// 16-byte loads per lane and pixel, every x row fetched by three workgroups (three L2s): 274 MB of HBM reads per launch for 90 MB of
// tensors, 105 us.  Re-indexed by the INPUT pixel,
//     dw[ky][kx][ci] = sum over (iy, ix) of x(iy, ix, ci) * g(iy - ky + 1, ix - kx + 1),      g = dY * act'(Y), zero outside the image,
// x is read exactly once (one coalesced 1 KB run per wave and row) and what a lane gathers is the 3x3 window of the ONE-channel g
// -- dwords that stay in L1.  A wave owns 64 / (Cin/4) columns and walks a run of rows: the window slides in registers, one new row of
// three g values per step.  ReflectionPad2d(1) folds into the window: output row 0 reads input row 1 through ky = 0, so at iy = 1
// tap ky = 0 also takes what ky = 2 takes (g row 0), at iy = H - 2 tap ky = 2 takes g row H - 1; columns likewise, the column fold
// applied once per loaded row.  Persistent waves (wave tiles dealt round-robin), 16 waves per workgroup which add into one LDS copy
// of the gradient, one atomic per weight and workgroup (256 workgroups).
// ---------------------------------------------------------------------------------------------
constexpr int WNT = 1024, PF = 3;
// DG: the DATA gradient on the same walk (round 3, last): dx(iy, ix, c) = sum over taps of g'(tap) * w[tap][c] with the same folded
// window, the lane's nine weight vectors in registers, one coalesced 16-byte store per lane and row (+ the optional fan-in gradient
// dx_res).  The row form (head_dgrad_rows_kernel) launches one 256-thread workgroup per 64-pixel row segment -- 57 600 of them at
// scale 0 -- and ran at 2.6x the time of its one write pass.
template <int ACTG, bool DG = false>       // ACTG: activation whose derivative scales dY (0: none)
__global__ __launch_bounds__(WNT) void head_wgrad_in_kernel(HeadParams p, int R, int n_rr, int n_cg) {
    extern __shared__ __attribute__((aligned(16))) float sred[];       // [9][Cin] + 1
    const int cpl = p.Cin >> 2, wpw = 64 / cpl;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = lane % cpl, slot = lane / cpl;
    const int H = p.H, W = p.W, Cin = p.Cin;
    const bool refl = p.reflect != 0;
    f32x4 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    f32x4 wv[DG ? 9 : 1];
    if constexpr (DG) {
#pragma unroll
        for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(p.w + (size_t)t * Cin + kq * 4);
    }
    using u32x4s = __attribute__((ext_vector_type(4))) unsigned;
    const int ntile = p.B * n_rr * n_cg, nwave = gridDim.x * (WNT / 64);
#pragma unroll 1
    for (int tile = blockIdx.x * (WNT / 64) + wave; tile < ntile; tile += nwave) {
        const int cg = tile % n_cg, rr = (tile / n_cg) % n_rr, b = tile / (n_cg * n_rr);
        const int ix = cg * wpw + slot;
        const bool col_ok = ix < W;
        const int ixc = min(ix, W - 1);
        // columns of the three g values of a row: b = 0 <-> ox = ix + 1, b = 1 <-> ix, b = 2 <-> ix - 1; zero outside the image
        const float m0 = (col_ok && ix + 1 < W) ? 1.f : 0.f, m1 = col_ok ? 1.f : 0.f, m2 = (col_ok && ix >= 1) ? 1.f : 0.f;
        const int c0 = min(ix + 1, W - 1), c2 = max(ixc - 1, 0);
        const float f0 = (refl && ix == 1) ? 1.f : 0.f, f2 = (refl && ix == W - 2) ? 1.f : 0.f;      // column folds
        // one descriptor per tensor and image, 32-bit lane offsets that never change during the run, the row as the scalar offset
        const size_t img = (size_t)b * H * W;
        const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy + img), 0, H * W * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.y + img), 0, H * W * 4, 0x00020000);
        // (DG: the ring's x slot carries dx_res; without one the descriptor is empty and the loads return zeros)
        const float* xsrc = DG ? (p.dx_res ? p.dx_res : p.dy) : p.x;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xsrc + img * Cin), 0,
                                                                             (DG && !p.dx_res) ? 0 : H * W * Cin * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t or_ = __builtin_amdgcn_make_buffer_rsrc(DG ? p.dx + img * Cin : nullptr, 0, DG ? H * W * Cin * 4 : 0, 0x00020000);
        const unsigned o0 = (unsigned)c0 * 4u, o1 = (unsigned)ixc * 4u, o2 = (unsigned)c2 * 4u, ox = (unsigned)(ixc * Cin + kq * 4) * 4u;
        const int grow_b = W * 4, xrow_b = W * Cin * 4;
        // Row gy of g at my three columns in two halves: the LOADS (raw dY, Y; what the ring keeps in flight) and the arithmetic on
        // them (activation derivative, masks, column fold) at the step that consumes the row -- arithmetic next to the load would make
        // the loop wait for its youngest load.  No branches either (a load behind a branch costs a full vmcnt(0) wait at the join):
        // rows outside the image read a clamped row and are multiplied by zero.
        struct Raw { float d[3], y[ACTG != 0 ? 3 : 1]; };
        auto gload = [&](int gy, Raw& o) {
            const int rb = min(max(gy, 0), H - 1) * grow_b;
            o.d[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dr, o0, rb, 0));
            o.d[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dr, o1, rb, 0));
            o.d[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dr, o2, rb, 0));
            if constexpr (ACTG != 0) {
                o.y[0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yr, o0, rb, 0));
                o.y[1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yr, o1, rb, 0));
                o.y[2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yr, o2, rb, 0));
            }
        };
        // rm: 1 if row gy exists and the step that consumes it is inside the run (both wave-uniform), else 0
        auto gfin = [&](const Raw& r, float rm, float (&o)[3]) {
            float d0 = r.d[0], d1 = r.d[1], d2 = r.d[2];
            if constexpr (ACTG != 0) {
                d0 *= act_grad_from_out(r.y[0], ACTG);
                d1 *= act_grad_from_out(r.y[1], ACTG);
                d2 *= act_grad_from_out(r.y[2], ACTG);
            }
            d0 *= m0 * rm; d1 *= m1 * rm; d2 *= m2 * rm;
            o[0] = fmaf(f0, d2, d0);
            o[1] = d1;
            o[2] = fmaf(f2, d0, d2);
        };
        const int iy0 = rr * R, iy1 = min(iy0 + R, H);
        auto xload = [&](int iy) { return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, ox, min(iy, H - 1) * xrow_b, 0)); };
        auto exists = [&](int gy) { return (gy >= 0 && gy < H) ? 1.f : 0.f; };
        // PF rows of x and of g in flight: ring slot k holds x row iy + k and g row iy + 1 + k of the current group of PF steps; a slot
        // is refilled at the end of the step that consumed it, so the loads of the next PF rows fly under this group's arithmetic.
        // Steps past the end of the run read clamped rows and see a zero window.
        float prev[3], cur[3];
        {
            Raw a, c;
            gload(iy0 - 1, a);
            gload(iy0, c);
            gfin(a, exists(iy0 - 1), prev);
            gfin(c, 1.f, cur);
        }
        f32x4 xq[PF];
        Raw gq[PF];
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            gload(iy0 + 1 + k, gq[k]);
            xq[k] = xload(iy0 + k);
        }
#pragma unroll 1
        for (int iyg = iy0; iyg < iy1; iyg += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int iy = iyg + k;
                const float live = iy < iy1 ? 1.f : 0.f;
                float next[3];
                gfin(gq[k], exists(iy + 1), next);
                // window rows by tap: ky = 0 <-> g row iy + 1, ky = 1 <-> iy, ky = 2 <-> iy - 1; row folds on the ORIGINAL rows
                float g0[3], g1[3], g2[3];
                const float r0 = (refl && iy == 1) ? live : 0.f, r2 = (refl && iy == H - 2) ? live : 0.f;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    g0[j] = fmaf(r0, prev[j], next[j] * live);
                    g1[j] = cur[j] * live;
                    g2[j] = fmaf(r2, next[j], prev[j] * live);
                }
                if constexpr (DG) {
                    f32x4 o = xq[k];                     // dx_res (zeros without one)
#pragma unroll
                    for (int j = 0; j < 3; ++j) o += wv[0 * 3 + j] * g0[j] + wv[1 * 3 + j] * g1[j] + wv[2 * 3 + j] * g2[j];
                    // rows past the run / columns past the image: an offset beyond the tensor, the store is dropped
                    const unsigned so = (iy < iy1 && col_ok) ? ox : 0x80000000u;
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4s, o), or_, so, min(iy, H - 1) * xrow_b, 0);
                } else {
                    bsum += g1[1];                       // g at (iy, ix): masked by m1, and the column folds do not touch b = 1
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        acc[0 * 3 + j] += xq[k] * g0[j];
                        acc[1 * 3 + j] += xq[k] * g1[j];
                        acc[2 * 3 + j] += xq[k] * g2[j];
                    }
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    prev[j] = cur[j];
                    cur[j] = next[j];
                }
                xq[k] = xload(iy + PF);
                gload(iy + 1 + PF, gq[k]);
                // keep the steps' loads in program order: the counter waits are in-order, and a scheduler that clusters the loads of
                // several steps ends up waiting for the youngest one
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    if constexpr (DG) return;
    // pixel slots of a wave: butterfly over the lane bits above the chunk index
    for (int off = cpl; off < 64; off <<= 1) {
        bsum += __shfl_xor(bsum, off, 64);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[t][j] += __shfl_xor(acc[t][j], off, 64);
    }
    // the sixteen waves add into one LDS copy of the gradient (ds_add_f32), then one atomic per weight and workgroup
    const int nw = 9 * Cin;
    for (int e = threadIdx.x; e <= nw; e += WNT) sred[e] = 0.f;        // [9][Cin] + the bias sum
    __syncthreads();
    if (lane < cpl) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(sred + t * Cin + lane * 4 + j, acc[t][j]);
    }
    if (lane == 0) atomicAdd(sred + nw, bsum);
    __syncthreads();
#ifdef HEAD_DBG_NOATOM          // timing cut (tools/build_variant.py --flag=-DHEAD_DBG_NOATOM): wrong results by construction
    if (sred[0] == 12345.678f)
#endif
    for (int e = threadIdx.x; e < nw; e += WNT) atomicAdd(p.dw + e, sred[e]);
    if (p.dbias && threadIdx.x == 0) atomicAdd(p.dbias, sred[nw]);
}

// Forward of the one-channel 3x3 heads on the same walk: a wave owns 64 / (Cin/4) columns and walks down a run of rows with the 3x3
// window of x in registers (nine 16-byte vectors per lane): a new output row costs THREE loads per lane (the row below, at the lane's
// column and its two neighbours') instead of nine, padding is resolved in the addresses (reflection) or by masks (zeros); the
// pixel's Cin/4 partial sums meet in a butterfly and chunk 0 stores bias + activation.
constexpr int FNT = 256;       // (the window takes 60 registers more than the gradient kernels' state: three waves per SIMD as three four-wave workgroups per CU)
template <int ACTF>
__global__ __launch_bounds__(FNT) void head_fwd_walk_kernel(HeadParams p, int R, int n_rr, int n_cg) {
    const int cpl = p.Cin >> 2, wpw = 64 / cpl;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int kq = lane % cpl, slot = lane / cpl;
    const int H = p.H, W = p.W, Cin = p.Cin;
    const bool refl = p.reflect != 0;
    f32x4 wv[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) wv[t] = *reinterpret_cast<const f32x4*>(p.w + (size_t)t * Cin + kq * 4);
    const float bias = p.bias ? p.bias[0] : 0.f;
    const int ntile = p.B * n_rr * n_cg, nwave = gridDim.x * (FNT / 64);
#pragma unroll 1
    for (int tile = blockIdx.x * (FNT / 64) + wave; tile < ntile; tile += nwave) {
        const int cg = tile % n_cg, rr = (tile / n_cg) % n_rr, b = tile / (n_cg * n_rr);
        const int ix = cg * wpw + slot;
        const bool col_ok = ix < W;
        const int ixc = min(ix, W - 1);
        // source columns of the taps kx = 0, 1, 2 and their masks (zero padding only: a reflected column always exists)
        const int cl = refl ? reflect_i(ixc - 1, W) : max(ixc - 1, 0), cr = refl ? reflect_i(ixc + 1, W) : min(ixc + 1, W - 1);
        const float ml = (refl || ixc >= 1) ? 1.f : 0.f, mr = (refl || ixc + 1 < W) ? 1.f : 0.f;
        const size_t img = (size_t)b * H * W;
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + img * Cin), 0, H * W * Cin * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y + img, 0, H * W * 4, 0x00020000);
        const unsigned ol = (unsigned)(cl * Cin + kq * 4) * 4u, oc = (unsigned)(ixc * Cin + kq * 4) * 4u, orr = (unsigned)(cr * Cin + kq * 4) * 4u;
        const int xrow_b = W * Cin * 4, yrow_b = W * 4;
        struct Row { f32x4 v[3]; };
        // x row sy at my three columns; sy is the padded row: reflected into the image, or (zero padding) clamped and masked at use
        auto rload = [&](int sy, Row& o) {
            const int r = refl ? reflect_i(min(max(sy, -1), H), H) : min(max(sy, 0), H - 1);
            o.v[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, ol, r * xrow_b, 0));
            o.v[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, oc, r * xrow_b, 0));
            o.v[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, orr, r * xrow_b, 0));
        };
        auto rowmask = [&](int sy) { return (refl || (sy >= 0 && sy < H)) ? 1.f : 0.f; };
        const int oy0 = rr * R, oy1 = min(oy0 + R, H);
        Row top, mid, rq[PF];
        rload(oy0 - 1, top);
        rload(oy0, mid);
        float mt = rowmask(oy0 - 1);                      // mask of the window's top row (the middle row always exists)
#pragma unroll
        for (int k = 0; k < PF; ++k) rload(oy0 + 1 + k, rq[k]);
#pragma unroll 1
        for (int oyg = oy0; oyg < oy1; oyg += PF) {
#pragma unroll
            for (int k = 0; k < PF; ++k) {
                const int oy = oyg + k;
                const Row bot = rq[k];
                const float mb = rowmask(oy + 1);
                f32x4 a = f32x4{0.f, 0.f, 0.f, 0.f};
                // taps (ky, kx): rows top / mid / bot, columns left / centre / right; column masks ride on the weights' side
                a += (top.v[0] * wv[0] * ml + top.v[1] * wv[1] + top.v[2] * wv[2] * mr) * mt;
                a += mid.v[0] * wv[3] * ml + mid.v[1] * wv[4] + mid.v[2] * wv[5] * mr;
                a += (bot.v[0] * wv[6] * ml + bot.v[1] * wv[7] + bot.v[2] * wv[8] * mr) * mb;
                float sum = (a[0] + a[1]) + (a[2] + a[3]);
                for (int off = cpl >> 1; off >= 1; off >>= 1) sum += __shfl_xor(sum, off, 64);
                const float yv = apply_act(sum + bias, ACTF);
                const unsigned so = (oy < oy1 && col_ok && kq == 0) ? (unsigned)ixc * 4u : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv), yr, so, min(oy, H - 1) * yrow_b, 0);
                top = mid;
                mid = bot;
                mt = 1.f;
                rload(oy + 1 + PF, rq[k]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

// One workgroup per CU (fewer measured slower at every scale: tools/head_bench.py with DVS_HEAD_WGRAD_WGS); rows per wave tile: the run
// length in [8, 32] with the cheapest schedule -- rounds of the persistent waves x (R + about half a step of extra g rows).
template <bool DG = false, bool FWD = false>
inline void launch_wgrad_in(const HeadParams& p, hipStream_t st) {
    static const int forced = [] { const char* e = getenv("DVS_HEAD_WGRAD_WGS"); return e ? atoi(e) : 0; }();
    const int cpl = p.Cin / 4, wpw = 64 / cpl, n_cg = (p.W + wpw - 1) / wpw;
    const int G = forced > 0 ? forced : 256;
    const long waves = (long)G * (WNT / 64);
    int bestR = p.H < 8 ? p.H : 8;
    double best = 1e30;
    for (int R = bestR; R <= 32 && R <= p.H; ++R) {
        const long tiles = (long)p.B * ((p.H + R - 1) / R) * n_cg;
        const long rounds = (tiles + waves - 1) / waves;
        const double t = (double)rounds * (R + 0.5);
        if (t <= best) { best = t; bestR = R; }
    }
    const int n_rr = (p.H + bestR - 1) / bestR;
    const long tiles = (long)p.B * n_rr * n_cg;
    int grid = (int)((tiles + WNT / 64 - 1) / (WNT / 64));
    if (grid > G) grid = G;
    const size_t lds = ((size_t)9 * p.Cin + 1) * sizeof(float);
    if constexpr (FWD) {
        int fgrid = (int)((tiles + FNT / 64 - 1) / (FNT / 64));
        if (fgrid > 3 * G) fgrid = 3 * G;
        if (p.act == ACT_SIGMOID) hipLaunchKernelGGL(head_fwd_walk_kernel<ACT_SIGMOID>, dim3(fgrid), dim3(FNT), 0, st, p, bestR, n_rr, n_cg);
        else hipLaunchKernelGGL(head_fwd_walk_kernel<ACT_NONE>, dim3(fgrid), dim3(FNT), 0, st, p, bestR, n_rr, n_cg);
        return;
    }
    if (p.act == ACT_SIGMOID) hipLaunchKernelGGL((head_wgrad_in_kernel<ACT_SIGMOID, DG>), dim3(grid), dim3(WNT), lds, st, p, bestR, n_rr, n_cg);
    else hipLaunchKernelGGL((head_wgrad_in_kernel<0, DG>), dim3(grid), dim3(WNT), lds, st, p, bestR, n_rr, n_cg);
}

inline bool rows_form_ok(const HeadParams& p, int cout) {
    static const bool enabled = [] { const char* e = getenv("DVS_HEAD_ROWS"); return !(e && e[0] == '0'); }();
    const int cpl = p.Cin / 4;
    return enabled && cpl >= 1 && cpl <= 64 && (cpl & (cpl - 1)) == 0 && cout * p.k * p.k <= 18;
}

template <int COUT, int KS>
void launch_rows(const HeadParams& p, int op, hipStream_t st) {
    const int rows = p.B * p.H;
    if constexpr (COUT * KS * KS <= 18) {
        if (op == 0) {
            static const int fwd_wgs = [] { const char* e = getenv("DVS_HEAD_FWD_WGS"); return e ? atoi(e) : 1024; }();
            const int per = (rows + fwd_wgs - 1) / fwd_wgs;
            hipLaunchKernelGGL((head_fwd_rows_kernel<COUT, KS>), dim3((rows + per - 1) / per), dim3(HNT), 0, st, p);
        } else {
            const size_t lds = ((size_t)4 * (p.Cin / 4) * COUT * KS * KS * 4 + 4 * COUT) * sizeof(float);
            // every workgroup ends with COUT * k * k * Cin same-address atomics: a small map (PoseNet's 1x1 head: 24 x 15 x 20 pixels
            // of 256 channels) gets one workgroup per 32 K elements, not one per row (360 workgroups there: 56 us, most of it atomics)
            const long elems = (long)rows * p.W * p.Cin;
            int wgs = (int)(elems >> 15);
            wgs = wgs < 16 ? 16 : (wgs > 512 ? 512 : wgs);
            hipLaunchKernelGGL((head_wgrad_rows_kernel<COUT, KS>), dim3(rows < wgs ? rows : wgs), dim3(HNT), lds, st, p);
        }
    }
}

template <int COUT>
int run(const HeadParams& p0, int op, hipStream_t st) {
    HeadParams p = p0;
    const int Ktot = p.k * p.k * p.Cin, M = p.B * p.H * p.W;
    const size_t wbytes = (size_t)COUT * Ktot * sizeof(float);
    if (op != 1 && rows_form_ok(p, COUT)) {
        static const bool in_form = [] { const char* e = getenv("DVS_HEAD_WGRAD_IN"); return !(e && e[0] == '0'); }();
        static const bool fwd_walk = [] { const char* e = getenv("DVS_HEAD_FWD_WALK"); return !(e && e[0] == '0'); }();
        const bool walk_ok = COUT == 1 && p.k == 3 && p.pad == 1 && (p.act == ACT_NONE || p.act == ACT_SIGMOID) &&
                             (size_t)p.H * p.W * p.Cin * 4 < ((size_t)1 << 31);
        if (op == 2 && walk_ok && in_form) launch_wgrad_in(p, st);
        else if (op == 0 && walk_ok && fwd_walk && (!p.reflect || (p.H >= 2 && p.W >= 2))) launch_wgrad_in<false, true>(p, st);
        else if (p.k == 3) launch_rows<COUT, 3>(p, op, st);
        else launch_rows<COUT, 1>(p, op, st);
        return DVS_OK;
    }
    if (op == 0) {
        if (p.k == 3) hipLaunchKernelGGL((head_fwd_kernel<COUT, 3>), dim3((M + HNT - 1) / HNT), dim3(HNT), wbytes, st, p);
        else hipLaunchKernelGGL((head_fwd_kernel<COUT, 1>), dim3((M + HNT - 1) / HNT), dim3(HNT), wbytes, st, p);
    } else if (static const bool dg_in = [] { const char* e = getenv("DVS_HEAD_DGRAD_IN"); return !(e && e[0] == '0'); }();
               op == 1 && COUT == 1 && p.k == 3 && p.pad == 1 && dg_in && rows_form_ok(p, COUT) && (p.act == 0 || p.act == ACT_SIGMOID) &&
               (size_t)p.H * p.W * p.Cin * 4 < ((size_t)1 << 31)) {
        launch_wgrad_in<true>(p, st);
    } else if (op == 1 && p.k == 3 && p.pad == 1 && COUT <= 2 && rows_form_ok(p, COUT) && p.H >= 3 && p.W >= 3) {
        const int segw = HNT / (p.Cin / 4), nseg = (p.W + segw - 1) / segw;
        const size_t lds = wbytes + (size_t)3 * (segw + 2) * COUT * sizeof(float);
        hipLaunchKernelGGL(head_dgrad_rows_kernel<COUT>, dim3((unsigned)(p.B * p.H * nseg)), dim3(HNT), lds, st, p);
    } else if (op == 1) {
        size_t n = (size_t)M * (p.Cin / 4);
        hipLaunchKernelGGL(head_dgrad_kernel<COUT>, dim3((unsigned)((n + HNT - 1) / HNT)), dim3(HNT), wbytes, st, p);
    } else {
        int blocks = (M + 511) / 512;
        if (blocks > 512) blocks = 512;      // every workgroup ends with one atomic per weight: keep same-address traffic low
        p.pix_per_block = (M + blocks - 1) / blocks;
        blocks = (M + p.pix_per_block - 1) / p.pix_per_block;
        hipLaunchKernelGGL(head_wgrad_kernel<COUT>, dim3(blocks), dim3(HNT), (size_t)HNT * COUT * 4 * sizeof(float), st, p);
    }
    return DVS_OK;
}

int dispatch(const HeadParams& p, int op, hipStream_t st) {
    switch (p.Cout) {
        case 1: return run<1>(p, op, st);
        case 2: return run<2>(p, op, st);
        case 6: return run<6>(p, op, st);
        case 8: return run<8>(p, op, st);
        default: return dvs::fail(DVS_ERR_UNSUPPORTED, "dvs_conv2d_head: Cout=%d (1, 2, 6, 8 supported)", p.Cout);
    }
}

int fill(HeadParams& p, const dvs_conv_desc* d, int act, const char* who) {
    DVS_REQUIRE(d, "%s: null descriptor", who);
    DVS_REQUIRE(d->stride == 1 && d->kh == d->kw && 2 * d->pad == d->kh - 1 && (d->kh == 1 || d->kh == 3),
                "%s: stride-1 'same' 1x1 / 3x3 convolutions only", who);
    DVS_REQUIRE((d->Cin & 3) == 0 && d->Cout <= MAXCO, "%s: Cin %% 4 == 0 and Cout <= %d", who, MAXCO);
    DVS_REQUIRE((size_t)d->Cout * d->kh * d->kw * d->Cin * 4 <= 60 * 1024, "%s: weights must fit 60 KB of LDS", who);
    DVS_REQUIRE(d->pad_mode == PAD_ZERO || (d->H >= 2 && d->W >= 2), "%s: reflection needs H, W >= 2", who);
    p.B = d->B; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Cout = d->Cout; p.k = d->kh; p.pad = d->pad;
    p.reflect = d->pad_mode == PAD_REFLECT; p.act = act;
    return DVS_OK;
}

}  // namespace

extern "C" {

int dvs_conv2d_head_fwd(const float* x, const float* w, const float* bias, float* y, const dvs_conv_desc* d, int act,
                        void* stream) {
    DVS_REQUIRE(x && w && y, "dvs_conv2d_head_fwd: null pointer");
    HeadParams p{};
    int rc = fill(p, d, act, "dvs_conv2d_head_fwd");
    if (rc) return rc;
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    rc = dispatch(p, 0, static_cast<hipStream_t>(stream));
    return rc ? rc : dvs::check_launch("dvs_conv2d_head_fwd");
}

int dvs_conv2d_head_bwd(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw,
                        float* dbias, const dvs_conv_desc* d, int act, void* stream) {
    return dvs_conv2d_head_bwd_res(x, w, y, dy, dx, dw, dbias, d, act, nullptr, stream);
}

int dvs_conv2d_head_bwd_res(const float* x, const float* w, const float* y, const float* dy, float* dx, float* dw,
                            float* dbias, const dvs_conv_desc* d, int act, const float* dx_residual, void* stream) {
    DVS_REQUIRE(x && w && y && dy && dw, "dvs_conv2d_head_bwd: null pointer");
    DVS_REQUIRE(!dx_residual || (dx && dx_residual != dx), "dvs_conv2d_head_bwd: the residual needs dx and may not alias it");
    HeadParams p{};
    int rc = fill(p, d, act, "dvs_conv2d_head_bwd");
    if (rc) return rc;
    p.x = x; p.w = w; p.y = const_cast<float*>(y); p.dy = dy; p.dx = dx; p.dw = dw; p.dbias = dbias; p.dx_res = dx_residual;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dx) {
        rc = dispatch(p, 1, st);
        if (rc) return rc;
    }
    rc = dispatch(p, 2, st);
    return rc ? rc : dvs::check_launch("dvs_conv2d_head_bwd");
}

}  // extern "C"
