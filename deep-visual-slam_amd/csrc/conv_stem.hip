// a1: the encoder stems -- Conv2d(3 or 6 -> 64, 7x7, stride 2, padding 3, no bias) on the planar [B,Cin,H,W] image
// (model/resnet_encoder.py:102-103; the (x - 0.45) / 0.225 normalisation of :141 folded into the read).
//
// The generic implicit-GEMM kernels gather this operand with four scalar loads and a dozen VALU instructions per
// 16 bytes and spend most of the stage on it (23-35 TF measured).  Here a workgroup stages the raw image patch of
// 2 x 64 output pixels ONCE into LDS (Cin x 9 rows x 133 columns, normalised, zero-padded) and every im2col element
// is a plain ds_read_b32 at  patch[tap offset (per lane) + pixel offset (an immediate)]:  no address arithmetic in
// the K loop, no barrier inside a tile, K ordered (ci, ky, kx8) with the 8th kx column multiplying nothing.
//   weight gradient: dW[co][k] += sum_pixels dY[p][co] * patch[p][k]   (M = 64 channels, N = Cin*56, K = pixels):
//                    A from a [128 px][64] LDS tile of dY, B from the patch, accumulators live in registers across
//                    all tiles of a persistent workgroup, one atomic per weight and workgroup at the end.
// LDS bank behaviour: a patch row is 136 floats (136 mod 64 = 8), so the 32 lanes of a B read -- 4 ky rows x 8 kx --
// fall on 32 distinct banks; the dY read is 32 consecutive floats.
#include "conv_common.h"

#include <cstdlib>

namespace {
using namespace dvsconv;

constexpr int SNT = 256;
constexpr int PROWS = 9, PSTRIDE = 136, PCOLS = 133;      // patch of 2 output rows x 64 output columns
constexpr int TPX = 128;                                   // pixels per tile

struct StemParams {
    const float* x;        // [B,Cin,H,W] planar
    const float* dy;       // [B,Ho,Wo,64]
    float* dw;             // [64][Cin][7][8] packed, atomics
    const float* sc;       // per-channel input scale / shift (NULL = identity)
    const float* sh;
    int B, H, W, Ho, Wo;
    int row_pairs, col_tiles, tiles;
    int dbg;               // timing experiments (DVS_STEM_DEBUG): 1 = no epilogue atomics, 2 = stage only the first tile
};

// Patch staging in two passes -- every global load of the tile is issued before the first LDS store -- so the ~15
// (Cin 3) / ~29 (Cin 6) loads of a thread overlap instead of paying one memory round trip each.
template <int CIN>
__device__ __forceinline__ void load_patch(const StemParams& p, float* patch, int b, int ry, int cx) {
    constexpr int N = CIN * PROWS * PSTRIDE, U = (N + SNT - 1) / SNT;
    float scv[CIN], shv[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
        scv[c] = p.sc ? p.sc[c] : 1.f;
        shv[c] = p.sc ? p.sh[c] : 0.f;
    }
    float v[U];
    int ci_of[U];
    bool ok[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = min((int)threadIdx.x + u * SNT, N - 1);
        const int i = e % PSTRIDE, j = (e / PSTRIDE) % PROWS, ci = e / (PSTRIDE * PROWS);
        const int iy = 4 * ry - 3 + j, ix = 128 * cx - 3 + i;
        ok[u] = i < PCOLS && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
        ci_of[u] = ci;
        const int iyc = min(max(iy, 0), p.H - 1), ixc = min(max(ix, 0), p.W - 1);
        v[u] = p.x[(((size_t)b * CIN + ci) * p.H + iyc) * p.W + ixc];       // always a valid address: no branch around the load
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int e = (int)threadIdx.x + u * SNT;
        float sc = scv[0], sh = shv[0];
#pragma unroll
        for (int c = 1; c < CIN; ++c) {
            sc = ci_of[u] == c ? scv[c] : sc;
            sh = ci_of[u] == c ? shv[c] : sh;
        }
        if (e < N) patch[e] = ok[u] ? v[u] * sc + sh : 0.f;
    }
}

template <int CIN>
__global__ __launch_bounds__(SNT) void stem_wgrad_kernel(StemParams p) {
    constexpr int KT = CIN * 56;                        // (ci, ky, kx8)
    constexpr int NT32 = (KT + 31) / 32;                // 32-wide N tiles: 6 / 11
    constexpr int TNW = (NT32 + 1) / 2;                 // per wave (2 waves along N): 3 / 6
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;                                // [CIN][9][136]
    float* dyt = smem + CIN * PROWS * PSTRIDE;          // [128][64]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    // per-lane patch offset of my column n of each N tile (clamped past the end: those results are not stored)
    int koff[TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t) {
        int n = min((wn * TNW + t) * 32 + r32, KT - 1);
        const int kx = n & 7, row = n >> 3, ci = row / 7, ky = row - ci * 7;
        koff[t] = (ci * PROWS + ky) * PSTRIDE + kx + 2 * h;          // + 2h: the odd pixel of a k-step is one column on
    }
    const int a_off = h * 64 + wm * 32 + r32;                         // dY tile: [pixel][channel]

    f32x16 acc[TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int cx = tile % p.col_tiles, rest = tile / p.col_tiles, ry = rest % p.row_pairs, b = rest / p.row_pairs;
        if (!(p.dbg & 2) || tile == (int)blockIdx.x) load_patch<CIN>(p, patch, b, ry, cx);
        if (!(p.dbg & 2) || tile == (int)blockIdx.x) {
            // dY tile: clamped (always valid) addresses, all eight 16-byte loads in flight, zeros selected afterwards --
            // a bounds branch around each load would cost one memory round trip per load
            constexpr int DU = TPX * 16 / SNT;
            f32x4 dv[DU];
            bool dok[DU];
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                const int q = threadIdx.x + u * SNT, px = q >> 4, c4 = q & 15;
                const int oy = 2 * ry + (px >> 6), ox = 64 * cx + (px & 63);
                dok[u] = oy < p.Ho && ox < p.Wo;
                dv[u] = *reinterpret_cast<const f32x4*>(p.dy + (((size_t)b * p.Ho + min(oy, p.Ho - 1)) * p.Wo + min(ox, p.Wo - 1)) * 64 + c4 * 4);
            }
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                const int q = threadIdx.x + u * SNT;
                f32x4 v = dv[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = dok[u] ? v[e] : 0.f;
                *reinterpret_cast<f32x4*>(dyt + q * 4) = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < TPX / 2; ++s) {
            // pixels 2s, 2s+1: output row r = (2s) / 64, column c = (2s) % 64 (+ h) -> patch origin (2r, 2c)
            constexpr int dummy = 0;
            (void)dummy;
            const int pix_off = (2 * ((2 * s) >> 6)) * PSTRIDE + 2 * ((2 * s) & 63);
            const float a = dyt[(2 * s) * 64 + a_off];
            float bv[TNW];
#pragma unroll
            for (int t = 0; t < TNW; ++t) bv[t] = patch[koff[t] + pix_off];
#pragma unroll
            for (int t = 0; t < TNW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[t], acc[t], 0, 0, 0);
        }
        __syncthreads();
    }
    // D map of the 32x32 MFMA: column n = lane & 31, rows (i & 3) + 8 * (i >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int t = 0; t < TNW; ++t) {
        const int n = (wn * TNW + t) * 32 + r32;
        if (n >= KT || (p.dbg & 1)) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            atomicAdd(p.dw + (size_t)co * KT + n, acc[t][i]);
        }
    }
}

template <int CIN>
void launch_wgrad(StemParams p, hipStream_t st) {
    const size_t lds = ((size_t)CIN * PROWS * PSTRIDE + TPX * 64) * sizeof(float);
    auto kern = stem_wgrad_kernel<CIN>;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024 - 256) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int per_cu = (int)(160 * 1024 / lds);
    int blocks = 256 * (per_cu < 1 ? 1 : (per_cu > 3 ? 3 : per_cu));
    if (blocks > p.tiles) blocks = p.tiles;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(SNT), lds, st, p);
}

}  // namespace

namespace dvsconv {

bool stem_enabled() {
    static const bool on = [] { const char* e = getenv("DVS_CONV_STEM"); return !(e && e[0] == '0'); }();
    return on;
}

// true when the shape is the ResNet stem this file specialises
bool stem_shape(const ConvShape& s) {
    return stem_enabled() && s.kh == 7 && s.kw == 7 && s.stride == 2 && s.pad == 3 && s.pad_mode == PAD_ZERO && s.Cout == 64 &&
           (s.Cin == 3 || s.Cin == 6);
}

void stem_wgrad(const float* x, const float* dy, float* dw, const ConvShape& s, const float* sc, const float* sh,
                hipStream_t st) {
    StemParams p{};
    p.x = x; p.dy = dy; p.dw = dw; p.sc = sc; p.sh = sh;
    p.B = s.B; p.H = s.H; p.W = s.W; p.Ho = s.Ho; p.Wo = s.Wo;
    p.row_pairs = (s.Ho + 1) / 2;
    p.col_tiles = (s.Wo + 63) / 64;
    p.tiles = s.B * p.row_pairs * p.col_tiles;
    static const int dbg = getenv("DVS_STEM_DEBUG") ? atoi(getenv("DVS_STEM_DEBUG")) : 0;
    p.dbg = dbg;
    if (s.Cin == 3) launch_wgrad<3>(p, st);
    else launch_wgrad<6>(p, st);
}

}  // namespace dvsconv
