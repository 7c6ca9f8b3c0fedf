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
template <int CIN, class T = float>        // T = __bf16: the bf16 mode's patch image
__device__ __forceinline__ void load_patch(const StemParams& p, T* patch, int b, int ry, int cx) {
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
        if (e < N) patch[e] = (T)(ok[u] ? v[u] * sc + sh : 0.f);
    }
}

// Everything about a thread's patch elements that does not depend on the tile is computed once: the element's
// offset from the tile's image origin, its (row, column) inside the patch for the bounds test, and its channel's
// normalisation.  Per tile and element that leaves two compares, a select and the load.
template <int CIN>
struct PatchPlan {
    static constexpr int N = CIN * PROWS * PSTRIDE, U = (N + SNT - 1) / SNT;
    int rel[U];            // ci * H * W + j * W + i
    int ji[U];             // j << 16 | i   (i >= PCOLS marks a padding column: never valid)
    float sc[U], sh[U];
    __device__ __forceinline__ void init(int H, int W, const float* scp, const float* shp) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = min((int)threadIdx.x + u * SNT, N - 1);
            const int i = e % PSTRIDE, j = (e / PSTRIDE) % PROWS, ci = e / (PSTRIDE * PROWS);
            rel[u] = (ci * H + j) * W + i;
            ji[u] = (j << 16) | (i < PCOLS ? i : 0x7fff);
            sc[u] = scp ? scp[ci] : 1.f;
            sh[u] = scp ? shp[ci] : 0.f;
        }
    }
};

template <int CIN>
struct PatchRegs {
    static constexpr int N = CIN * PROWS * PSTRIDE, U = (N + SNT - 1) / SNT;
    float v[U];
    unsigned okmask;       // bit u: element u is inside the image (and a real patch column)
};

template <int CIN>
__device__ __forceinline__ void fetch_patch(const float* __restrict__ x, int H, int W, int b, int ry, int cx,
                                            const PatchPlan<CIN>& pl, PatchRegs<CIN>& r) {
    const int iy0 = 4 * ry - 3, ix0 = 128 * cx - 3;
    const float* base = x + ((size_t)b * CIN * H + iy0) * (ptrdiff_t)W + ix0;      // may point before the row: only valid taps are read
    r.okmask = 0;
#pragma unroll
    for (int u = 0; u < PatchRegs<CIN>::U; ++u) {
        const int j = pl.ji[u] >> 16, i = pl.ji[u] & 0xffff;
        const bool ok = (unsigned)(iy0 + j) < (unsigned)H && (unsigned)(ix0 + i) < (unsigned)W;
        r.okmask |= ok ? (1u << u) : 0u;
        const float* a = ok ? base + pl.rel[u] : x;                                  // always a valid address
        r.v[u] = *a;
    }
}

template <int CIN, class T = float>       // T = __bf16: the bf16 mode's patch image (rounded here, once per element)
__device__ __forceinline__ void store_patch(T* patch, const PatchRegs<CIN>& r, const PatchPlan<CIN>& pl) {
#pragma unroll
    for (int u = 0; u < PatchRegs<CIN>::U; ++u) {
        const int e = (int)threadIdx.x + u * SNT;
        if (e < PatchRegs<CIN>::N) patch[e] = (T)(((r.okmask >> u) & 1u) ? r.v[u] * pl.sc[u] + pl.sh[u] : 0.f);
    }
}

template <int CIN>
__global__ __launch_bounds__(SNT) void stem_wgrad_kernel(StemParams p) {
    constexpr int KT = CIN * 56;                        // row of dw: (ci, ky, kx8), the kx = 7 column is never written (stays zero)
    constexpr int KD = CIN * 49;                        // the GEMM's N: the real (ci, ky, kx) columns, dense -- 147 / 294
    constexpr int NT32 = (KD + 31) / 32;                // 32-wide N tiles: 5 / 10 (the padded rows took 6 / 11, i.e. 6 / 12 with two waves along N)
    constexpr int TNW = (NT32 + 1) / 2;                 // per wave (2 waves along N): 3 / 5
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* patch = smem;                                // [CIN][9][136]
    float* dyt = smem + CIN * PROWS * PSTRIDE;          // [128][64]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    // per-lane patch offset of my column n of each N tile (clamped past the end: those results are not stored) and its place in dw
    int koff[TNW], kdst[TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t) {
        const int n = (wn * TNW + t) * 32 + r32, nc = min(n, KD - 1);
        const int ci = nc / 49, rem = nc - ci * 49, ky = rem / 7, kx = rem - ky * 7;
        koff[t] = (ci * PROWS + ky) * PSTRIDE + kx + 2 * h;          // + 2h: the odd pixel of a k-step is one column on
        kdst[t] = n < KD ? (ci * 7 + ky) * 8 + kx : -1;
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
        if (kdst[t] < 0 || (p.dbg & 1)) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            atomicAdd(p.dw + (size_t)co * KT + kdst[t], acc[t][i]);
        }
    }
}

// bf16 mode: the same GEMM on v_mfma_f32_32x32x16_bf16, a k-step = 16 consecutive pixels of an output row.  Both operands are
// k-strided in their LDS images: A = dY comes through the transposed read (ds_read_b64_tr_b16) from a bf16 [pixel][64 + 32] tile
// (192-byte rows: the four pixel rows a half-wave reads land on disjoint banks); B[k = pixel][column = tap] is every SECOND element
// of a bf16 patch row (stride 2 in the image), so a lane reads the eight dwords that hold its eight elements and keeps their low or
// high halves -- by the parity of its tap's kx -- with four v_perm_b32.
using s16x4w = __attribute__((ext_vector_type(4))) short;
using s16x8w = __attribute__((ext_vector_type(8))) short;
using u32x4w = __attribute__((ext_vector_type(4))) unsigned;
template <int CIN>
__global__ __launch_bounds__(SNT, 2) void stem_wgrad_bf16_kernel(StemParams p) {
    constexpr int KT = CIN * 56, KD = CIN * 49, NT32 = (KD + 31) / 32, TNW = (NT32 + 1) / 2;
    constexpr int LDD = 96;                              // dY tile row (elements): 64 channels + 32 of padding
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* patch = reinterpret_cast<__bf16*>(smem);    // [CIN][9][136]
    __bf16* dyt = patch + CIN * PROWS * PSTRIDE;        // [128][LDD]   (CIN * 9 * 136 * 2 bytes is a multiple of 16)

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    int kword[TNW], kdst[TNW];
    unsigned ksel[TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t) {
        const int n = (wn * TNW + t) * 32 + r32, nc = min(n, KD - 1);
        const int ci = nc / 49, rem = nc - ci * 49, ky = rem / 7, kx = rem - ky * 7;
        const int e0 = (ci * PROWS + ky) * PSTRIDE + kx + 16 * h;       // my first element of a k-step at patch origin 0: pixels 8 h .. 8 h + 7
        kword[t] = e0 >> 1;                                             // the dword that holds it
        ksel[t] = (e0 & 1) ? 0x07060302u : 0x05040100u;                 // v_perm_b32: high / low halves of two dwords
        kdst[t] = n < KD ? (ci * 7 + ky) * 8 + kx : -1;
    }
    // transposed read of dY (see conv_wgrad.hip): pixel (8 h' + q) of the k-step, channels 16 (g & 1) + 4 pp ..
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int trow = 8 * (g >> 1) + q, tcol = wm * 32 + 16 * (g & 1) + 4 * pp;

    f32x16 acc[TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    for (int tile = blockIdx.x; tile < p.tiles; tile += gridDim.x) {
        const int cx = tile % p.col_tiles, rest = tile / p.col_tiles, ry = rest % p.row_pairs, b = rest / p.row_pairs;
        load_patch<CIN, __bf16>(p, patch, b, ry, cx);
        {
            constexpr int DU = TPX * 16 / SNT;
            f32x4 dv[DU];
            bool dok[DU];
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                const int qq = threadIdx.x + u * SNT, px = qq >> 4, c4 = qq & 15;
                const int oy = 2 * ry + (px >> 6), ox = 64 * cx + (px & 63);
                dok[u] = oy < p.Ho && ox < p.Wo;
                dv[u] = *reinterpret_cast<const f32x4*>(p.dy + (((size_t)b * p.Ho + min(oy, p.Ho - 1)) * p.Wo + min(ox, p.Wo - 1)) * 64 + c4 * 4);
            }
#pragma unroll
            for (int u = 0; u < DU; ++u) {
                const int qq = threadIdx.x + u * SNT, px = qq >> 4, c4 = qq & 15;
                f32x4 v = dv[u];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = dok[u] ? v[e] : 0.f;
                *reinterpret_cast<bf16x4*>(dyt + px * LDD + c4 * 4) = to_bf16(v);
            }
        }
        __syncthreads();
        const unsigned* pw = reinterpret_cast<const unsigned*>(patch);
#pragma unroll
        for (int s = 0; s < TPX / 16; ++s) {
            // pixels 16 s .. 16 s + 15: output row (16 s) / 64, columns (16 s) % 64 .. -> patch origin (2 row, 2 column)
            const int pix_word = ((2 * ((16 * s) >> 6)) * PSTRIDE + 2 * ((16 * s) & 63)) >> 1;
            const __bf16* dbase = dyt + (16 * s + trow) * LDD + tcol;
            using lds_ptr = __attribute__((address_space(3))) s16x4w*;
            const s16x4w alo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(dbase));
            const s16x4w ahi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(dbase + 4 * LDD));
            const bf16x8 a = __builtin_bit_cast(bf16x8, s16x8w{alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]});
#pragma unroll
            for (int t = 0; t < TNW; ++t) {
                const unsigned* w8 = pw + kword[t] + pix_word;
                u32x4w pk;
#pragma unroll
                for (int j = 0; j < 4; ++j) pk[j] = __builtin_amdgcn_perm(w8[2 * j + 1], w8[2 * j], ksel[t]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, pk), acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < TNW; ++t) {
        if (kdst[t] < 0) continue;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int co = wm * 32 + (i & 3) + 8 * (i >> 2) + 4 * h;
            atomicAdd(p.dw + (size_t)co * KT + kdst[t], acc[t][i]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Forward: y[p][co] = sum_k patch[p][k] * w[co][k]  (M = pixels, N = 64 channels, K = Cin*56).
// The packed weights stay resident in LDS as [k][64] for the whole persistent workgroup; the image patch is
// double-buffered: the next tile's pixels are fetched into registers before the MFMAs of the current tile and
// written to the other buffer after them, so one barrier per tile is all the synchronisation there is.
// A wave owns 64 pixels x 32 channels (two 32x32 accumulators): per k-pair two patch reads and one weight read,
// all at   per-lane base + immediate offset.   Epilogue: NHWC store and the BatchNorm statistics of bn1
// (per-channel sum and sum of squares, one atomic pair per channel and wave at the very end).
// ---------------------------------------------------------------------------------------------
struct StemFwdParams {
    const float* x;        // [B,Cin,H,W] planar
    const float* w;        // [64][Cin][7][8] packed, kx = 7 column zero
    float* y;              // [B,Ho,Wo,64]
    float* stats;          // [2][64] (or [2][2][64] with two batch groups) or NULL
    int b_split;           // images >= b_split are counted into the second statistics set (B: one set)
    const float* sc;
    const float* sh;
    int B, H, W, Ho, Wo;
    int row_pairs, col_tiles, tiles;
    int dbg;               // timing experiments (DVS_STEM_DEBUG): 1 = no output stores, 2 = no patch re-staging, 4 = no MFMAs
};

template <int CIN>
__global__ __launch_bounds__(SNT) void stem_fwd_kernel(StemFwdParams p) {
    // K runs over the real (ci, ky, kx) taps, dense: 147 / 294 instead of the packed rows' 168 / 336 (the 8th kx column multiplied a
    // zero weight: one MFMA in eight); a k-step's two taps k = 2s + h differ irregularly between the lane halves, so the patch
    // offset is a select between two immediates (two vector instructions per 128 cycles of MFMA).
    constexpr int KP = CIN * 56, KD = CIN * 49, KS = (KD + 1) / 2, PN = CIN * PROWS * PSTRIDE;
    static_assert(PatchRegs<CIN>::U <= 32, "okmask is 32 bits");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* wl = smem;                    // [2 KS][64]
    float* patch0 = smem + 2 * KS * 64;  // [2][CIN][9][136]

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    for (int e = threadIdx.x; e < 2 * KS * 64; e += SNT) {       // wl[n][co] = w[co][(ci, ky, kx8)] of dense tap n (0 past the end)
        const int n = e >> 6, co = e & 63;
        const int ci = n / 49, rem = n - ci * 49, ky = rem / 7, kx = rem - ky * 7;
        wl[e] = n < KD ? p.w[co * KP + (ci * 7 + ky) * 8 + kx] : 0.f;
    }
    PatchPlan<CIN> plan;
    plan.init(p.H, p.W, p.sc, p.sh);
    // per-lane bases: my two pixel rows of the patch (tile rows wm, columns tm*32 + r32) and my weight column
    const int a_base0 = (2 * wm) * PSTRIDE + 2 * r32, a_base1 = a_base0 + 64;
    const int b_base = h * 64 + wn * 32 + r32;

    float ssum = 0.f, ssq = 0.f, ssum1 = 0.f, ssq1 = 0.f;
    int tile = blockIdx.x;
    PatchRegs<CIN> pr;
    if (tile < p.tiles) {
        const int cx = tile % p.col_tiles, rest = tile / p.col_tiles;
        fetch_patch<CIN>(p.x, p.H, p.W, rest / p.row_pairs, rest % p.row_pairs, cx, plan, pr);
        store_patch<CIN>(patch0, pr, plan);
    }
    __syncthreads();
    int cur = 0;
    for (; tile < p.tiles; tile += gridDim.x) {
        const int cx = tile % p.col_tiles, rest = tile / p.col_tiles, ry = rest % p.row_pairs, b = rest / p.row_pairs;
        const int nxt = tile + gridDim.x;
        if (nxt < p.tiles && !(p.dbg & 2)) {
            const int ncx = nxt % p.col_tiles, nrest = nxt / p.col_tiles;
            fetch_patch<CIN>(p.x, p.H, p.W, nrest / p.row_pairs, nrest % p.row_pairs, ncx, plan, pr);   // in flight under the MFMAs
        }
        const float* patch = patch0 + cur * PN;
        f32x16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
        // k = 2s + h -> dense tap (ci, ky, kx).  Operands are fetched two k-steps ahead into a rotating register set and the order
        // is pinned (sched_barrier): left to itself the compiler reuses one register set and waits for every ds_read right in
        // front of its MFMA.
        auto koff_of = [](int n) {                           // patch offset of dense tap n (a tap past the end reads offset 0: its weight is 0)
            const int nc = n < KD ? n : 0;
            const int ci = nc / 49, rem = nc - ci * 49, ky = rem / 7, kx = rem - ky * 7;
            return (ci * PROWS + ky) * PSTRIDE + kx;
        };
        auto lane_off = [&](int s) { return h ? koff_of(2 * s + 1) : koff_of(2 * s); };
        constexpr int AHEAD = 2;
        float ra0[AHEAD + 1], ra1[AHEAD + 1], rb[AHEAD + 1];
#pragma unroll
        for (int s = 0; s < AHEAD; ++s) {
            const int o = lane_off(s);
            ra0[s] = patch[a_base0 + o];
            ra1[s] = patch[a_base1 + o];
            rb[s] = wl[b_base + 2 * s * 64];
        }
        if (!(p.dbg & 4))
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s + AHEAD < KS) {
                const int o = lane_off(s + AHEAD);
                ra0[(s + AHEAD) % (AHEAD + 1)] = patch[a_base0 + o];
                ra1[(s + AHEAD) % (AHEAD + 1)] = patch[a_base1 + o];
                rb[(s + AHEAD) % (AHEAD + 1)] = wl[b_base + 2 * (s + AHEAD) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra0[s % (AHEAD + 1)], rb[s % (AHEAD + 1)], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(ra1[s % (AHEAD + 1)], rb[s % (AHEAD + 1)], acc1, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        // D map: column n = lane & 31 (channel), rows (i & 3) + 8 * (i >> 2) + 4 * h (pixel of the 32-pixel sub-tile)
        const int oy = 2 * ry + wm, co = wn * 32 + r32;
        float ts = 0.f, tq = 0.f;                       // this tile's statistics (one image, hence one group)
        if (oy < p.Ho && !(p.dbg & 1)) {
            float* yrow = p.y + ((size_t)b * p.Ho + oy) * p.Wo * 64 + co;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c0 = (i & 3) + 8 * (i >> 2) + 4 * h;
                const int ox0 = 64 * cx + c0, ox1 = ox0 + 32;
                if (ox0 < p.Wo) {
                    yrow[(size_t)ox0 * 64] = acc0[i];
                    ts += acc0[i];
                    tq += acc0[i] * acc0[i];
                }
                if (ox1 < p.Wo) {
                    yrow[(size_t)ox1 * 64] = acc1[i];
                    ts += acc1[i];
                    tq += acc1[i] * acc1[i];
                }
            }
        }
        if (b >= p.b_split) {
            ssum1 += ts;
            ssq1 += tq;
        } else {
            ssum += ts;
            ssq += tq;
        }
        if (nxt < p.tiles && !(p.dbg & 2)) store_patch<CIN>(patch0 + (cur ^ 1) * PN, pr, plan);
        __syncthreads();
        cur ^= 1;
    }
    if (p.stats) {
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (h == 0) {
            atomicAdd(p.stats + wn * 32 + r32, ssum);
            atomicAdd(p.stats + 64 + wn * 32 + r32, ssq);
        }
        if (p.b_split < p.B) {
            ssum1 += __shfl_xor(ssum1, 32, 64);
            ssq1 += __shfl_xor(ssq1, 32, 64);
            if (h == 0) {
                atomicAdd(p.stats + 128 + wn * 32 + r32, ssum1);
                atomicAdd(p.stats + 192 + wn * 32 + r32, ssq1);
            }
        }
    }
}

// bf16 mode (dvs_set_precision(1)): the same persistent kernel, patch staging and epilogue; the K loop runs over the packed
// (ci, ky) ROWS of eight kx each -- lane half h takes row 2 s + h of 16-k step s -- on v_mfma_f32_32x32x16_bf16: the A operand is the
// eight consecutive elements of a row of the patch, which is kept as bf16 here (4-byte aligned: four ds_read_b32), the B operand comes
// from a bf16 image of the weights [step][co][16] (one ds_read_b128).  11 / 21 steps of two MFMAs per wave and tile instead of
// 74 / 147: the kernel is then the write of its 64-channel output (0.47 GB for PoseNet's 24 images).
template <int CIN>
__global__ __launch_bounds__(SNT, 2) void stem_fwd_bf16_kernel(StemFwdParams p) {
    constexpr int KP = CIN * 56, ROWS = CIN * 7, KS = (ROWS + 1) / 2, PN = CIN * PROWS * PSTRIDE;
    static_assert(PatchRegs<CIN>::U <= 32, "okmask is 32 bits");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __bf16* wl = reinterpret_cast<__bf16*>(smem);          // [KS][64][16]: element j < 8: row 2 s, kx j; j >= 8: row 2 s + 1, kx j - 8
    __bf16* patch0 = wl + KS * 64 * 16;                    // [2][CIN][9][136] bf16

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wm = wave >> 1, wn = wave & 1;
    const int r32 = lane & 31, h = lane >> 5;
    for (int e = threadIdx.x; e < KS * 64 * 16; e += SNT) {
        const int j = e & 15, co = (e >> 4) & 63, st = e >> 10;
        const int row = 2 * st + (j >> 3), kx = j & 7;
        wl[e] = (__bf16)(row < ROWS ? p.w[co * KP + row * 8 + kx] : 0.f);
    }
    PatchPlan<CIN> plan;
    plan.init(p.H, p.W, p.sc, p.sh);
    const int a_base0 = (2 * wm) * PSTRIDE + 2 * r32, a_base1 = a_base0 + 64;
    const int b_base = (wn * 32 + r32) * 16 + 8 * h;

    float ssum = 0.f, ssq = 0.f, ssum1 = 0.f, ssq1 = 0.f;
    int tile = blockIdx.x;
    PatchRegs<CIN> pr;
    if (tile < p.tiles) {
        const int cx = tile % p.col_tiles, rest = tile / p.col_tiles;
        fetch_patch<CIN>(p.x, p.H, p.W, rest / p.row_pairs, rest % p.row_pairs, cx, plan, pr);
        store_patch<CIN, __bf16>(patch0, pr, plan);
    }
    __syncthreads();
    int cur = 0;
    for (; tile < p.tiles; tile += gridDim.x) {
        const int cx = tile % p.col_tiles, rest = tile / p.col_tiles, ry = rest % p.row_pairs, b = rest / p.row_pairs;
        const int nxt = tile + gridDim.x;
        if (nxt < p.tiles) {
            const int ncx = nxt % p.col_tiles, nrest = nxt / p.col_tiles;
            fetch_patch<CIN>(p.x, p.H, p.W, nrest / p.row_pairs, nrest % p.row_pairs, ncx, plan, pr);   // in flight under the MFMAs
        }
        const __bf16* patch = patch0 + cur * PN;
        f32x16 acc0, acc1;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
        auto roff_of = [](int row) {                       // patch offset of packed row (ci, ky); a row past the end reads row 0 (its weights are 0)
            const int rc = row < ROWS ? row : 0;
            return ((rc / 7) * PROWS + rc % 7) * PSTRIDE;
        };
        using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
        auto frag = [&](int base) __attribute__((always_inline)) {                 // (base is even: 4-byte aligned words)
            const unsigned* q = reinterpret_cast<const unsigned*>(patch + base);
            return __builtin_bit_cast(bf16x8, u32x4{q[0], q[1], q[2], q[3]});
        };
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int o = h ? roff_of(2 * s + 1) : roff_of(2 * s);
            const bf16x8 a0 = frag(a_base0 + o), a1 = frag(a_base1 + o);
            const bf16x8 bw = *reinterpret_cast<const bf16x8*>(wl + s * 64 * 16 + b_base);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bw, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bw, acc1, 0, 0, 0);
        }
        // D map: column n = lane & 31 (channel), rows (i & 3) + 8 * (i >> 2) + 4 * h (pixel of the 32-pixel sub-tile)
        const int oy = 2 * ry + wm, co = wn * 32 + r32;
        float ts = 0.f, tq = 0.f;
        if (oy < p.Ho) {
            float* yrow = p.y + ((size_t)b * p.Ho + oy) * p.Wo * 64 + co;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c0 = (i & 3) + 8 * (i >> 2) + 4 * h;
                const int ox0 = 64 * cx + c0, ox1 = ox0 + 32;
                if (ox0 < p.Wo) {
                    yrow[(size_t)ox0 * 64] = acc0[i];
                    ts += acc0[i];
                    tq += acc0[i] * acc0[i];
                }
                if (ox1 < p.Wo) {
                    yrow[(size_t)ox1 * 64] = acc1[i];
                    ts += acc1[i];
                    tq += acc1[i] * acc1[i];
                }
            }
        }
        if (b >= p.b_split) {
            ssum1 += ts;
            ssq1 += tq;
        } else {
            ssum += ts;
            ssq += tq;
        }
        if (nxt < p.tiles) store_patch<CIN, __bf16>(patch0 + (cur ^ 1) * PN, pr, plan);
        __syncthreads();
        cur ^= 1;
    }
    if (p.stats) {
        ssum += __shfl_xor(ssum, 32, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (h == 0) {
            atomicAdd(p.stats + wn * 32 + r32, ssum);
            atomicAdd(p.stats + 64 + wn * 32 + r32, ssq);
        }
        if (p.b_split < p.B) {
            ssum1 += __shfl_xor(ssum1, 32, 64);
            ssq1 += __shfl_xor(ssq1, 32, 64);
            if (h == 0) {
                atomicAdd(p.stats + 128 + wn * 32 + r32, ssum1);
                atomicAdd(p.stats + 192 + wn * 32 + r32, ssq1);
            }
        }
    }
}

template <int CIN>
void launch_fwd_bf16(StemFwdParams p, hipStream_t st) {
    const size_t lds = (size_t)((CIN * 7 + 1) / 2) * 64 * 16 * 2 + (size_t)2 * CIN * PROWS * PSTRIDE * 2;
    auto kern = stem_fwd_bf16_kernel<CIN>;
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

template <int CIN>
void launch_fwd(StemFwdParams p, hipStream_t st) {
    const size_t lds = ((size_t)2 * ((CIN * 49 + 1) / 2) * 64 + 2 * CIN * PROWS * PSTRIDE) * sizeof(float);
    auto kern = stem_fwd_kernel<CIN>;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024 - 256) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    const int per_cu = (int)(160 * 1024 / lds);
    int blocks = 256 * (per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu));
    if (blocks > p.tiles) blocks = p.tiles;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(SNT), lds, st, p);
}

template <int CIN>
void launch_wgrad_bf16(StemParams p, hipStream_t st) {
    const size_t lds = ((size_t)CIN * PROWS * PSTRIDE + TPX * 96) * 2;
    auto kern = stem_wgrad_bf16_kernel<CIN>;
    const int per_cu = (int)(160 * 1024 / lds);
    int blocks = 256 * (per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu));
    if (blocks > p.tiles) blocks = p.tiles;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(SNT), lds, st, p);
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
    static const int cap = [] { const char* e = getenv("DVS_STEM_WGRAD_PER_CU"); return e ? atoi(e) : 3; }();
    const int per_cu = (int)(160 * 1024 / lds);
    int blocks = 256 * (per_cu < 1 ? 1 : (per_cu > cap ? cap : per_cu));
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
    static const int dbg = dvs::experiment_flags("DVS_STEM_DEBUG");
    p.dbg = dbg;
    static const bool w16 = [] { const char* e = getenv("DVS_BF16_STEM_WGRAD"); return !(e && e[0] == '0'); }();
    if (dvs::precision_bf16() && w16) {
        if (s.Cin == 3) launch_wgrad_bf16<3>(p, st);
        else launch_wgrad_bf16<6>(p, st);
        return;
    }
    if (s.Cin == 3) launch_wgrad<3>(p, st);
    else launch_wgrad<6>(p, st);
}

void stem_fwd(const float* x, const float* w, float* y, float* stats, int stat_groups, const ConvShape& s,
              const float* sc, const float* sh, hipStream_t st) {
    StemFwdParams p{};
    p.x = x; p.w = w; p.y = y; p.stats = stats; p.sc = sc; p.sh = sh;
    p.b_split = stat_groups == 2 ? s.B / 2 : s.B;
    p.B = s.B; p.H = s.H; p.W = s.W; p.Ho = s.Ho; p.Wo = s.Wo;
    p.row_pairs = (s.Ho + 1) / 2;
    p.col_tiles = (s.Wo + 63) / 64;
    p.tiles = s.B * p.row_pairs * p.col_tiles;
    static const int dbg = dvs::experiment_flags("DVS_STEM_DEBUG");
    p.dbg = dbg;
    static const bool fwd16 = [] { const char* e = getenv("DVS_BF16_STEM_FWD"); return !(e && e[0] == '0'); }();
    if (dvs::precision_bf16() && fwd16) {
        if (s.Cin == 3) launch_fwd_bf16<3>(p, st);
        else launch_fwd_bf16<6>(p, st);
        return;
    }
    if (s.Cin == 3) launch_fwd<3>(p, st);
    else launch_fwd<6>(p, st);
}

}  // namespace dvsconv
