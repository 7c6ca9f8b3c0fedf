// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of the ResNet encoder (62 % of the path's flops) on the fp32 matrix
// cores -- experiment of round 2 (DESIGN.md section 7 has the analysis; tools/wino_bench.py the measurement).
//
//   Y = A^T [ (G g G^T) o (B^T d B) ] A        per 2x2 output tile, 4x4 input tile d, 3x3 filter g
//
// 16 products per 4 outputs instead of 36: 2.25x fewer MFMAs.  The 16 transform components xi are 16 independent GEMMs
// [tiles x Cin] x [Cin x Cout]; a wave keeps all 16 accumulator tiles (32 tiles x 32 output channels each = 256 registers,
// one wave per SIMD) so that the output transform is in-lane: in the C/D map of v_mfma_f32_32x32x2_f32 a lane holds output
// channel (lane & 31) of 16 tiles, the same register index in all 16 accumulators.
//   A operand (xi): V_xi[tile][c] = (B^T d B)_xi, transformed once per workgroup and chunk of channels into LDS (see the kernel);
//   B operand (xi): U_xi[c][cout] = (G g G^T)_xi, transformed once per optimiser step into [Cin][4][Cout][4] (wino_weights_kernel)
//                   and read straight from L2 (four coalesced float4 per lane and k-step), prefetched one k-step ahead.
// Data gradient of the same convolution = the same kernel on dY with the filter rotated by 180 degrees and transposed
// (flip != 0 in the weight transform).
#include "common.h"
#include <cstdlib>
#include <type_traits>

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int NT = 256;
constexpr int kActRelu = 1, kActElu = 2;   // activation codes of conv_common.h (dvs_conv_fusion.act)
constexpr int CIN_MULT = 16;             // input channels per LDS stage of the widest configuration

struct WinoParams {
    const float* x;        // [B][H][W][Cin]
    const float* u;        // [Cin][4][Cout][4]
    const float* bias;     // [Cout] or null
    float* y;              // [B][H][W][Cout]
    float* stats;          // null, or [G][2][Cout] += sum / sum of squares of the raw output (BatchNorm statistics)
    int B, H, W, Cin, Cout;
    int tiles_x, tiles_y;  // channel blocks, tile blocks (set by the launcher)
    int relu;
    int stat_split;        // images [0, stat_split) -> group 0, the rest -> group 1 (0x7fffffff: one group)
    // general gather (GEN > 0): the decoder's ReflectionPad2d(1) + [nearest 2x upsample of x (+ concat with x2)] and the
    // full correlation of the padded-domain data gradient
    const float* x2;       // second source, channels [C1, Cin) at full resolution, or null
    int C1;                // channels taken from x
    int up;                // x is [B][H/2][W/2][C1]: nearest-neighbour upsampled in the gather
    int reflect;           // out-of-image patch pixels mirror (pad 1) instead of reading zero
    int Ho, Wo, org;       // output size and patch origin: output tile (2ty, 2tx) reads input rows 2ty - org ... (1: 'same', 2: full)
    int act;               // epilogue activation code of conv_common.h (0 none, 1 ReLU, 2 ELU)
    const float* res;      // null, or a tensor of y's shape added to the convolution (before bias / activation)
    int stat_mask, stat_stride;   // statistics slots: workgroup w adds into stats + (w & stat_mask) * stat_stride (0, 0: one copy)
};

// w [Cout][3][3][Cin] -> u [Cin][4][Cout][4] (flip = 0), or the data-gradient filter: u [Cout][4][Cin][4] from w rotated 180 degrees
__global__ __launch_bounds__(NT) void wino_weights_kernel(const float* __restrict__ w, float* __restrict__ u, int Cout, int Cin,
                                                          int flip) {
    const int n = Cout * Cin;
    for (int i = blockIdx.x * NT + threadIdx.x; i < n; i += gridDim.x * NT) {
        const int co = i / Cin, ci = i - co * Cin;
        float g[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                g[ky][kx] = w[((size_t)co * 9 + (flip ? (2 - ky) * 3 + (2 - kx) : ky * 3 + kx)) * Cin + ci];
        float t[4][3];     // G g
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            t[0][kx] = g[0][kx];
            t[1][kx] = 0.5f * (g[0][kx] + g[1][kx] + g[2][kx]);
            t[2][kx] = 0.5f * (g[0][kx] - g[1][kx] + g[2][kx]);
            t[3][kx] = g[2][kx];
        }
        // (G g) G^T ; K = reduction channel of the GEMM, N = output channel; component row a of (K, N) at [K][a][N] as one float4
        const int K = flip ? co : ci, N = flip ? ci : co, NN = flip ? Cin : Cout;
        f32x4* out = reinterpret_cast<f32x4*>(u) + (size_t)K * 4 * NN + N;
#pragma unroll
        for (int a = 0; a < 4; ++a)
            out[(size_t)a * NN] = f32x4{t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
    }
}

// One launch for all the weights of a network (dp.FusedAdam calls it after the Adam kernel): entry = one weight, both
// orientations; a workgroup transforms 256 (output channel, input channel) pairs of one entry.
struct WinoEntry {
    const float* w;    // [Cout][3][3][Cin]
    float* u;          // forward operand   [Cin][4][Cout][4]
    float* uf;         // data-gradient operand [Cout][4][Cin][4]
    int Cout, Cin, wg_begin, pad_;
};
__device__ __forceinline__ void wino_g(const float (&g)[3][3], f32x4 (&out)[4]) {
    float t[4][3];     // G g
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        t[0][kx] = g[0][kx];
        t[1][kx] = 0.5f * (g[0][kx] + g[1][kx] + g[2][kx]);
        t[2][kx] = 0.5f * (g[0][kx] - g[1][kx] + g[2][kx]);
        t[3][kx] = g[2][kx];
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)     // (G g) G^T
        out[a] = f32x4{t[a][0], 0.5f * (t[a][0] + t[a][1] + t[a][2]), 0.5f * (t[a][0] - t[a][1] + t[a][2]), t[a][2]};
}
__global__ __launch_bounds__(NT) void wino_weights_batch_kernel(const WinoEntry* __restrict__ tab, int n) {
    __shared__ int s_e;
    __shared__ f32x4 sU[16 * 65];                          // [ci][a][co] of a 16 x 16 block, row stride 65 (bank spread)
    if (threadIdx.x == 0) {
        int e = 0;
        while (e + 1 < n && (int)blockIdx.x >= tab[e + 1].wg_begin) ++e;
        s_e = e;
    }
    __syncthreads();
    const WinoEntry en = tab[s_e];
    const int bidx = blockIdx.x - en.wg_begin;
    // 16 x 16 blocks of (output, input) channels when both counts allow it: the forward operand [Cin][4][Cout] is contiguous along
    // the OUTPUT channel, the weight and the data-gradient operand along the input channel -- with 256 consecutive (co, ci) pairs
    // per workgroup the forward operand's 16-byte writes landed 64 Cout bytes apart (584 MB written per step for 300 MB of
    // operands); the block's forward operand now turns through LDS and leaves as 256-byte runs
    const bool blocked = (en.Cout & 15) == 0 && (en.Cin & 15) == 0;
    int co, ci;
    if (blocked) {
        const int nbc = en.Cin >> 4;
        co = (bidx / nbc) * 16 + (threadIdx.x >> 4);
        ci = (bidx % nbc) * 16 + (threadIdx.x & 15);
    } else {
        const int i = bidx * NT + threadIdx.x;
        if (i >= en.Cout * en.Cin) return;
        co = i / en.Cin;
        ci = i - co * en.Cin;
    }
    float g[3][3], gr[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            g[ky][kx] = en.w[((size_t)co * 9 + ky * 3 + kx) * en.Cin + ci];
            gr[2 - ky][2 - kx] = g[ky][kx];
        }
    f32x4 o[4];
    wino_g(gr, o);
#pragma unroll
    for (int a = 0; a < 4; ++a) reinterpret_cast<f32x4*>(en.uf)[((size_t)co * 4 + a) * en.Cin + ci] = o[a];
    wino_g(g, o);
    if (!blocked) {
#pragma unroll
        for (int a = 0; a < 4; ++a) reinterpret_cast<f32x4*>(en.u)[((size_t)ci * 4 + a) * en.Cout + co] = o[a];
        return;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) sU[(threadIdx.x & 15) * 65 + a * 16 + (threadIdx.x >> 4)] = o[a];
    __syncthreads();
    const int co2 = co - (threadIdx.x >> 4) + (threadIdx.x & 15), ci2 = ci - (threadIdx.x & 15) + (threadIdx.x >> 4);
#pragma unroll
    for (int a = 0; a < 4; ++a)
        reinterpret_cast<f32x4*>(en.u)[((size_t)ci2 * 4 + a) * en.Cout + co2] = sU[(threadIdx.x >> 4) * 65 + a * 16 + (threadIdx.x & 15)];
}

// Workgroup = 4 waves = WT tile groups x WC channel groups; a wave owns 32 tiles (linear tile index over batch, tile row, tile
// column -- no padding of the image to a block shape) x 32 output channels x all 16 components.  Per chunk of CK = 8 input channels
// the workgroup transforms its tiles ONCE (thread = one tile x VEC channels: 16 buffer loads, out-of-image pixels come back as
// zeros from the buffer bounds check) into sV[xi][tile][channel]; every wave then reads its A operands as one b128 per component
// (four k-steps per read; the tile stride of CK + 4 floats keeps the eight lanes of a read group on distinct banks).  Lane half h
// works on channels [4h, 4h + 4) of the chunk, for A and B alike.
// Software pipeline, one barrier per chunk c: | LDS read of chunk c+1's A operands | k-steps 0..3 of chunk c, with the transform +
// LDS write of chunk c+2 spread over them and the global loads of chunk c+3 issued in k-step 0 (a whole chunk ahead of their use) | barrier |.
// Three LDS buffers: the one written in chunk c was last read two barriers ago.
// GEN: 0 = zero-padded 'same' convolution of one tensor (the BasicBlock layers); 1 = general gather from one source (reflection,
// upsample, output size / origin); 2 = general gather from two concatenated sources.
template <int WT, int WC, int DBG = 0, int GEN = 0, bool RES = false>     // RES: y += res (same shape); DBG (tools/wino_bench.py): 1 no U loads, 2 no A reads, 4 no staging, 8 no barrier in the loop, 16 no epilogue
__global__ __launch_bounds__(NT, 1) void wino_fwd_kernel(WinoParams p) {
    static_assert(WT * WC == 4 && (WT == 1 || WT == 2), "four waves");
    constexpr int MT = 32 * WT, CK = 8, CKP = CK + 4, KH = 4, VEC = WT, PP = CK / VEC;
    static_assert(MT * PP == NT, "one staging item per thread");
    __shared__ __attribute__((aligned(16))) float sV[3][16][MT][CKP];
    __shared__ int sT[MT];
    using f32x2 = __attribute__((ext_vector_type(2))) float;

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave / WC, wc = wave % WC;
    const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
    const int Ho = GEN ? p.Ho : H, Wo = GEN ? p.Wo : W, org = GEN ? p.org : 1;
    const int TXn = (Wo + 1) >> 1, TYn = (Ho + 1) >> 1, ntiles = p.B * TYn * TXn;
    // XCD-aware order: workgroups that share input tiles (the channel blocks of one tile block) run on the same XCD / L2
    const int nb = p.tiles_x, ntb = p.tiles_y;                        // channel blocks, tile blocks
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int cb = slot % nb, tb = (slot / nb) * 8 + xcd;
    if (tb >= ntb) return;
    const int tile0 = tb * MT;
    const int co = cb * 32 * WC + wc * 32 + r;                        // my output channel
    const bool co_ok = co < Cout;
    const int coc = min(co, Cout - 1);

    auto tile_coords = [&](int t, int& bb, int& ty, int& tx) {
        bb = t / (TYn * TXn);
        const int rem = t - bb * (TYn * TXn);
        ty = rem / TXn;
        tx = rem - ty * TXn;
    };
    if (tid < MT) {
        const int t = tile0 + tid;
        int bb, ty, tx;
        tile_coords(min(t, ntiles - 1), bb, ty, tx);
        // first output pixel of the tile * 8 + flags (1: second row exists, 2: second column exists, 4: statistics group 1)
        sT[tid] = t < ntiles ? (((bb * Ho + 2 * ty) * Wo + 2 * tx) << 3) | (2 * ty + 1 < Ho ? 1 : 0) | (2 * tx + 1 < Wo ? 2 : 0) |
                                   (bb >= p.stat_split ? 4 : 0)
                             : -1;
    }
    // ---- my staging item: tile st_tile, channels [VEC st_c, VEC st_c + VEC) of each chunk; byte offsets of its 16 patch pixels
    const int C1 = GEN ? p.C1 : Cin, C2 = Cin - C1;
    const int Hs = (GEN && p.up) ? H >> 1 : H, Ws = (GEN && p.up) ? W >> 1 : W;          // geometry of source x
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((size_t)p.B * Hs * Ws * C1 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(GEN == 2 ? p.x2 : p.x), 0, GEN == 2 ? (int)((size_t)p.B * H * W * C2 * 4) : 0, 0x00020000);
    const int st_tile = tid / PP, st_c = tid % PP;
    unsigned st_off[4][4];       // 0xC0000000 = outside the image (or no such tile): the load returns zeros
    unsigned st_off2[GEN == 2 ? 4 : 1][GEN == 2 ? 4 : 1];
    {
        const int t = tile0 + st_tile;
        int bb, ty, tx;
        tile_coords(min(t, ntiles - 1), bb, ty, tx);
        const int iy0 = 2 * ty - org, ix0 = 2 * tx - org;
        if constexpr (GEN == 0) {
            // one multiply chain for the patch origin, the 16 pixels by adding wave-uniform strides; a pixel outside the image
            // (or a tile past the end) gets the out-of-bounds mask OR-ed in (every tensor is smaller than 2 GiB)
            const unsigned base = (unsigned)((((bb * H + iy0) * W + ix0) * Cin + VEC * st_c) * 4);
            const unsigned dead = t < ntiles ? 0u : 0xC0000000u;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned rm = (unsigned)(iy0 + i) < (unsigned)H ? dead : 0xC0000000u;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const unsigned cm = (unsigned)(ix0 + j) < (unsigned)W ? 0u : 0xC0000000u;
                    st_off[i][j] = (base + (unsigned)((i * W + j) * Cin * 4)) | rm | cm;
                }
            }
        } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                int iy = iy0 + i, ix = ix0 + j;
                bool ok = t < ntiles;
                if (GEN && p.reflect) {                   // ReflectionPad2d(1): -1 -> 1, H -> H - 2 (H, W >= 2)
                    iy = iy < 0 ? -iy : iy >= H ? 2 * H - 2 - iy : iy;
                    ix = ix < 0 ? -ix : ix >= W ? 2 * W - 2 - ix : ix;
                    ok = ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;      // (the last odd tile row / column)
                } else {
                    ok = ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                }
                const int sy = (GEN && p.up) ? iy >> 1 : iy, sx = (GEN && p.up) ? ix >> 1 : ix;
                st_off[i][j] = ok ? (unsigned)((((bb * Hs + sy) * Ws + sx) * C1 + VEC * st_c) * 4) : 0xC0000000u;
                if constexpr (GEN == 2) st_off2[i][j] = ok ? (unsigned)((((bb * H + iy) * W + ix) * C2 + VEC * st_c) * 4) : 0xC0000000u;
            }
        }
    }
    using vec_t = typename std::conditional<VEC == 2, f32x2, float>::type;
    auto load_pixel = [&](int chunk, int i, int j) -> vec_t {
        int soff = chunk * CK * 4;
        __amdgpu_buffer_rsrc_t rs = xr;
        unsigned off = st_off[i][j];
        if constexpr (GEN == 2) {                         // chunk-uniform choice of the source
            const bool second = chunk * CK >= C1;
            soff = second ? soff - C1 * 4 : soff;
            rs = second ? xr2 : xr;
            off = second ? st_off2[i][j] : off;
        }
        if constexpr (VEC == 2) return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, soff, 0));
        else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, soff, 0));
    };
    auto load_stage = [&](int chunk, vec_t (&d)[4][4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) d[i][j] = load_pixel(chunk, i, j);
    };
    auto load_stage_part = [&](int chunk, vec_t (&d)[4][4], int part) {       // part 0: pixels 0..5, 1: 6..10, 2: 11..15
        const int lo = part == 0 ? 0 : part == 1 ? 6 : 11, hi = part == 0 ? 6 : part == 1 ? 11 : 16;
#pragma unroll
        for (int e = lo; e < hi; ++e) d[e >> 2][e & 3] = load_pixel(chunk, e >> 2, e & 3);
    };
    // transform in four pieces (one per k-step): piece 0 = B^T d, piece k = row k of (B^T d) B and its four LDS writes ... rows 0..3
    // are split 1 + 1 + 1 + 1 with the column transform in front of row 0
    auto transform_piece = [&](int piece, int buf, const vec_t (&d)[4][4], vec_t (&t)[4][4]) {
        if (piece == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {                 // B^T d
                t[0][j] = d[0][j] - d[2][j];
                t[1][j] = d[1][j] + d[2][j];
                t[2][j] = d[2][j] - d[1][j];
                t[3][j] = d[1][j] - d[3][j];
            }
        }
        const int i = piece;                              // (B^T d) B, row i
        float* o = &sV[buf][4 * i][st_tile][VEC * st_c];
        *reinterpret_cast<vec_t*>(o) = t[i][0] - t[i][2];
        *reinterpret_cast<vec_t*>(o + MT * CKP) = t[i][1] + t[i][2];
        *reinterpret_cast<vec_t*>(o + 2 * MT * CKP) = t[i][2] - t[i][1];
        *reinterpret_cast<vec_t*>(o + 3 * MT * CKP) = t[i][1] - t[i][3];
    };

    // ---- B operands: u [K][4][N] of float4; my channel of k-step j of chunk c is 8 c + 4 h + j
    const __amdgpu_buffer_rsrc_t ur =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.u), 0, (int)((size_t)Cin * Cout * 64), 0x00020000);
    const unsigned u_voff = (unsigned)(((size_t)h * KH * 4 * Cout + coc) * 16);
    const int u_kstride = 4 * Cout * 16, u_qstride = Cout * 16;       // bytes per channel, per component row
    f32x4 un[4];
    auto load_u = [&](int chunk, int j) {
        const int soff = (chunk * CK + j) * u_kstride;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // (bit_cast of the builtin's own vector type: an implicit conversion to an ext_vector_type splats element 0)
            un[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, u_voff, soff + q * u_qstride, 0));
        }
    };
    load_u(0, 0);

    const int nchunk = Cin / CK;
    vec_t stg[4][4], tt[4][4];
    f32x4 a[16], an[16];
    auto read_a = [&](int buf, f32x4 (&dst)[16]) {
#pragma unroll
        for (int q = 0; q < 16; ++q) dst[q] = *reinterpret_cast<const f32x4*>(&sV[buf][q][32 * wt + r][h * KH]);
    };
    // prologue: chunks 0 and 1 into buffers 0 and 1 (both loads in flight together), chunk 2 loaded (chunks past the end
    // re-read the last one; never used)
    {
        vec_t stg1[4][4];
        load_stage(0, stg);
        load_stage(1, stg1);                              // nchunk >= 2
#pragma unroll
        for (int k = 0; k < 4; ++k) transform_piece(k, 0, stg, tt);
        load_stage(min(2, nchunk - 1), stg);
#pragma unroll
        for (int k = 0; k < 4; ++k) transform_piece(k, 1, stg1, tt);
    }
    __syncthreads();
    read_a(0, a);
    f32x16 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;

    // one chunk: MFMAs of chunk ch from `ac`; `anx` <- A operands of chunk ch + 1.  Every k-step is one scheduling region in which
    // the next k-step's B operands, a quarter of the A reads, a transform piece and a third of the stage loads are interleaved
    // with the 16 MFMAs (one memory instruction behind each of the first MFMAs, B operands first: they are needed soonest).
    int bnext = 1, bwrite = 2;                             // buffers of chunk ch + 1 (to read) and ch + 2 (to write)
    auto chunk_body = [&](int ch, f32x4 (&ac)[16], f32x4 (&anx)[16]) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            __builtin_amdgcn_sched_barrier(0);
            float uc[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) uc[q] = un[q >> 2][q & 3];
            if (!(DBG & 1)) {
                if (j < 3) load_u(ch, j + 1);
                else load_u(min(ch + 1, nchunk - 1), 0);   // after the last chunk: a re-read, never used
            }
            if (!(DBG & 2)) {
#pragma unroll
                for (int q = 4 * j; q < 4 * j + 4; ++q)
                    anx[q] = *reinterpret_cast<const f32x4*>(&sV[bnext][q][32 * wt + r][h * KH]);
            }
            if (!(DBG & 4)) {
                transform_piece(j, bwrite, stg, tt);
                // stg is free once piece 0 has formed B^T d: the loads of chunk ch + 3 follow in k-steps 0..2 (6 + 5 + 5)
                if (j < 3) load_stage_part(min(ch + 3, nchunk - 1), stg, j);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[q][j], uc[q], acc[q], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // DS read
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // VALU
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);     // DS write
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(DBG & 8)) __syncthreads();
        const int bfree = bnext == 0 ? 2 : bnext - 1;      // (ch mod 3): read during the previous chunk
        bnext = bwrite;
        bwrite = bfree;
    };
    int ch = 0;
    do {                                                   // Cin % 16 == 0: an even, non-zero number of chunks (no loop guard:
        chunk_body(ch, a, an);                             // with one the compiler zeroes the 256 accumulators on both paths)
        chunk_body(ch + 1, an, a);
        ch += 2;
    } while (ch < nchunk);

    // ---- output transform, bias / ReLU / statistics, store: reg i <-> tile m = (i & 3) + 8 (i >> 2) + 4 h of my channel
    if (DBG & 16) {
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) sum += acc[q][i];
        if (co_ok && sum == 123.f) p.y[co] = sum;
        return;
    }
    // Branch-free: an output that does not exist (tile past the end, odd H / W, channel past Cout) gets a buffer offset beyond
    // the tensor and the store is dropped by the bounds check; tiles are processed in register pairs (packed fp32 adds).
    const float bv = (p.bias && co_ok) ? p.bias[co] : 0.f;
    const float lo = p.relu ? 0.f : -__builtin_inff();
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.B * Ho * Wo * Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(RES ? p.res : p.y), 0,
                                                                         RES ? (int)((size_t)p.B * Ho * Wo * Cout * 4) : 0, 0x00020000);
    const bool want_stats = p.stats != nullptr;
    int tinfo[16];
    unsigned tbase[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) tinfo[i] = sT[32 * wt + (i & 3) + 8 * (i >> 2) + 4 * h];
#pragma unroll
    for (int i = 0; i < 16; ++i) tbase[i] = ((unsigned)(tinfo[i] >> 3) * Cout + co) * 4u;
    float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};
    const unsigned row_b = (unsigned)Wo * Cout * 4, col_b = (unsigned)Cout * 4;
    float rv[RES ? 16 : 1][4];                            // residual values: all 64 loads in flight before the transform arithmetic
    if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned ob = ((tinfo[i] >= 0) & co_ok) ? tbase[i] : 0xC0000000u;      // (rows / columns past an odd edge: masked at the store)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                rv[i][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, ob, (k >> 1) * row_b + (k & 1) * col_b, 0));
        }
    }
#pragma unroll
    for (int ip = 0; ip < 8; ++ip) {
        const int i = 2 * ip;
        f32x2 s[2][4], y[2][2];
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {                  // A^T M
            const f32x2 m0{acc[0 + bq][i], acc[0 + bq][i + 1]}, m1{acc[4 + bq][i], acc[4 + bq][i + 1]};
            const f32x2 m2{acc[8 + bq][i], acc[8 + bq][i + 1]}, m3{acc[12 + bq][i], acc[12 + bq][i + 1]};
            s[0][bq] = m0 + m1 + m2;
            s[1][bq] = m1 - m2 - m3;
        }
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {                  // (A^T M) A
            y[a2][0] = s[a2][0] + s[a2][1] + s[a2][2];
            y[a2][1] = s[a2][1] - s[a2][2] - s[a2][3];
        }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int info = tinfo[i + e];
            const bool ok = (info >= 0) & co_ok;
            const unsigned base = tbase[i + e];
            const bool okr = ok & ((info & 1) != 0), okc = ok & ((info & 2) != 0), okrc = okr & okc;
            const bool v[2][2] = {{ok, okc}, {okr, okrc}};
#pragma unroll
            for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                for (int c2 = 0; c2 < 2; ++c2) {
                    float yv = y[a2][c2][e];
                    const unsigned off = v[a2][c2] ? base + a2 * row_b + c2 * col_b : 0xC0000000u;
                    if constexpr (RES) yv += rv[i + e][2 * a2 + c2];
                    float z = fmaxf(yv + bv, lo);
                    if constexpr (GEN != 0) {
                        if (p.act == kActElu) z = yv + bv > 0.f ? yv + bv : expm1f(yv + bv);
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, z), yr, off, 0, 0);
                }
            if (want_stats) {
                float ls = 0.f, lq = 0.f;
#pragma unroll
                for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
                    for (int c2 = 0; c2 < 2; ++c2) {
                        const float yv = v[a2][c2] ? y[a2][c2][e] : 0.f;
                        ls += yv;
                        lq += yv * yv;
                    }
                const bool g1 = info & 4;
                ssum[0] += g1 ? 0.f : ls;
                ssq[0] += g1 ? 0.f : lq;
                ssum[1] += g1 ? ls : 0.f;
                ssq[1] += g1 ? lq : 0.f;
            }
        }
    }
    if (p.stats) {
        // One workgroup per CU: a workgroup's last act is these atomics, and its CU stays occupied until they are acknowledged.  Thousands
        // of workgroups adding to the SAME 2 x Cout addresses serialise in the L2 (~35 ns each: +120 us on a 140 us layer-1 launch at
        // batch 24), so the workgroups spread over several copies of the table that the BatchNorm kernel adds up (dvs_bn_fwd_slots).
        float* st = p.stats + (size_t)((int)blockIdx.x & p.stat_mask) * p.stat_stride;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float a2 = ssum[g] + __shfl_xor(ssum[g], 32, 64), q2 = ssq[g] + __shfl_xor(ssq[g], 32, 64);
            if (h == 0 && co_ok && (a2 != 0.f || q2 != 0.f)) {
                atomicAdd(st + g * 2 * Cout + co, a2);
                atomicAdd(st + g * 2 * Cout + Cout + co, q2);
            }
        }
    }
}

// ---- weight gradient:  dL/dg = G^T [ sum over tiles (A dY A^T) o (B^T d B) ] G  (the same bilinear form read for g) ------------
// 16 GEMMs dU_xi [Cout x Cin] = sum_tiles P_xi[tile][co] * V_xi[tile][ci] with K = tiles.  A wave owns a 32 x 32 block of
// (output channel, input channel) pairs for all 16 components (256 accumulator registers) and a range of tiles; both operands are
// transformed in-lane from global memory (lane = one channel of one tile: 4 dY values -> 16, 4 x 4 input values -> 16; all loads
// are dword buffer loads that are contiguous over the 32 channels of a half wave; out-of-image pixels come back as zeros from the
// buffer bounds check).  No LDS and no barrier in the loop.  Three-stage software pipeline per k-step k: the loads of k+2 are
// issued, the operands of k+1 are transformed and the offsets of k+3 computed between the MFMAs of k -- none of that vector work
// depends on the MFMAs it is interleaved with.  The four waves of a workgroup split the workgroup's tile range and add their
// accumulators through LDS; the first wave applies G^T . G in-lane (all 16 components of a (co, ci) pair sit in one lane) and
// adds the 3 x 3 results into dw -- plain adds when the tile range is not split over workgroups, float atomics otherwise.
struct WinoWgradParams {
    const float* x;        // [B][H][W][Cin]
    const float* dy;       // [B][H][W][Cout]
    float* dw;             // [Cout][3][3][Cin], +=
    int B, H, W, Cin, Cout;
    int nblk_ci, nblk;     // 32-channel blocks of Cin; blocks of (Cout, Cin)
    int S;                 // splits of the tile range over workgroups
    int tiles_per_wg;      // multiple of 16
    int CinW;              // row stride of dw in channels (= Cin, or the concatenated channel count when x is one of two sources)
    const float* yact;     // DACT: the forward output y [B][H][W][Cout]; dy is multiplied by act'(y) as it is loaded
    int dact;              // activation code of conv_common.h (1 ReLU, 2 ELU)
    float* dbias;          // DACT: null, or [Cout] += the column sums of dy act'(y) (taken by the workgroups of input-channel block 0)
};

// MODE 0: zero padding 1 (the BasicBlock layers).  The decoder's gathers (model/layers.py:26-41, model/depth_decoder.py:52-62):
// MODE 1: ReflectionPad2d(1) -- a patch pixel outside the image moves two rows / columns back inside instead of reading zero (the
//         same two vector operations per offset: an add instead of an or);  MODE 2: x is the half-resolution operand of the nearest
//         2x upsample, [B][H/2][W/2][Cin] -- the 4 x 4 patch of output tile (ty, tx) is source pixels {ty-1, ty, ty, ty+1} x
//         {tx-1, tx, tx, tx+1}, clamped at the border (= reflection of the upsampled image): 9 loads instead of 16.
// The two sources of an upsample + concat layer are two launches, each adding into its own channel range of dw (CinW).
// DACT (the thin decoder layers, whose gradient kernels apply the activation derivative themselves): dZ = dY act'(Y) is formed from
// four more loads per tile and the bias gradient rides along.
template <int MODE, bool DACT = false>
__global__ __launch_bounds__(NT, 1) void wino_wgrad_kernel(WinoWgradParams p) {
    __shared__ float sR[2][256][64];                      // 128 KB: accumulators of two waves during the reduction
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
    const int TXn = (W + 1) >> 1, TYn = (H + 1) >> 1, ntiles = p.B * TYn * TXn;
    // workgroups of one XCD walk the channel blocks of the same tile range first (its x / dY stay in that L2); with fewer than
    // eight tile ranges (S = 1, 2 or 4) 8 / S XCDs share a range and deal its channel blocks between them
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int blk, sp;
    if (p.S >= 8) {
        blk = slot % p.nblk;
        sp = (slot / p.nblk) * 8 + xcd;
    } else {
        sp = xcd % p.S;
        blk = slot * (8 / p.S) + xcd / p.S;
    }
    if (sp >= p.S || blk >= p.nblk) return;
    const int co0 = (blk / p.nblk_ci) * 32, ci0 = (blk % p.nblk_ci) * 32;
    const int per_wave = p.tiles_per_wg >> 2;              // multiple of 4: an even number of k-steps
    const int t_begin = sp * p.tiles_per_wg + wave * per_wave;
    const int t_end = min(t_begin + per_wave, ntiles);
    const int ksteps = t_begin < ntiles ? per_wave >> 1 : 0;

    // the x resource starts (W + 1) pixels before the tensor so that patch offsets are non-negative: pixel (i, j) of the patch of
    // a tile whose first output pixel has index pb is at pb + (i - 1) W + (j - 1); reads in front of the tensor are masked out
    constexpr int NX = MODE == 2 ? 9 : 16;                 // loads of one patch
    constexpr int NY = DACT ? 8 : 4;                       // dY (and Y) values of one tile
    const int Hs = MODE == 2 ? H >> 1 : H, Ws = MODE == 2 ? W >> 1 : W;      // geometry of x
    // MODE 1 / 2 move a mirrored / clamped pixel by up to two rows and two columns (one and one) towards the front: the resource starts
    // that much earlier still and every offset carries the shift, so that voffset alone never goes below zero (the bounds check does
    // not wrap)
    const unsigned kshift = MODE == 1 ? 2u * (unsigned)((W + 1) * Cin * 4) : MODE == 2 ? (unsigned)((Ws + 1) * Cin * 4) : 0u;
    const size_t lead = (size_t)(Ws + 1) * Cin * 4 + kshift;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x)) - lead, 0, (int)((size_t)p.B * Hs * Ws * Cin * 4 + lead), 0x00020000);
    const __amdgpu_buffer_rsrc_t yr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)((size_t)p.B * H * W * Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DACT ? p.yact : p.dy), 0,
                                                                        (int)((size_t)p.B * H * W * Cout * 4), 0x00020000);
    constexpr unsigned OOB = 0xC0000000u;
    float bsum = 0.f;                                      // DACT: my channel's sum of dZ over my tiles

    // my tile of k-step k: t_begin + 2 k + h
    int t = t_begin + h, tb, ty, tx;
    {
        const int tc = min(t, ntiles - 1);
        tb = tc / (TYn * TXn);
        const int rem = tc - tb * (TYn * TXn);
        ty = rem / TXn;
        tx = rem - ty * TXn;
    }
    // offsets of the current tile's 16 (9) + 4 loads: v | row mask | column mask, a mask being 0 (inside) or OOB (outside: the OR
    // lands beyond every tensor < 2 GiB); then two tiles on
    unsigned ox_[NX], oy_[4];
    auto offsets_and_advance = [&]() {
        const int oy = 2 * ty, ox = 2 * tx;
        const unsigned pb = (unsigned)((tb * H + oy) * W + ox);
        const unsigned vy = (pb * Cout + co0 + r) * 4u;
        const unsigned dead = t < t_end ? 0u : OOB;
        if constexpr (MODE == 0) {
            const unsigned vx = (pb * Cin + ci0 + r) * 4u;
            const unsigned rm[4] = {oy >= 1 ? dead : OOB, dead, oy + 1 < H ? dead : OOB, oy + 2 < H ? dead : OOB};
            const unsigned cm[4] = {ox >= 1 ? 0u : OOB, 0u, ox + 1 < W ? 0u : OOB, ox + 2 < W ? 0u : OOB};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) ox_[4 * i + j] = vx | rm[i] | cm[j];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) oy_[2 * i + j] = vy | rm[1 + i] | cm[1 + j];
        } else if constexpr (MODE == 1) {
            // rows oy - 1 .. oy + 2: -1 -> 1, H -> H - 2 (with H odd the last tile row has oy + 1 == H: its first output row reads
            // that mirrored row); row H + 1 only meets the masked second dY row of such a tile: it reads zero.  Columns alike.
            const unsigned vx = (pb * Cin + ci0 + r) * 4u + kshift;
            const unsigned rowb = (unsigned)(W * Cin * 4), colb = (unsigned)(Cin * 4);
            const unsigned in_r = oy + 1 < H ? dead : OOB, in_c = ox + 1 < W ? 0u : OOB;
            const unsigned rm[4] = {dead, dead, dead, in_r};
            const unsigned cm[4] = {0u, 0u, 0u, in_c};
            const unsigned vr[4] = {vx + (oy >= 1 ? 0u : 2u * rowb), vx, vx - (oy + 1 == H ? 2u * rowb : 0u), vx - (oy + 2 == H ? 2u * rowb : 0u)};
            const unsigned ca[4] = {ox >= 1 ? 0u : 2u * colb, 0u, ox + 1 == W ? 0u - 2u * colb : 0u, ox + 2 == W ? 0u - 2u * colb : 0u};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) ox_[4 * i + j] = (vr[i] + ca[j]) | rm[i] | cm[j];
            oy_[0] = vy | dead;
            oy_[1] = vy | dead | in_c;
            oy_[2] = vy | in_r;
            oy_[3] = vy | in_r | in_c;
        } else {
            // source pixel (ty, tx) and its eight neighbours, clamped (H and W are even: every tile is whole)
            const unsigned vs = (unsigned)((((tb * Hs + ty) * Ws + tx) * Cin + ci0 + r) * 4) + kshift;
            const unsigned rowb = (unsigned)(Ws * Cin * 4), colb = (unsigned)(Cin * 4);
            const unsigned vr[3] = {vs + (ty >= 1 ? 0u : rowb), vs, vs - (ty + 1 < Hs ? 0u : rowb)};
            const unsigned ca[3] = {tx >= 1 ? 0u : colb, 0u, tx + 1 < Ws ? 0u : 0u - colb};
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j) ox_[3 * i + j] = (vr[i] + ca[j]) | dead;
#pragma unroll
            for (int e = 0; e < 4; ++e) oy_[e] = vy | dead;
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            ++tx;
            const bool wx = tx == TXn;
            tx = wx ? 0 : tx;
            ty += wx ? 1 : 0;
            const bool wy = ty == TYn;
            ty = wy ? 0 : ty;
            tb += wy ? 1 : 0;
        }
        t += 2;
    };
    // one batch = the 4 (8) dY (and Y) values of my tile, then its 16 (9) patch pixels; the staging moves (stage_loaded) read them in
    // the same order, so the waits in front of those moves count down through the batch
    auto issue_loads = [&](float (&xd)[NX], float (&yd)[NY]) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            yd[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(yr, oy_[e], ((e >> 1) * W + (e & 1)) * Cout * 4, 0));
        if constexpr (DACT) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                yd[4 + e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ar, oy_[e], ((e >> 1) * W + (e & 1)) * Cout * 4, 0));
        }
#pragma unroll
        for (int e = 0; e < NX; ++e) {
            const int soff = MODE == 2 ? ((e / 3) * Ws + (e % 3)) * Cin * 4 : ((e >> 2) * W + (e & 3)) * Cin * 4;
            xd[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xr, ox_[e], soff, 0));
        }
    };
    auto transform = [&](const float (&xs)[NX], const float (&yl)[NY], float (&v)[16], float (&pm)[16]) {
        float tt[4][4], pr[4][2], xc[16], yc[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            yc[e] = yl[e];
            if constexpr (DACT) {                         // ELU: 1 + min(y, 0); ReLU: [y > 0]  (a masked pixel has dY = Y = 0)
                const float ya = yl[NY - 4 + e];
                yc[e] = p.dact == kActElu ? fmaf(yc[e], fminf(ya, 0.f), yc[e]) : (ya > 0.f ? yc[e] : 0.f);
                bsum += yc[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {                    // MODE 2: patch row i = source row {0, 1, 1, 2}[i], columns alike
            constexpr int dup[4] = {0, 1, 1, 2};
            xc[e] = MODE == 2 ? xs[3 * dup[e >> 2] + dup[e & 3]] : xs[e];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {                     // B^T d
            tt[0][j] = xc[j] - xc[8 + j];
            tt[1][j] = xc[4 + j] + xc[8 + j];
            tt[2][j] = xc[8 + j] - xc[4 + j];
            tt[3][j] = xc[4 + j] - xc[12 + j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                     // (B^T d) B
            v[4 * i + 0] = tt[i][0] - tt[i][2];
            v[4 * i + 1] = tt[i][1] + tt[i][2];
            v[4 * i + 2] = tt[i][2] - tt[i][1];
            v[4 * i + 3] = tt[i][1] - tt[i][3];
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {                     // A dY: rows (1,0), (1,1), (1,-1), (0,-1)
            pr[0][j] = yc[j];
            pr[1][j] = yc[j] + yc[2 + j];
            pr[2][j] = yc[j] - yc[2 + j];
            pr[3][j] = -yc[2 + j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                     // (A dY) A^T
            pm[4 * i + 0] = pr[i][0];
            pm[4 * i + 1] = pr[i][0] + pr[i][1];
            pm[4 * i + 2] = pr[i][0] - pr[i][1];
            pm[4 * i + 3] = -pr[i][1];
        }
    };

    f32x16 acc[16];
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[q][i] = 0.f;

    // step k: the values loaded during step k-1 (k-step k+1's) move from the in-flight set rl to the staging set sn -- real moves,
    // placed where the transform needs the data anyway; the loads of k+2 -> rl (its offsets were computed in step k-1); MFMAs from
    // (vc, pc); (vn, pn) <- transform of sn.  With two alternating load sets the register allocator carried a few loaded values
    // around the loop in other registers and put the copies -- hence an s_waitcnt for the batch issued in that very k-step -- at the
    // back edge: every second k-step ran without any prefetch distance.
    // Two in-flight sets (ra, rb) alternate, so a batch has TWO k-steps (~0.9 us) to arrive: it is issued in step k and staged in
    // step k + 2.
    float rax[NX], ray[NY], rbx[NX], rby[NY], snx[NX], sny[NY], v0[16], p0[16], v1[16], p1[16];
    auto stage_loaded = [&](const float (&lx)[NX], const float (&ly)[NY]) {
#pragma unroll
        for (int e = 0; e < NY; ++e) asm volatile("v_mov_b32 %0, %1" : "=v"(sny[e]) : "v"(ly[e]));
#pragma unroll
        for (int e = 0; e < NX; ++e) asm volatile("v_mov_b32 %0, %1" : "=v"(snx[e]) : "v"(lx[e]));
    };
    // step k: k-step k+1's values (loaded during step k-2) -> sn; loads of k+3 -> the same set; (vn, pn) <- transform of sn; offsets
    // of k+4; MFMAs from (vc, pc)
    auto kstep = [&](float (&lx)[NX], float (&ly)[NY], const float (&vc)[16], const float (&pc)[16], float (&vn)[16], float (&pn)[16]) {
        __builtin_amdgcn_sched_barrier(0);
        stage_loaded(lx, ly);
        __builtin_amdgcn_sched_barrier(0);
        issue_loads(lx, ly);
        __builtin_amdgcn_sched_barrier(0);
        transform(snx, sny, vn, pn);
        offsets_and_advance();
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(pc[q], vc[q], acc[q], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);     // VALU
        }
    };
    offsets_and_advance();
    issue_loads(rax, ray);                                // k-step 0
    offsets_and_advance();
    issue_loads(rbx, rby);                                // k-step 1
    offsets_and_advance();
    stage_loaded(rax, ray);
    issue_loads(rax, ray);                                // k-step 2
    offsets_and_advance();                                // offsets of k-step 3
    transform(snx, sny, v0, p0);
    for (int k = 0; k < ksteps; k += 2) {
        kstep(rbx, rby, v0, p0, v1, p1);
        kstep(rax, ray, v1, p1, v0, p0);
    }
    __builtin_amdgcn_sched_barrier(0);

    if constexpr (DACT) {
        // every tile's dZ went through transform() exactly once per (output-channel, input-channel) block: block column 0 reports
        if (p.dbias && ci0 == 0) {
            const float b2 = bsum + __shfl_xor(bsum, 32, 64);
            if (h == 0) atomicAdd(p.dbias + co0 + r, b2);
        }
    }
    // ---- add the four waves' accumulators: waves 2, 3 -> LDS, waves 0, 1 add; wave 1 -> LDS, wave 0 adds
    if (wave >= 2) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) sR[wave - 2][q * 16 + i][lane] = acc[q][i];
    }
    __syncthreads();
    if (wave < 2) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[q][i] += sR[wave][q * 16 + i][lane];
    }
    __syncthreads();
    if (wave == 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) sR[0][q * 16 + i][lane] = acc[q][i];
    }
    __syncthreads();
    if (wave == 0) {
        const bool exclusive = p.S == 1;                  // this workgroup alone owns its block of dw
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int m = (i & 3) + 8 * (i >> 2) + 4 * h;  // output channel row of the block
            float u[4][4], t3[3][4];
#pragma unroll
            for (int q = 0; q < 16; ++q) u[q >> 2][q & 3] = acc[q][i] + sR[0][q * 16 + i][lane];
#pragma unroll
            for (int b = 0; b < 4; ++b) {                  // G^T dU
                const float hs = 0.5f * (u[1][b] + u[2][b]);
                t3[0][b] = u[0][b] + hs;
                t3[1][b] = 0.5f * (u[1][b] - u[2][b]);
                t3[2][b] = hs + u[3][b];
            }
            float* o = p.dw + (size_t)(co0 + m) * 9 * p.CinW + ci0 + r;
#pragma unroll
            for (int k = 0; k < 3; ++k) {                  // (G^T dU) G
                const float hs = 0.5f * (t3[k][1] + t3[k][2]);
                const float w3[3] = {t3[k][0] + hs, 0.5f * (t3[k][1] - t3[k][2]), hs + t3[k][3]};
#pragma unroll
                for (int l = 0; l < 3; ++l) {
                    float* a = o + (size_t)(3 * k + l) * p.CinW;
                    if (exclusive) *a += w3[l];
                    else atomicAdd(a, w3[l]);
                }
            }
        }
    }
}

template <int WT, int WC, int GEN = 0>
void launch_wino(WinoParams& p, hipStream_t st) {
    const int Ho = GEN ? p.Ho : p.H, Wo = GEN ? p.Wo : p.W;
    const int ntiles = p.B * ((Ho + 1) / 2) * ((Wo + 1) / 2);
    p.tiles_x = (p.Cout + 32 * WC - 1) / (32 * WC);       // channel blocks
    p.tiles_y = (ntiles + 32 * WT - 1) / (32 * WT);       // tile blocks
    const size_t grid = (size_t)((p.tiles_y + 7) / 8) * 8 * p.tiles_x;
    if constexpr (GEN == 0) {
        static const int dbg = getenv("DVS_WINO_DBG") ? atoi(getenv("DVS_WINO_DBG")) : 0;
#define WINO_DBG_CASE(D) \
    case D: hipLaunchKernelGGL((wino_fwd_kernel<WT, WC, D>), dim3((unsigned)grid), dim3(NT), 0, st, p); break;
        if (p.res) {
            hipLaunchKernelGGL((wino_fwd_kernel<WT, WC, 0, 0, true>), dim3((unsigned)grid), dim3(NT), 0, st, p);
            return;
        }
        switch (dbg) {
            WINO_DBG_CASE(1) WINO_DBG_CASE(2) WINO_DBG_CASE(4) WINO_DBG_CASE(8) WINO_DBG_CASE(15) WINO_DBG_CASE(16) WINO_DBG_CASE(31)
            default: hipLaunchKernelGGL((wino_fwd_kernel<WT, WC>), dim3((unsigned)grid), dim3(NT), 0, st, p);
        }
#undef WINO_DBG_CASE
    } else {
        hipLaunchKernelGGL((wino_fwd_kernel<WT, WC, 0, GEN>), dim3((unsigned)grid), dim3(NT), 0, st, p);
    }
}

// x: the source tensor of this launch (MODE 2: at half resolution); dw: already offset to the source's first channel, CinW its row stride
template <int MODE, bool DACT = false>
void launch_wino_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int CinW, int target_workgroups,
                       hipStream_t st, const float* yact = nullptr, int dact = 0, float* dbias = nullptr) {
    WinoWgradParams p{x, dy, dw, B, H, W, Cin, Cout, Cin / 32, (Cin / 32) * (Cout / 32), 0, 0, CinW, yact, dact, dbias};
    const int ntiles = B * ((H + 1) / 2) * ((W + 1) / 2);
    // default: one round of one workgroup per CU (fewer, longer tile ranges: less reduction and atomic traffic); the 512-channel
    // layers take two tile ranges so that one XCD's share of x and dY fits its L2 (measured: profiles/r02_f_wino_wgrad_split.txt)
    if (target_workgroups <= 0) target_workgroups = p.nblk >= 256 ? 512 : 256;
    int S = (target_workgroups + p.nblk / 2) / p.nblk;
    S = S < 1 ? 1 : S;
    int tpw = ((ntiles + S - 1) / S + 15) & ~15;          // tiles per workgroup: four waves x two tiles x an even number of k-steps
    tpw = tpw < 64 ? 64 : tpw;                             // at least eight k-steps per wave
    S = (ntiles + tpw - 1) / tpw;
    if (S < 8) {                                           // 1, 2 or 4 ranges (the XCD map of the kernel); ranges past the end add zeros
        S = S >= 4 ? 4 : S >= 2 ? 2 : 1;
        tpw = ((ntiles + S - 1) / S + 15) & ~15;
    }
    p.tiles_per_wg = tpw;
    p.S = S;
    const size_t grid = S >= 8 ? (size_t)((S + 7) / 8) * 8 * p.nblk : (size_t)((p.nblk + 8 / S - 1) / (8 / S)) * 8;
    hipLaunchKernelGGL((wino_wgrad_kernel<MODE, DACT>), dim3((unsigned)grid), dim3(NT), 0, st, p);
}

}  // namespace

extern "C" {

int dvs_wino_weights(const float* w, float* u, int Cout, int Cin, int flip, void* stream) {
    DVS_REQUIRE(w && u && Cout > 0 && Cin > 0, "dvs_wino_weights: bad argument");
    int blocks = (Cout * Cin + NT - 1) / NT;
    blocks = blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(wino_weights_kernel, dim3(blocks), dim3(NT), 0, static_cast<hipStream_t>(stream), w, u, Cout, Cin, flip);
    return dvs::check_launch("dvs_wino_weights");
}

int dvs_wino_weights_batch(const void* table, int n_entries, int total_workgroups, void* stream) {
    DVS_REQUIRE(table && n_entries > 0 && total_workgroups > 0, "dvs_wino_weights_batch: bad argument");
    hipLaunchKernelGGL(wino_weights_batch_kernel, dim3(total_workgroups), dim3(NT), 0, static_cast<hipStream_t>(stream),
                       static_cast<const WinoEntry*>(table), n_entries);
    return dvs::check_launch("dvs_wino_weights_batch");
}

int dvs_conv3x3_wino_fwd(const float* x, const float* u, const float* bias, const float* res, float* y, float* stats, int stat_groups,
                         int B, int H, int W, int Cin, int Cout, int relu, int as_dgrad, void* stream) {
    return dvs_conv3x3_wino_fwd_slots(x, u, bias, res, y, stats, stat_groups, 1, B, H, W, Cin, Cout, relu, as_dgrad, stream);
}

int dvs_conv3x3_wino_fwd_slots(const float* x, const float* u, const float* bias, const float* res, float* y, float* stats,
                               int stat_groups, int stat_slots, int B, int H, int W, int Cin, int Cout, int relu, int as_dgrad,
                               void* stream) {
    DVS_REQUIRE(x && u && y && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "dvs_conv3x3_wino_fwd: bad argument");
    DVS_REQUIRE(stat_slots >= 1 && stat_slots <= 64 && (stat_slots & (stat_slots - 1)) == 0,
                "dvs_conv3x3_wino_fwd: stat_slots must be a power of two in [1, 64] (got %d)", stat_slots);
    DVS_REQUIRE(Cin % CIN_MULT == 0 && (Cout & 3) == 0, "dvs_conv3x3_wino_fwd: Cin %% 16 == 0 and Cout %% 4 == 0 (got %d, %d)", Cin, Cout);
    DVS_REQUIRE(stat_groups >= 0 && stat_groups <= 2 && (stat_groups != 2 || (B & 1) == 0), "dvs_conv3x3_wino_fwd: stat_groups");
    DVS_REQUIRE((double)B * H * W * (Cin > Cout ? Cin : Cout) * 4 < 2147483648.0 && (double)Cin * Cout * 64 < 2147483648.0,
                "dvs_conv3x3_wino_fwd: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    WinoParams p{x, u, bias, y, stats, B, H, W, Cin, Cout, 0, 0, relu, stat_groups == 2 ? B / 2 : 0x7fffffff,
                 nullptr, Cin, 0, 0, H, W, 1, relu ? kActRelu : 0, res, stat_slots - 1,
                 stat_slots > 1 ? (stat_groups == 2 ? 2 : 1) * 2 * Cout : 0};
    hipStream_t st = static_cast<hipStream_t>(stream);
    dvs::ProfScope prof(as_dgrad ? dvs::SLOT_CONV_DGRAD : dvs::SLOT_CONV_FWD, st);      // flops of the direct convolution
    prof.work(2.0 * B * H * W * Cout * (double)Cin * 9);
    if (Cout > 64) launch_wino<1, 4>(p, st);
    else launch_wino<2, 2>(p, st);
    return dvs::check_launch("dvs_conv3x3_wino_fwd");
}

int dvs_conv3x3_wino_gen(const float* x, const float* x2, const float* u, const float* bias, float* y, int B, int H, int W, int C1, int C2,
                         int Cout, int Ho, int Wo, int org, int upsample, int reflect, int act, int as_dgrad, void* stream) {
    DVS_REQUIRE(x && u && y && B > 0 && H > 0 && W > 0 && C1 > 0 && C2 >= 0 && Cout > 0, "dvs_conv3x3_wino_gen: bad argument");
    DVS_REQUIRE((C2 == 0) == (x2 == nullptr), "dvs_conv3x3_wino_gen: x2 and C2 go together");
    DVS_REQUIRE(C1 % 8 == 0 && (C1 + C2) % CIN_MULT == 0 && (Cout & 3) == 0,
                "dvs_conv3x3_wino_gen: C1 %% 8 == 0, (C1 + C2) %% 16 == 0, Cout %% 4 == 0 (got %d, %d, %d)", C1, C2, Cout);
    DVS_REQUIRE((org == 1 && Ho == H && Wo == W) || (org == 2 && Ho == H + 2 && Wo == W + 2 && !reflect),
                "dvs_conv3x3_wino_gen: origin 1 = 'same' output, origin 2 = full correlation (H + 2, W + 2, zero padding)");
    DVS_REQUIRE(!reflect || (H >= 2 && W >= 2), "dvs_conv3x3_wino_gen: ReflectionPad2d(1) needs H, W >= 2");
    DVS_REQUIRE(!upsample || ((H & 1) == 0 && (W & 1) == 0), "dvs_conv3x3_wino_gen: upsampled input has even H, W");
    DVS_REQUIRE(act == 0 || act == kActRelu || act == kActElu, "dvs_conv3x3_wino_gen: activation %d (0, 1 = ReLU, 2 = ELU)", act);
    const int Cin = C1 + C2;
    DVS_REQUIRE((double)B * Ho * Wo * (Cin > Cout ? Cin : Cout) * 4 < 2147483648.0 && (double)Cin * Cout * 64 < 2147483648.0,
                "dvs_conv3x3_wino_gen: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    WinoParams p{x, u, bias, y, nullptr, B, H, W, Cin, Cout, 0, 0, act == kActRelu, 0x7fffffff, x2, C1, upsample, reflect, Ho, Wo, org, act, nullptr, 0, 0};
    hipStream_t st = static_cast<hipStream_t>(stream);
    dvs::ProfScope prof(as_dgrad ? dvs::SLOT_CONV_DGRAD : dvs::SLOT_CONV_FWD, st);      // flops of the direct convolution
    prof.work(2.0 * B * (as_dgrad ? H * W : Ho * Wo) * Cout * (double)Cin * 9);
    if (x2) {
        if (Cout > 64) launch_wino<1, 4, 2>(p, st);
        else launch_wino<2, 2, 2>(p, st);
    } else {
        if (Cout > 64) launch_wino<1, 4, 1>(p, st);
        else launch_wino<2, 2, 1>(p, st);
    }
    return dvs::check_launch("dvs_conv3x3_wino_gen");
}

int dvs_conv3x3_wino_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int target_workgroups,
                           void* stream) {
    DVS_REQUIRE(x && dy && dw && B > 0 && H > 0 && W > 0, "dvs_conv3x3_wino_wgrad: bad argument");
    DVS_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0 && Cin > 0 && Cout > 0, "dvs_conv3x3_wino_wgrad: channel counts must be multiples of 32 (got %d, %d)",
                Cin, Cout);
    DVS_REQUIRE(((double)B * H * W + W + 1) * (Cin > Cout ? Cin : Cout) * 4 < 2147483648.0,
                "dvs_conv3x3_wino_wgrad: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    dvs::ProfScope prof(dvs::SLOT_CONV_WGRAD, st);       // flops of the direct weight gradient
    prof.work(2.0 * B * H * W * Cout * (double)Cin * 9);
    launch_wino_wgrad<0>(x, dy, dw, B, H, W, Cin, Cout, Cin, target_workgroups, st);
    return dvs::check_launch("dvs_conv3x3_wino_wgrad");
}

int dvs_conv3x3_wino_wgrad_gen(const float* x, const float* x2, const float* dy, const float* y_out, float* dw, float* dbias, int B, int H, int W,
                               int C1, int C2, int Cout, int upsample, int dact, int target_workgroups, void* stream) {
    DVS_REQUIRE(x && dy && dw && B > 0 && H >= 2 && W >= 2, "dvs_conv3x3_wino_wgrad_gen: bad argument (ReflectionPad2d(1) needs H, W >= 2)");
    DVS_REQUIRE((C2 == 0) == (x2 == nullptr) && C2 >= 0, "dvs_conv3x3_wino_wgrad_gen: x2 and C2 go together");
    DVS_REQUIRE(C1 > 0 && C1 % 32 == 0 && C2 % 32 == 0 && Cout > 0 && Cout % 32 == 0,
                "dvs_conv3x3_wino_wgrad_gen: channel counts must be multiples of 32 (got %d + %d, %d)", C1, C2, Cout);
    DVS_REQUIRE(!upsample || ((H & 1) == 0 && (W & 1) == 0), "dvs_conv3x3_wino_wgrad_gen: an upsampled input has even H, W");
    DVS_REQUIRE(upsample || C2 == 0, "dvs_conv3x3_wino_wgrad_gen: a second source comes with the upsampled first one");
    DVS_REQUIRE(dact == 0 || ((dact == kActRelu || dact == kActElu) && y_out),
                "dvs_conv3x3_wino_wgrad_gen: activation %d (0, 1 = ReLU, 2 = ELU; with the forward output)", dact);
    DVS_REQUIRE(!dbias || dact, "dvs_conv3x3_wino_wgrad_gen: the bias gradient rides on the activation-derivative path (dact != 0)");
    const int cmax = (C1 > C2 ? C1 : C2) > Cout ? (C1 > C2 ? C1 : C2) : Cout;
    DVS_REQUIRE(((double)B * H * W + 3 * W + 3) * cmax * 4 < 2147483648.0,
                "dvs_conv3x3_wino_wgrad_gen: tensors must be smaller than 2 GiB (32-bit buffer offsets)");
    hipStream_t st = static_cast<hipStream_t>(stream);
    dvs::ProfScope prof(dvs::SLOT_CONV_WGRAD, st);       // flops of the direct weight gradient
    prof.work(2.0 * B * H * W * Cout * (double)(C1 + C2) * 9);
    const int Ct = C1 + C2, tw = target_workgroups;
    if (dact) {
        if (upsample) launch_wino_wgrad<2, true>(x, dy, dw, B, H, W, C1, Cout, Ct, tw, st, y_out, dact, dbias);
        else launch_wino_wgrad<1, true>(x, dy, dw, B, H, W, C1, Cout, Ct, tw, st, y_out, dact, dbias);
        if (C2) launch_wino_wgrad<1, true>(x2, dy, dw + C1, B, H, W, C2, Cout, Ct, tw, st, y_out, dact, nullptr);
    } else {
        if (upsample) launch_wino_wgrad<2>(x, dy, dw, B, H, W, C1, Cout, Ct, tw, st);
        else launch_wino_wgrad<1>(x, dy, dw, B, H, W, C1, Cout, Ct, tw, st);
        if (C2) launch_wino_wgrad<1>(x2, dy, dw + C1, B, H, W, C2, Cout, Ct, tw, st);
    }
    return dvs::check_launch("dvs_conv3x3_wino_wgrad_gen");
}

}  // extern "C"
