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
    int ksplit, grid0;            // KSPLIT kernels: the input channels are dealt to `ksplit` workgroups per tile block (grid0 each)
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
// KSPLIT: the workgroups of a tile block split the input channels and ADD their outputs into a zero-filled y with float atomics
// (no bias / activation / statistics in the kernel; the residual rides with split 0): layer 4 at batch 12 is 120 workgroups of 256
// k-steps for 256 CUs -- as 240 workgroups of 128 it keeps every CU busy (launch_wino decides).
template <int WT, int WC, int DBG = 0, int GEN = 0, bool RES = false, bool KSPLIT = false>     // RES: y += res (same shape); DBG (tools/wino_bench.py): 1 no U loads, 2 no A reads, 4 no staging, 8 no barrier in the loop, 16 no epilogue
__global__ __launch_bounds__(NT, 1) void wino_fwd_kernel(WinoParams p) {
    static_assert(WT * WC == 4 && (WT == 1 || WT == 2), "four waves");
    constexpr int MT = 32 * WT, CK = 8, CKP = CK + 4, KH = 4, VEC = WT, PP = CK / VEC;
    static_assert(MT * PP == NT, "one staging item per thread");
    __shared__ __attribute__((aligned(16))) float sV[3][16][MT][CKP];
    __shared__ int sT[MT];
    __shared__ unsigned sTb[MT];       // byte offset of the tile's first output pixel (channel 0)
    using f32x2 = __attribute__((ext_vector_type(2))) float;

    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wt = wave / WC, wc = wave % WC;
    const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
    const int Ho = GEN ? p.Ho : H, Wo = GEN ? p.Wo : W, org = GEN ? p.org : 1;
    const int TXn = (Wo + 1) >> 1, TYn = (Ho + 1) >> 1, ntiles = p.B * TYn * TXn;
    // XCD-aware order: workgroups that share input tiles (the channel blocks of one tile block) run on the same XCD / L2
    const int nb = p.tiles_x, ntb = p.tiles_y;                        // channel blocks, tile blocks
    const int ks = KSPLIT ? (int)blockIdx.x / p.grid0 : 0;           // my share of the input channels
    const int bid = KSPLIT ? (int)blockIdx.x - ks * p.grid0 : (int)blockIdx.x;
    const int xcd = bid & 7, slot = bid >> 3;
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
        sTb[tid] = (unsigned)((bb * Ho + 2 * ty) * Wo + 2 * tx) * (unsigned)Cout * 4u;
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
    const int nchunk = KSPLIT ? Cin / CK / p.ksplit : Cin / CK, chunk0 = ks * nchunk;      // my chunks: [chunk0, chunk0 + nchunk)
    auto load_pixel = [&](int chunk, int i, int j) -> vec_t {
        int soff = (chunk0 + chunk) * CK * 4;
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
        const int soff = ((chunk0 + chunk) * CK + j) * u_kstride;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // (bit_cast of the builtin's own vector type: an implicit conversion to an ext_vector_type splats element 0)
            un[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ur, u_voff, soff + q * u_qstride, 0));
        }
    };
    load_u(0, 0);

    vec_t stg[4][4], tt[4][4];
    f32x4 a[16], an[16];
    auto read_a = [&](int buf, f32x4 (&dst)[16]) {
#pragma unroll
        for (int q = 0; q < 16; ++q) dst[q] = *reinterpret_cast<const f32x4*>(&sV[buf][q][32 * wt + r][h * KH]);
    };
    // prologue: chunks 0 and 1 into buffers 0 and 1 (both loads in flight together), chunk 2 loaded (chunks past the end
    // re-read the last one; never used)
    f32x16 acc[16];
    {
        vec_t stg1[4][4];
        load_stage(0, stg);
        load_stage(1, stg1);                              // nchunk >= 2
        // the 256 accumulator writes (0.5 us of issue) go HERE, under the flight time of the first two chunks -- left to itself the
        // compiler put half of them in front of these loads and the other half in front of the first MFMA
        // (asm: a plain `= 0.f` is a rematerialisable constant that the register allocator sinks to the first MFMA)
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float z;
                asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z));
                acc[q][i] = z;
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) transform_piece(k, 0, stg, tt);
        load_stage(min(2, nchunk - 1), stg);
#pragma unroll
        for (int k = 0; k < 4; ++k) transform_piece(k, 1, stg1, tt);
    }
    __syncthreads();
    read_a(0, a);

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
    const bool want_stats = !KSPLIT && p.stats != nullptr;
    int tinfo[16];
    unsigned tbase[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) tinfo[i] = sT[32 * wt + (i & 3) + 8 * (i >> 2) + 4 * h];
#pragma unroll
    for (int i = 0; i < 16; ++i) tbase[i] = sTb[32 * wt + (i & 3) + 8 * (i >> 2) + 4 * h] + (unsigned)co * 4u;
    float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};
    // (readfirstlane: a store's soffset must be an SGPR; once one of these lands in a vector register every store is a waterfall loop)
    const unsigned row_b = __builtin_amdgcn_readfirstlane((unsigned)Wo * Cout * 4), col_b = __builtin_amdgcn_readfirstlane((unsigned)Cout * 4);
    float rv[RES ? 16 : 1][4];                            // residual values: all 64 loads in flight before the transform arithmetic
    if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned ob = ((tinfo[i] >= 0) & co_ok) ? tbase[i] : 0xC0000000u;      // (rows / columns past an odd edge: masked at the store)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                rv[i][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, ob, (k >> 1) * row_b + (k & 1) * col_b, 0));
        }
        // (fence: the scheduler otherwise starts reading accumulators among these loads, runs out of registers and spills residual
        // values as they arrive -- every spill store waits for its load, i.e. the 64 loads went one at a time: + 40 % on layer 1)
        if constexpr (!KSPLIT) __builtin_amdgcn_sched_barrier(0);
    }
    // (a - b on a register pair as ONE v_pk_add_f32 with negated second operand: left to itself the compiler emits two v_sub_f32)
    auto pk_sub = [](f32x2 a, f32x2 b) __attribute__((always_inline)) {
        f32x2 d;
        asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
        return d;
    };
    auto out_transform = [&](int i, f32x2 (&y)[2][2]) __attribute__((always_inline)) {
        f32x2 s[2][4];
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {                  // A^T M
            const f32x2 m0{acc[0 + bq][i], acc[0 + bq][i + 1]}, m1{acc[4 + bq][i], acc[4 + bq][i + 1]};
            const f32x2 m2{acc[8 + bq][i], acc[8 + bq][i + 1]}, m3{acc[12 + bq][i], acc[12 + bq][i + 1]};
            s[0][bq] = m0 + m1 + m2;
            s[1][bq] = pk_sub(m1, m2 + m3);
        }
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2) {                  // (A^T M) A
            y[a2][0] = s[a2][0] + s[a2][1] + s[a2][2];
            y[a2][1] = pk_sub(s[a2][1], s[a2][2] + s[a2][3]);
        }
    };
    // Fast path (wave-uniform): every tile of the wave exists with both rows and columns, one statistics group, no bias / activation
    // -- the BasicBlock case.  A store is then the tile's offset + one of four SCALAR offsets, no select, no clamp; the statistics
    // need no masks.  (Every vector instruction here is time no matrix instruction runs in: 1 383 -> ~800 per wave.)
    bool fast = false;
    int g0 = 0;
    if constexpr (!KSPLIT) {
        g0 = __builtin_amdgcn_readfirstlane(tinfo[0]) & 4;
        bool mine = co_ok && p.bias == nullptr && !p.relu && (GEN == 0 || p.act == 0);
#pragma unroll
        for (int i = 0; i < 16; ++i) mine = mine & (tinfo[i] >= 0) & ((tinfo[i] & 7) == (3 | g0));      // (&: no branch chain)
        fast = __all(mine) != 0;
    }
    if (fast) {
        f32x2 fs{0.f, 0.f}, fq{0.f, 0.f};                  // statistics on register pairs (packed add / fma)
        auto fast_body = [&](auto with_stats) __attribute__((always_inline)) {
#pragma unroll
            for (int ip = 0; ip < 8; ++ip) {
                const int i = 2 * ip;
                f32x2 y[2][2];
                out_transform(i, y);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const f32x2 yk = y[k >> 1][k & 1];
                    if constexpr (decltype(with_stats)::value) {
                        fs += yk;
                        fq += yk * yk;
                    }
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        float yv = yk[e];
                        if constexpr (RES) yv += rv[i + e][k];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, yv), yr, tbase[i + e], (k >> 1) * row_b + (k & 1) * col_b, 0);
                    }
                }
                // RES: the 64 residual values already take a quarter of the vector registers; without a fence the scheduler starts the
                // accumulator reads of several tile pairs at once and spills (a reload then waits, in order, for every residual load)
                if constexpr (RES) __builtin_amdgcn_sched_barrier(0);
            }
        };
        if (want_stats) fast_body(std::true_type{});
        else fast_body(std::false_type{});
        const float f1 = fs[0] + fs[1], q1 = fq[0] + fq[1];
        ssum[0] = g0 ? 0.f : f1;
        ssq[0] = g0 ? 0.f : q1;
        ssum[1] = g0 ? f1 : 0.f;
        ssq[1] = g0 ? q1 : 0.f;
    } else
#pragma unroll
    for (int ip = 0; ip < 8; ++ip) {
        const int i = 2 * ip;
        f32x2 y[2][2];
        out_transform(i, y);
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
                    if constexpr (RES) yv += (KSPLIT && ks != 0) ? 0.f : rv[i + e][2 * a2 + c2];
                    if constexpr (KSPLIT) {
                        if (v[a2][c2]) atomicAdd(p.y + (off >> 2), yv);
                        continue;
                    }
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
    if (want_stats) {
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
// (output channel, input channel) pairs for all 16 components (256 accumulator registers) and a range of tile PAIRS; both operands
// are transformed in-lane from global memory (lane = one channel of one tile: 4 dY values -> 16, 4 x 4 input values -> 16; all
// loads are dword buffer loads that are contiguous over the 32 channels of a half wave).  No LDS and no barrier in the loop.
//
// Round 3: the fp32 MFMA runs on the vector ALU's multipliers (64 flop / clock / SIMD either way), so every vector instruction a
// wave issues is taken out of its own matrix time (counters of the round-2 kernel: 9 % of the wave cycles parked on a counter,
// 51 % waiting to ISSUE; 10.5 vector instructions per MFMA, of which 2.7 were the transforms).  What is left in the loop now is
// the two transforms (44 vector instructions per 16 MFMAs) and a handful of selects:
//   * a k-step is one PAIR of horizontally neighbouring tiles (tx = 2 txp + h, h = the lane half = the k index of the 32x32x2
//     MFMA; a row of tiles is padded to an even count, the pad tile reads zeros), so the pair's position is wave-uniform: the
//     byte offset of its patch origin lives in SGPRs, advances on the scalar unit and goes into the loads' soffset; a lane's
//     voffset -- channel, lane half, patch column -- is loop invariant;
//   * rows outside the image (or mirrored / clamped ones: MODE 1 / 2) are a scalar matter: the row's soffset, and for rows that
//     must read zeros a buffer resource with no records (selected on the scalar unit); columns outside the image only occur in the
//     first and the last pair of a tile row: the lanes' voffsets of those two cases are precomputed and selected by a scalar
//     condition (7 v_cndmask per k-step);
//   * the loaded values are transformed BEFORE their registers are loaded again (issue order is all the hardware needs), so the
//     round-2 staging moves (20 per k-step) are gone: step k transforms the operands of k+1, then issues the loads of k+3 into
//     the same registers between the MFMAs of k; a batch has almost two k-steps to arrive;
//   * the pure sign changes of A dY A^T (its last row and column) are left out of the loop and applied to the 7 affected
//     components in the epilogue.
// The four waves of a workgroup split the workgroup's pair range and add their accumulators through LDS; the first wave applies
// G^T . G in-lane (all 16 components of a (co, ci) pair sit in one lane) and adds the 3 x 3 results into dw -- plain adds when the
// range is not split over workgroups, float atomics otherwise; with a partial-sum workspace (deterministic mode) the workgroup
// stores its 32 x 32 x 9 block there instead and wino_wgrad_reduce_kernel adds the blocks up in a fixed order.
struct WinoWgradParams {
    const float* x;        // [B][H][W][Cin]
    const float* dy;       // [B][H][W][Cout]
    float* dw;             // [Cout][3][3][Cin], +=
    int B, H, W, Cin, Cout;
    int nblk_ci, nblk;     // 32-channel blocks of Cin; blocks of (Cout, Cin)
    int S;                 // splits of the pair range over workgroups
    int pairs_per_wg;      // multiple of 8 (four waves x an even number of k-steps)
    int CinW;              // row stride of dw in channels (= Cin, or the concatenated channel count when x is one of two sources)
    const float* yact;     // DACT: the forward output y [B][H][W][Cout]; dy is multiplied by act'(y) as it is loaded
    int dact;              // activation code of conv_common.h (1 ReLU, 2 ELU)
    float* dbias;          // DACT: null, or [Cout] += the column sums of dy act'(y) (taken by the workgroups of input-channel block 0)
    float* part;           // null, or the partial-sum workspace [nblk][S][9][32][32] (ordered reduction instead of atomics)
};

// MODE 0: zero padding 1 (the BasicBlock layers).  The decoder's gathers (model/layers.py:26-41, model/depth_decoder.py:52-62):
// MODE 1: ReflectionPad2d(1) -- a patch pixel outside the image moves two rows / columns back inside instead of reading zero;
// MODE 2: x is the half-resolution operand of the nearest 2x upsample, [B][H/2][W/2][Cin] -- the 4 x 4 patch of output tile
//         (ty, tx) is source pixels {ty-1, ty, ty, ty+1} x {tx-1, tx, tx, tx+1}, clamped at the border (= reflection of the
//         upsampled image): 9 loads instead of 16.
// The two sources of an upsample + concat layer are two launches, each adding into its own channel range of dw (CinW).
// DACT (the thin decoder layers, whose gradient kernels apply the activation derivative themselves): dZ = dY act'(Y) is formed from
// four more loads per tile and the bias gradient rides along.
// 16 bytes per lane straight into LDS (64 lanes land lane-linear at M0 + IMM).  In a __device__ function of its own: called from the
// kernel body directly, the host pass of hipcc (which has no gfx950 target feature for the 16-byte form) silently drops the
// kernel's launch stub and the library fails to load with an undefined __device_stub__ symbol.
template <int IMM>
__device__ __forceinline__ void wino_dma16(__amdgpu_buffer_rsrc_t rs, float* lds, unsigned voff, unsigned soff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, IMM, 0);
}

template <int MODE, bool DACT = false>
__global__ __launch_bounds__(NT, 1) void wino_wgrad_kernel(WinoWgradParams p) {
    // 128 KB: in the loop the four waves' DMA rings (RING slots of NL x 64 floats each), afterwards the accumulators of two waves
    // during the reduction.  ONE shared array: a second one beside an LDS-DMA target makes hipcc drain vmcnt before every ds_read.
    __shared__ float sMem[2 * 256 * 64];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
    const int TXn = (W + 1) >> 1, TYn = (H + 1) >> 1, NP = (TXn + 1) >> 1, npairs = p.B * TYn * NP;
    // workgroups of one XCD walk the channel blocks of the same pair range first (its x / dY stay in that L2); with fewer than
    // eight ranges (S = 1, 2 or 4) 8 / S XCDs share a range and deal its channel blocks between them
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
    const int per_wave = p.pairs_per_wg >> 2;              // even
    const int q_begin = sp * p.pairs_per_wg + wave * per_wave;
    const int nk = max(0, min(q_begin + per_wave, npairs) - q_begin);      // pairs (= k-steps) of this wave

    constexpr int NC = MODE == 2 ? 3 : 4;                  // patch rows / columns of one tile
    constexpr int NX = NC * NC;
    constexpr int NY = DACT ? 8 : 4;                       // dY (and Y) values of one tile
    const int Hs = MODE == 2 ? H >> 1 : H, Ws = MODE == 2 ? W >> 1 : W;      // geometry of x
    const unsigned Cin4 = (unsigned)Cin * 4u, Cout4 = (unsigned)Cout * 4u, RowB = (unsigned)Ws * Cin4;
    // One batch = the UNION of the pair's two patches, pixel by pixel: NROW rows x ROWP columns of x (4 x 6, the half-resolution
    // source of MODE 2: 3 x 4) and the 2 x 4 pixels of dY (and Y), each pixel the 32 channels (128 bytes) of the block.  A DMA
    // instruction moves 16 bytes per lane = 8 pixels: 3 (2) + 1 (+ 1) instructions per k-step where one dword per lane took 20 (24),
    // and a pixel that both tiles read is fetched once.  Lane l of DMA d fetches channels 4 (l & 7) .. + 3 of pixel 8 d + (l >> 3);
    // the LDS image of a batch is [pixel][32 channels].
    // Addressing: address = resource base + soffset (scalar: the pair's origin) + voffset (per lane: channel quad, row and column of
    // its pixel inside the union) + the instruction's immediate (which only places the DMA's 1 KB inside the LDS image and is taken
    // back out of the voffset).  The voffset counts rows from the row ABOVE the pair and columns from the column LEFT of it, so the x
    // resource starts one source row and one pixel in front of the tensor -- and another DLEAD bytes earlier for the immediates; what
    // lies there is never read (row / column -1 are masked, mirrored or clamped).
    constexpr int ROWP = MODE == 2 ? 4 : 6, NPIX = NC * ROWP, ND = (NPIX + 7) / 8;
    constexpr unsigned DLEAD = 4096;
    const size_t lead = (size_t)(Ws + 1) * Cin4 + DLEAD;
    const unsigned xbytes = (unsigned)((size_t)p.B * Hs * Ws * Cin4 + lead), ybytes = (unsigned)((size_t)p.B * H * W * Cout4 + DLEAD);
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(p.x)) - lead, 0, (int)xbytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.dy)) - DLEAD, 0, (int)ybytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(DACT ? p.yact : p.dy)) - DLEAD, 0, (int)ybytes, 0x00020000);
    // marks of a masked column / row: offsets stay below 2^30 (the launcher checks the tensors), so no sum of an offset and marks
    // wraps, and every marked sum lies beyond the resource's records: the load returns 0
    constexpr unsigned OOB_COL = 0x40000000u, OOB_ROW = 0x80000000u;
    float bsum = 0.f;                                      // DACT: my channel's sum of dZ over my tiles

    // ---- per-lane voffsets of the ND + 1 DMAs.  vM: a pair in the middle of a tile row in the middle of the image; dF / dL / dT / dB:
    // what the first / last pair of a tile row and the top / bottom tile row add to it (a mark, or the distance to the mirrored /
    // clamped pixel).  The cases touch different lanes (column 0 vs the last columns, row 0 vs the last rows), so they simply add up.
    const int g8 = lane >> 3, cq = lane & 7;
    unsigned vM[ND + 1], dF[ND + 1], dL[ND + 1], dT[ND + 1], dB[ND + 1];
#pragma unroll
    for (int d = 0; d < ND; ++d) {
        const int pix = 8 * d + g8, pi = pix / ROWP, pc = pix - pi * ROWP;
        const bool used = pix < NPIX;
        vM[d] = used ? (unsigned)(ci0 + 4 * cq) * 4u + DLEAD + (unsigned)pi * RowB + (unsigned)pc * Cin4 - (unsigned)(d * 1024) : OOB_COL;
        // columns: union column pc is source column cstep * txp - 1 + pc
        constexpr int cstep = MODE == 2 ? 2 : 4;
        const int cl = cstep * (NP - 1) - 1 + pc;          // in the last pair
        int df = 0, dl = 0;
        if constexpr (MODE == 0) {
            df = pc == 0 ? (int)OOB_COL : 0;
            dl = cl >= Ws ? (int)OOB_COL : 0;
        } else if constexpr (MODE == 1) {
            df = pc == 0 ? 2 * (int)Cin4 : 0;              // column -1 -> 1
            const int cm = max(2 * (Ws - 1) - cl, cstep * (NP - 1) - 1);     // (column W + 1 only meets a masked dY column: any finite value does)
            dl = cl >= Ws ? (cm - cl) * (int)Cin4 : 0;
        } else {
            df = pc == 0 ? (int)Cin4 : 0;                  // column -1 -> 0
            dl = cl >= Ws ? (Ws - 1 - cl) * (int)Cin4 : 0;
        }
        // rows: union row pi is source row rstep * ty - 1 + pi
        constexpr int rstep = MODE == 2 ? 1 : 2;
        const int rl = rstep * (TYn - 1) - 1 + pi;         // in the bottom tile row
        int dt = 0, db = 0;
        if constexpr (MODE == 0) {
            dt = pi == 0 ? (int)OOB_ROW : 0;
            db = rl >= Hs ? (int)OOB_ROW : 0;
        } else if constexpr (MODE == 1) {
            dt = pi == 0 ? 2 * (int)RowB : 0;              // row -1 -> 1
            const int rm = max(2 * (Hs - 1) - rl, rstep * (TYn - 1) - 1);
            db = rl >= Hs ? (rm - rl) * (int)RowB : 0;
        } else {
            dt = pi == 0 ? (int)RowB : 0;                  // row -1 -> 0
            db = rl >= Hs ? (Hs - 1 - rl) * (int)RowB : 0;
        }
        dF[d] = used ? (unsigned)df : 0u;
        dL[d] = used ? (unsigned)dl : 0u;
        dT[d] = used ? (unsigned)dt : 0u;
        dB[d] = used ? (unsigned)db : 0u;
    }
    {   // dY (and Y): pixel (row g8 >> 2, column g8 & 3) of the pair's 2 x 4 outputs
        const int pa = g8 >> 2, pc = g8 & 3;
        vM[ND] = (unsigned)(co0 + 4 * cq) * 4u + DLEAD + (unsigned)(pa * W + pc) * Cout4;
        dF[ND] = 0u;
        dL[ND] = 4 * (NP - 1) + pc >= W ? OOB_COL : 0u;
        dT[ND] = 0u;
        dB[ND] = 2 * (TYn - 1) + pa >= H ? OOB_ROW : 0u;
    }

    // ---- the wave's position: pair q = (tb, ty, txp); everything about it is wave-uniform (scalar unit)
    int tb, ty, txp;
    {
        const int q = min(q_begin, max(npairs - 1, 0));
        tb = q / (TYn * NP);
        const int rem = q - tb * (TYn * NP);
        ty = rem / NP;
        txp = rem - ty * NP;
        // the divisions are expanded on the vector unit; without these the whole scalar chain below follows them into VGPRs and
        // every buffer load is wrapped in a waterfall loop
        tb = __builtin_amdgcn_readfirstlane(tb);
        ty = __builtin_amdgcn_readfirstlane(ty);
        txp = __builtin_amdgcn_readfirstlane(txp);
    }
    // The fp32 MFMA holds the wave's issue for its 64 cycles (counters: the MFMA, vector, scalar and memory issue cycles of a wave ADD
    // UP to its run time at one wave per SIMD), so what counts is the NUMBER of instructions per k-step, of any kind.  The voffsets of
    // a batch therefore live in registers (cv) that are only rewritten where the pair's class changes -- entering a tile row, leaving
    // its first pair, entering its last pair: a handful of k-steps per tile row, in a branch the others skip -- and the soffsets (the
    // pair's origin) take one scalar add each per k-step.
    unsigned cv[ND + 1], sxb = 0, syb = 0;
    int kq = 0;                                            // k-step (of this wave) at (tb, ty, txp)
    auto origin = [&]() __attribute__((always_inline)) {
        if constexpr (MODE == 2) sxb = (unsigned)((tb * Hs + ty) * Ws + 2 * txp) * Cin4;
        else sxb = (unsigned)((tb * H + 2 * ty) * W + 4 * txp) * Cin4;
        syb = (unsigned)((tb * H + 2 * ty) * W + 4 * txp) * Cout4;
    };
    auto refresh = [&]() __attribute__((always_inline)) {
        const bool live = kq < nk, first = txp == 0, last = txp == NP - 1, top = ty == 0, bot = ty == TYn - 1;
#pragma unroll
        for (int d = 0; d <= ND; ++d) {
            const unsigned m = vM[d], f = dF[d], l = dL[d];     // values first, then the selects: a ?: over the captured variables
            cv[d] = m + (first ? f : 0u) + (last ? l : 0u);     // themselves selects between their ADDRESSES inside the closure
        }
        if (top || bot || !live) {
#pragma unroll
            for (int d = 0; d <= ND; ++d) {
                const unsigned t = dT[d], b = dB[d];
                cv[d] += (top ? t : 0u) + (bot ? b : 0u);
                cv[d] |= live ? 0u : OOB_ROW;              // a k-step past the wave's range reads zeros
            }
        }
    };
    // to the next pair (an interior k-step: four scalar adds and the compares that fall through)
    auto advance = [&]() __attribute__((always_inline)) {
        ++kq;
        ++txp;
        sxb += (MODE == 2 ? 2u : 4u) * Cin4;
        syb += 4u * Cout4;
        if (txp == NP) {
            txp = 0;
            ++ty;
            if (ty == TYn) {
                ty = 0;
                ++tb;
            }
            origin();
        }
        if (txp <= 1 || txp == NP - 1 || kq == nk) refresh();      // entering a tile row, leaving its first pair, entering its last
    };

    constexpr int XS = ND * 8, SLOTF = (XS + 8 + (DACT ? 8 : 0)) * 32, NDMA = ND + 1 + (DACT ? 1 : 0), RING = 4;
    float* const ring = sMem + wave * (RING * SLOTF);     // ring slots = prefetch distance (3 k-steps) + 1
    auto issue_dma = [&](int slot) __attribute__((always_inline)) {
        float* const dst = ring + slot * SLOTF;
        wino_dma16<0>(yr, dst + XS * 32, cv[ND], syb);
        if constexpr (DACT) wino_dma16<0>(ar, dst + (XS + 8) * 32, cv[ND], syb);
        wino_dma16<0>(xr, dst, cv[0], sxb);
        wino_dma16<1024>(xr, dst, cv[1], sxb);
        if constexpr (ND > 2) wino_dma16<2048>(xr, dst, cv[ND > 2 ? 2 : 0], sxb);
    };
    // my values of a batch: the lane's tile starts (2 h) -- MODE 2: h -- pixels into the union's rows
    const int bx = (MODE == 2 ? h : 2 * h) * 32 + r, by = 2 * h * 32 + r;
    auto read_slot = [&](const float* src, float (&xv)[NX], float (&yv)[NY]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            yv[e] = src[by + (XS + 4 * (e >> 1) + (e & 1)) * 32];
            if constexpr (DACT) yv[4 + e] = src[by + (XS + 8 + 4 * (e >> 1) + (e & 1)) * 32];
        }
#pragma unroll
        for (int e = 0; e < NX; ++e) xv[e] = src[bx + ((e / NC) * ROWP + (e % NC)) * 32];
    };
    // v = B^T d B;  pm = |A| dY |A|^T: rows (1,0), (1,1), (1,-1), (0,1) -- the true A has (0,-1) as its last row, i.e. components
    // (3, l) and (i, 3) carry a factor -1 each that the epilogue applies (kWgradSign)
    auto transform = [&](const float (&xs)[NX], const float (&yl)[NY], float (&v)[16], float (&pm)[16]) __attribute__((always_inline)) {
        float tt[4][4], xc[16], yc[4];
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
        // |A| dY: rows y0, y0 + y1, y0 - y1, y1 (per column j); then the same along the columns
        const float pr[4][2] = {{yc[0], yc[1]}, {yc[0] + yc[2], yc[1] + yc[3]}, {yc[0] - yc[2], yc[1] - yc[3]}, {yc[2], yc[3]}};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            pm[4 * i + 0] = pr[i][0];
            pm[4 * i + 1] = pr[i][0] + pr[i][1];
            pm[4 * i + 2] = pr[i][0] - pr[i][1];
            pm[4 * i + 3] = pr[i][1];
        }
    };

    f32x16 acc[16];

    // Step k:  DMA of k-step k+3 -> the slot that held k-1 (read and transformed during step k-2)  |  the first MFMAs of k  |  wait
    // until the batch of k+1 has landed (two younger batches stay in flight)  |  its 20 values LDS -> registers  |  the other MFMAs  |
    // transform of k+1 into the operand registers the MFMAs have just read.
    float tx[NX], ty_[NY], v[16], pm[16];
#define WINO_WAIT_VMCNT(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
    origin();
    refresh();
    int slot_w = 0;                                        // slot the next batch goes to
#pragma unroll
    for (int j = 0; j < RING - 1; ++j) {                   // k-steps 0, 1, 2
        if (j) advance();
        issue_dma(slot_w);
        slot_w = (slot_w + 1) & (RING - 1);
    }
    // the 256 accumulator writes (0.5 us of issue) under the flight time of the first batches (asm: a plain `= 0.f` is a
    // rematerialisable constant that the register allocator sinks to the first MFMA)
#pragma unroll
    for (int q = 0; q < 16; ++q)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            float z;
            asm volatile("v_accvgpr_write_b32 %0, 0" : "=a"(z));
            acc[q][i] = z;
        }
    WINO_WAIT_VMCNT((RING - 2) * NDMA);                    // k-step 0 has landed
    read_slot(ring, tx, ty_);
    transform(tx, ty_, v, pm);
    int slot_r = 1;                                        // slot of k-step k+1
    const float* rd = ring + slot_r * SLOTF;
    for (int k = 0; k < nk; ++k) {
        advance();                                         // (at the head of the step: behind the transform the compiler sank the
        __builtin_amdgcn_sched_barrier(0);                 // transform -- and with it the LDS reads -- below advance's branches)
        issue_dma(slot_w);
#pragma unroll
        for (int q = 0; q < 6; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(pm[q], v[q], acc[q], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);     // VMEM read (the DMAs)
        }
        __builtin_amdgcn_sched_barrier(0);
        WINO_WAIT_VMCNT((RING - 2) * NDMA);
        read_slot(rd, tx, ty_);
#pragma unroll
        for (int q = 6; q < 16; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(pm[q], v[q], acc[q], 0, 0, 0);
#pragma unroll
        for (int i = 6; i < 16; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // MFMA
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);     // DS read
        }
        __builtin_amdgcn_sched_barrier(0);
        transform(tx, ty_, v, pm);
        slot_w = (slot_w + 1) & (RING - 1);
        slot_r = (slot_r + 1) & (RING - 1);
        rd = ring + slot_r * SLOTF;
    }
    __builtin_amdgcn_sched_barrier(0);
    WINO_WAIT_VMCNT(0);                                    // the batches past the end (they read zeros) must not land in sR
#undef WINO_WAIT_VMCNT
    __syncthreads();                                       // every wave is done with its ring: the reduction may overwrite it

    if constexpr (DACT) {
        // every tile's dZ went through transform() exactly once per (output-channel, input-channel) block: block column 0 reports
        // (the batches loaded beyond the wave's last k-step read zeros: dead k-steps have no records)
        if (p.dbias && ci0 == 0) {
            const float b2 = bsum + __shfl_xor(bsum, 32, 64);
            if (h == 0) atomicAdd(p.dbias + co0 + r, b2);
        }
    }
    // ---- add the four waves' accumulators, in two passes over the register index (i < 8, then i >= 8): every wave stores its 128
    // values of the pass (all 16 components) in LDS, and after a barrier wave w collects registers 2w, 2w + 1 of that half from all
    // four waves -- in a fixed order: a workgroup's sum does not depend on timing --, applies G^T . G in-lane (the 16 components of a
    // (co, ci) pair share a lane) and adds its 2 x 9 values per lane into dw.  Identical code in every wave (which registers a wave
    // collects is an LDS address, not a register index), and all four SIMDs work through the whole epilogue: round 2's tree
    // (waves 2, 3 -> waves 0, 1 -> wave 0, which then transformed and stored all 16 registers alone) left three of them idle for
    // most of an epilogue that is a fifth of a batch-12 launch.
    float (*const sP)[128][64] = reinterpret_cast<float (*)[128][64]>(sMem);     // [wave][q * 8 + (i & 7)][lane]: 32 KB each
    const bool exclusive = p.S == 1;                       // this workgroup alone owns its block of dw
    float* const part = p.part ? p.part + ((size_t)blk * p.S + sp) * (9 * 1024) : nullptr;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();                         // the first pass's reads are done
#pragma unroll
        for (int q = 0; q < 16; ++q)
#pragma unroll
            for (int i8 = 0; i8 < 8; ++i8) sP[wave][q * 8 + i8][lane] = acc[q][8 * half + i8];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int i = 8 * half + 2 * wave + e;         // (wave-uniform, only used in addresses)
            const int m = (i & 3) + 8 * (i >> 2) + 4 * h;  // output channel row of the block
            float u[4][4], t3[3][4];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float* c = &sP[0][q * 8 + 2 * wave + e][lane];
                const float sv = ((c[0] + c[128 * 64]) + c[2 * 128 * 64]) + c[3 * 128 * 64];
                u[q >> 2][q & 3] = ((q >> 2) == 3) != ((q & 3) == 3) ? -sv : sv;   // the signs left out of the loop's dY transform
            }
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
                    if (part) {
                        part[((3 * k + l) * 32 + m) * 32 + r] = w3[l];
                        continue;
                    }
                    float* a = o + (size_t)(3 * k + l) * p.CinW;
                    if (exclusive) *a += w3[l];
                    else atomicAdd(a, w3[l]);
                }
            }
        }
    }
}

// Ordered second pass of the deterministic weight gradient: dw[co][tap][ci] += sum over the S partial blocks, in split order.
// One thread per element of a (32 x 32 x 9) block; grid = nblk * 9 * 4 workgroups of 256.
__global__ __launch_bounds__(NT) void wino_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nblk_ci, int S,
                                                               int CinW) {
    const int blk = blockIdx.x / 36, e = (blockIdx.x % 36) * NT + threadIdx.x;      // e = (tap * 32 + m) * 32 + r
    const int tap = e >> 10, m = (e >> 5) & 31, r = e & 31;
    const float* src = part + (size_t)blk * S * (9 * 1024) + e;
    float s = 0.f;
    for (int i = 0; i < S; ++i) s += src[(size_t)i * (9 * 1024)];
    const int co = (blk / nblk_ci) * 32 + m, ci = (blk % nblk_ci) * 32 + r;
    dw[((size_t)co * 9 + tap) * CinW + ci] += s;
}

// Helpers of the K-split launches: zero fill of y, and the BatchNorm statistics (sum, sum of squares per channel and group) of the
// finished y -- what the single-launch kernel takes in its epilogue.  One workgroup per slab of rows, float4 per lane, one atomic
// per channel and workgroup into copy 0 of the statistics table.
__global__ __launch_bounds__(NT) void wino_zero_kernel(f32x4* __restrict__ y, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += (size_t)gridDim.x * NT) y[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__global__ __launch_bounds__(NT) void wino_stats_kernel(const float* __restrict__ y, float* __restrict__ stats, int rows, int C, int rows_per_wg,
                                                        int split_row, int groups) {
    __shared__ float sAcc[2][2][1024];                     // [group][sum / sumsq][channel], C <= 1024
    const int c4n = C >> 2, lanes_rows = NT / c4n;         // C / 4 divides 256 (the launcher checks)
    const int c4 = threadIdx.x % c4n, rsub = threadIdx.x / c4n;
    for (int i = threadIdx.x; i < 4 * 1024; i += NT) (&sAcc[0][0][0])[i] = 0.f;
    __syncthreads();
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    f32x4 s[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, q[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 4
    for (int rw = r0 + rsub; rw < r1; rw += lanes_rows) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(y + (size_t)rw * C + 4 * c4);
        const int g = rw >= split_row ? 1 : 0;
        s[g] += v;
        q[g] += v * v;
    }
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(&sAcc[g][0][4 * c4 + e], s[g][e]);
            atomicAdd(&sAcc[g][1][4 * c4 + e], q[g][e]);
        }
    __syncthreads();
    for (int i = threadIdx.x; i < groups * 2 * C; i += NT) {
        const int g = i / (2 * C), k = (i / C) & 1, c = i % C;
        const float v = sAcc[g][k][c];
        if (v != 0.f) atomicAdd(stats + (size_t)g * 2 * C + k * C + c, v);
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
        // K split (see the kernel): wide layers whose tile blocks fill less than ~0.6 of the CUs, no bias / activation in the epilogue
        static const bool ks_on = [] { const char* e = getenv("DVS_WINO_KSPLIT"); return !(e && e[0] == '0'); }();
        if constexpr (WT == 1 && WC == 4) {
            const int nchunk = p.Cin / 8;
            if (ks_on && !dvs::deterministic() && dbg == 0 && !p.bias && !p.relu && p.act == 0 && grid <= 152 && nchunk % 4 == 0 && nchunk >= 16 &&
                (p.Cout & 3) == 0 && 256 % (p.Cout >> 2) == 0 && p.Cout <= 1024) {
                p.ksplit = 2;
                p.grid0 = (int)grid;
                const size_t n4 = (size_t)p.B * p.H * p.W * p.Cout / 4;
                hipLaunchKernelGGL(wino_zero_kernel, dim3(256), dim3(NT), 0, st, reinterpret_cast<f32x4*>(p.y), n4);
                if (p.res) hipLaunchKernelGGL((wino_fwd_kernel<WT, WC, 0, 0, true, true>), dim3((unsigned)grid * 2), dim3(NT), 0, st, p);
                else hipLaunchKernelGGL((wino_fwd_kernel<WT, WC, 0, 0, false, true>), dim3((unsigned)grid * 2), dim3(NT), 0, st, p);
                if (p.stats) {
                    // (16 rows per workgroup: with 128 the 3 600 rows of layer 4 made 29 workgroups whose lanes walked 64 dependent
                    // iterations each -- 33 us for 7 MB; off the critical path: the step time did not move)
                    const int rows = p.B * p.H * p.W, rpw = 16;
                    const int groups = p.stat_split == 0x7fffffff ? 1 : 2;
                    const int split_row = groups == 2 ? p.stat_split * p.H * p.W : 0x7fffffff;
                    hipLaunchKernelGGL(wino_stats_kernel, dim3((unsigned)((rows + rpw - 1) / rpw)), dim3(NT), 0, st, p.y, p.stats, rows, p.Cout,
                                       rpw, split_row, groups);
                }
                return;
            }
        }
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
// pair ranges of one launch: S splits of pairs_per_wg tile pairs each (pairs of a tile row padded to an even tile count)
struct WgradSplit {
    int S, ppw;
};
inline WgradSplit wino_wgrad_split(int B, int H, int W, int Cin, int Cout, int target_workgroups) {
    const int nblk = (Cin / 32) * (Cout / 32);
    const int TXn = (W + 1) / 2, TYn = (H + 1) / 2, NP = (TXn + 1) / 2, npairs = B * TYn * NP;
    // default: one round of one workgroup per CU (fewer, longer ranges: less reduction and atomic traffic); the 512-channel
    // layers take two ranges so that one XCD's share of x and dY fits its L2 (measured: profiles/r02_f_wino_wgrad_split.txt)
    if (target_workgroups <= 0) target_workgroups = nblk >= 256 ? 512 : 256;
    int S = (target_workgroups + nblk / 2) / nblk;
    S = S < 1 ? 1 : S;
    int ppw = ((npairs + S - 1) / S + 7) & ~7;            // pairs per workgroup: four waves x an even number of k-steps
    ppw = ppw < 32 ? 32 : ppw;                             // at least eight k-steps per wave
    S = (npairs + ppw - 1) / ppw;
    if (S < 8) {                                           // 1, 2 or 4 ranges (the XCD map of the kernel); ranges past the end add zeros
        S = S >= 4 ? 4 : S >= 2 ? 2 : 1;
        ppw = ((npairs + S - 1) / S + 7) & ~7;
    }
    return {S, ppw};
}

// x: the source tensor of this launch (MODE 2: at half resolution); dw: already offset to the source's first channel, CinW its row stride
template <int MODE, bool DACT = false>
void launch_wino_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int CinW, int target_workgroups,
                       hipStream_t st, const float* yact = nullptr, int dact = 0, float* dbias = nullptr, float* part = nullptr) {
    WinoWgradParams p{x, dy, dw, B, H, W, Cin, Cout, Cin / 32, (Cin / 32) * (Cout / 32), 0, 0, CinW, yact, dact, dbias, part};
    const WgradSplit sp = wino_wgrad_split(B, H, W, Cin, Cout, target_workgroups);
    p.pairs_per_wg = sp.ppw;
    p.S = sp.S;
    const int S = sp.S;
    const size_t grid = S >= 8 ? (size_t)((S + 7) / 8) * 8 * p.nblk : (size_t)((p.nblk + 8 / S - 1) / (8 / S)) * 8;
    hipLaunchKernelGGL((wino_wgrad_kernel<MODE, DACT>), dim3((unsigned)grid), dim3(NT), 0, st, p);
    if (part)
        hipLaunchKernelGGL(wino_wgrad_reduce_kernel, dim3((unsigned)(p.nblk * 36)), dim3(NT), 0, st, part, dw, p.nblk_ci, S, CinW);
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

size_t dvs_conv3x3_wino_wgrad_workspace(int B, int H, int W, int Cin, int Cout, int target_workgroups) {
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 32) return 0;
    const WgradSplit sp = wino_wgrad_split(B, H, W, Cin, Cout, target_workgroups);
    return (size_t)(Cin / 32) * (Cout / 32) * sp.S * 9 * 1024 * sizeof(float);
}

int dvs_conv3x3_wino_wgrad_ws(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int target_workgroups,
                              float* workspace, size_t workspace_bytes, void* stream) {
    DVS_REQUIRE(x && dy && dw && B > 0 && H > 0 && W > 0, "dvs_conv3x3_wino_wgrad: bad argument");
    DVS_REQUIRE(Cin % 32 == 0 && Cout % 32 == 0 && Cin > 0 && Cout > 0, "dvs_conv3x3_wino_wgrad: channel counts must be multiples of 32 (got %d, %d)",
                Cin, Cout);
    DVS_REQUIRE(((double)B * H * W + W + 1) * (Cin > Cout ? Cin : Cout) * 4 + 8192 < 1073741824.0,
                "dvs_conv3x3_wino_wgrad: tensors must be smaller than 1 GiB (32-bit buffer offsets with two mask bits)");
    DVS_REQUIRE(!workspace || workspace_bytes >= dvs_conv3x3_wino_wgrad_workspace(B, H, W, Cin, Cout, target_workgroups),
                "dvs_conv3x3_wino_wgrad_ws: workspace of %zu bytes is too small", workspace_bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dvs::ProfScope prof(dvs::SLOT_CONV_WGRAD, st);       // flops of the direct weight gradient
    prof.work(2.0 * B * H * W * Cout * (double)Cin * 9);
    launch_wino_wgrad<0>(x, dy, dw, B, H, W, Cin, Cout, Cin, target_workgroups, st, nullptr, 0, nullptr, workspace);
    return dvs::check_launch("dvs_conv3x3_wino_wgrad");
}

int dvs_conv3x3_wino_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int target_workgroups,
                           void* stream) {
    return dvs_conv3x3_wino_wgrad_ws(x, dy, dw, B, H, W, Cin, Cout, target_workgroups, nullptr, 0, stream);
}

int dvs_conv3x3_wino_wgrad_gen_ws(const float* x, const float* x2, const float* dy, const float* y_out, float* dw, float* dbias, int B, int H,
                                  int W, int C1, int C2, int Cout, int upsample, int dact, int target_workgroups, float* workspace,
                                  size_t workspace_bytes, void* stream) {
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
    DVS_REQUIRE(((double)B * H * W + 3 * W + 3) * cmax * 4 + 8192 < 1073741824.0,
                "dvs_conv3x3_wino_wgrad_gen: tensors must be smaller than 1 GiB (32-bit buffer offsets with two mask bits)");
    DVS_REQUIRE(!workspace || workspace_bytes >= dvs_conv3x3_wino_wgrad_workspace(B, H, W, C1 > C2 ? C1 : C2, Cout, target_workgroups),
                "dvs_conv3x3_wino_wgrad_gen_ws: workspace of %zu bytes is too small", workspace_bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dvs::ProfScope prof(dvs::SLOT_CONV_WGRAD, st);       // flops of the direct weight gradient
    prof.work(2.0 * B * H * W * Cout * (double)(C1 + C2) * 9);
    const int Ct = C1 + C2, tw = target_workgroups;
    float* const ws = workspace;                           // the two sources' launches run one after the other on `st`: one workspace
    if (dact) {
        if (upsample) launch_wino_wgrad<2, true>(x, dy, dw, B, H, W, C1, Cout, Ct, tw, st, y_out, dact, dbias, ws);
        else launch_wino_wgrad<1, true>(x, dy, dw, B, H, W, C1, Cout, Ct, tw, st, y_out, dact, dbias, ws);
        if (C2) launch_wino_wgrad<1, true>(x2, dy, dw + C1, B, H, W, C2, Cout, Ct, tw, st, y_out, dact, nullptr, ws);
    } else {
        if (upsample) launch_wino_wgrad<2>(x, dy, dw, B, H, W, C1, Cout, Ct, tw, st, nullptr, 0, nullptr, ws);
        else launch_wino_wgrad<1>(x, dy, dw, B, H, W, C1, Cout, Ct, tw, st, nullptr, 0, nullptr, ws);
        if (C2) launch_wino_wgrad<1>(x2, dy, dw + C1, B, H, W, C2, Cout, Ct, tw, st, nullptr, 0, nullptr, ws);
    }
    return dvs::check_launch("dvs_conv3x3_wino_wgrad_gen");
}

int dvs_conv3x3_wino_wgrad_gen(const float* x, const float* x2, const float* dy, const float* y_out, float* dw, float* dbias, int B, int H, int W,
                               int C1, int C2, int Cout, int upsample, int dact, int target_workgroups, void* stream) {
    return dvs_conv3x3_wino_wgrad_gen_ws(x, x2, dy, y_out, dw, dbias, B, H, W, C1, C2, Cout, upsample, dact, target_workgroups, nullptr, 0,
                                         stream);
}

}  // extern "C"
