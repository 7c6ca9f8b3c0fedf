// a1-a3 in the opt-in bf16 mode (dvs_set_precision(1)): the stride-1, zero-pad-1 3x3 convolutions of the BasicBlocks -- forward and
// data gradient -- on v_mfma_f32_32x32x16_bf16.
//
// The implicit-GEMM kernels (conv_fwd.hip) gather every (tap, 32-channel) slice of the im2col matrix separately: nine times the
// loads and, with fp32 tensors in HBM, nine times the fp32 -> bf16 conversions.  At bf16 matrix rates (16x the fp32 rate) that
// gather, not the arithmetic, is the kernel.  Here a workgroup owns an 8 x 16 patch of output pixels of one image: the 10 x 18
// input patch (64 channels at a time) is read ONCE, rounded to bf16 and kept in LDS as [pixel][channel]; the A operand of every
// tap is the same image read at a shifted pixel -- one ds_read_b128 per 32 x 16 fragment at a per-lane base + an immediate (no
// address arithmetic in the loop; pixel rows are 144 bytes apart, so the eight lanes a b128 read serves together start on banks
// 0, 36, 8, 44, ... -- conflict-free).  The B operand (weights, pre-packed bf16 [tap][k / 16][n][16] by p16_pack_kernel, a few
// hundred KB that stay in L2) goes global -> registers, one coalesced 1 KB load per 32-column fragment and 16-k step.
//
//   A[row = pixel r of a 32-pixel m-tile (two patch rows)][k = 8 h + j]   B[k = 8 h + j][col = output channel]
//   C/D: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)          (the fp32 kernels' map)
//
// Tensors stay fp32 in HBM on both sides; accumulation is fp32; the BatchNorm statistics are taken from the fp32 results as in
// the fp32 kernels.  Data gradient: the same kernel over dY with the rotated / transposed pack (flip).
//
// In this file:  conv3x3_p16_kernel<TN, GEN>   the patch kernel; GEN = behind the decoder's gathers (ReflectionPad2d, nearest 2x
//                                              upsample, concat with the skip), + bias + ELU, and the full correlation of the
//                                              padded-domain data gradient                       (dvs_conv3x3_bf16_fwd / _gen)
//                conv3x3_p16_thin_kernel<CKT>  the 32- / 16-channel decoder levels: 32 output channels per workgroup, the chunk's
//                                              weights in LDS, fused activation derivative        (dvs_conv3x3_bf16_gen)
//                conv3x3_p16_wgrad_kernel<GEN> weight gradient, both operands through ds_read_b64_tr_b16
//                                                                                                 (dvs_conv3x3_bf16_wgrad / _wgrad_gen)
//                p16_pack_kernel               fp32 [Cout][3][3][Cin] -> bf16 [9][K / 16][N][16]  (dvs_conv3x3_bf16_pack)
// Replaces (in that mode) conv1 / conv2 of model/resnet_encoder.py's BasicBlocks (torchvision resnet18 layout) and the
// Conv3x3 / ConvBlock layers of model/layers.py:26-41,106-118 inside model/depth_decoder.py:52-62.
#include "conv_common.h"

#include <cstdint>
#include <cstdlib>

namespace {
using namespace dvsconv;

constexpr int PH = 8, PW = 16, IH = PH + 2, IW = PW + 2, NPIX = IH * IW;      // output patch, input patch (180 pixels)
constexpr int CK = 64, LDP = CK + 8;                                           // channels per chunk, LDS pixel stride (elements)
constexpr int NITEM = NPIX * (CK / 4), NLOAD = (NITEM + NT - 1) / NT;          // 16-byte staging items per chunk, per thread (12)
constexpr unsigned OOB = 0x80000000u;                                          // voffset past every descriptor: the load returns 0

struct P16Params {
    const float* x;          // [B][H][W][K] fp32
    const __bf16* w;         // [9][K / 16][N][16] bf16
    const float* res;        // optional [B][H][W][N]: added to the result (a skip path's gradient)
    float* y;                // [B][H][W][N]
    float* stats;            // optional [slots][G][2][N]
    int B, H, W, K, N;
    int tiles_x, tiles_y;    // patches per image
    int stat_split;          // images >= stat_split count into statistics group 1
    int stat_mask, stat_stride;
    // GEN (the decoder's layers): logical input cat(x [, x2]) of K = C1 + C2 channels at H x W; x is [B][H/2][W/2][C1] when `up`
    // (nearest 2x in the gather), x2 [B][H][W][C2]; reflect: ReflectionPad2d(1) instead of zeros; org = 1: y [B][H][W][N], org = 2: the
    // full correlation y [B][H+2][W+2][N]; + bias, activation (0 none, 1 ReLU, 2 ELU)
    const float* x2;
    const float* bias;
    int C1, up, reflect, org, Ho, Wo, act;
    // thin kernel only: x is a gradient dY and aux the forward output Y of the same shape -- the staged value is dY * act'(Y)
    // (dact: 1 ReLU, 2 ELU), i.e. the activation derivative of the layer whose data gradient this launch computes
    const float* aux;
    int dact;
};

// fp32 [N_w = Cout][3][3][Cin] -> bf16 [9][K / 16][N][16].  flip = 0: K = Cin, N = Cout, tap as stored; flip = 1 (data gradient):
// K = Cout, N = Cin, tap 8 - t (the filter rotated by 180 degrees).  One thread per 16-element output row.
// (N is padded to a multiple of 32 with zero columns: the thin layers' 16 output channels still fill a 32-column fragment.)
__global__ __launch_bounds__(256) void p16_pack_kernel(const float* __restrict__ w, __bf16* __restrict__ out, int Cout, int Cin, int flip) {
    const int K = flip ? Cout : Cin, Nr = flip ? Cin : Cout, N = (Nr + 31) / 32 * 32;
    const int rows = 9 * (K / 16) * N;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    const int n = i % N, k16 = (i / N) % (K / 16), t = i / (N * (K / 16));
    bf16x8 lo, hi;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = 16 * k16 + j;
        const float v = n >= Nr ? 0.f : flip ? w[((size_t)k * 9 + (8 - t)) * Cin + n] : w[((size_t)n * 9 + t) * Cin + k];
        if (j < 8) lo[j] = (__bf16)v;
        else hi[j - 8] = (__bf16)v;
    }
    bf16x8* o = reinterpret_cast<bf16x8*>(out + (size_t)i * 16);
    o[0] = lo;
    o[1] = hi;
}

// TN: 32-column fragments per wave; GEN: see P16Params; WIDE: wave layout.  false: 2 x 2 waves, a wave owns 64 pixels x 32 TN channels
// (the workgroup covers 64 TN output channels); true (TN = 1): 1 x 4 waves, a wave owns all 128 pixels x 32 channels (128 output
// channels per workgroup) -- every B fragment is then fetched by ONE wave instead of two, half the weight bytes from L2, which is what
// bounds the layers with many channel chunks
template <int TN, bool GEN = false, bool WIDE = false>
__global__ __launch_bounds__(NT) void conv3x3_p16_kernel(P16Params p) {
    constexpr int TM = WIDE ? 4 : 2, WNW = WIDE ? 4 : 2, BNW = 32 * WNW * TN;      // m-tiles per wave, waves along N, channels per workgroup
    static_assert(!WIDE || TN == 1, "the 1 x 4 layout has one fragment column per wave");
    __shared__ __attribute__((aligned(16))) __bf16 sP[2][NPIX * LDP];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WNW, wn = wave % WNW;
    const int H = p.H, W = p.W, K = p.K, N = p.N;
    const int nblk = N / BNW;
    const int bid = blockIdx.x, nb = bid % nblk, tile = bid / nblk;             // the channel blocks of a patch are neighbours (L2)
    const int tpi = p.tiles_x * p.tiles_y;
    const int b = tile / tpi, trem = tile - b * tpi, ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * PH, x0 = tx * PW, n0 = nb * BNW;
    const int Ho = GEN ? p.Ho : H, Wo = GEN ? p.Wo : W, org = GEN ? p.org : 1;
    const int C1 = GEN ? p.C1 : K, C2 = K - C1;
    const int Hs = (GEN && p.up) ? H >> 1 : H, Ws = (GEN && p.up) ? W >> 1 : W;             // geometry of source x

    // ---- staging items: item = tid + 256 j -> pixel item / 16 of the input patch, channels 4 (item % 16) .. of the chunk
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((size_t)p.B * Hs * Ws * C1 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((GEN && C2) ? p.x2 : p.x), 0,
                                                                          (GEN && C2) ? (int)((size_t)p.B * H * W * C2 * 4) : 0, 0x00020000);
    unsigned voff[NLOAD], voff2[GEN ? NLOAD : 1];
    const int c4 = tid & 15, pix0 = tid >> 4;
#pragma unroll
    for (int j = 0; j < NLOAD; ++j) {
        const int pix = pix0 + 16 * j;
        const int iy = pix / IW, ix = pix - iy * IW;
        int gy = y0 - org + iy, gx = x0 - org + ix;
        if (GEN && p.reflect) {                           // ReflectionPad2d(1): -1 -> 1, H -> H - 2 (H, W >= 2)
            gy = gy < 0 ? -gy : gy >= H ? 2 * H - 2 - gy : gy;
            gx = gx < 0 ? -gx : gx >= W ? 2 * W - 2 - gx : gx;
        }
        const bool ok = pix < NPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const int sy = (GEN && p.up) ? gy >> 1 : gy, sx = (GEN && p.up) ? gx >> 1 : gx;
        voff[j] = ok ? (unsigned)((((b * Hs + sy) * Ws + sx) * C1 + 4 * c4) * 4) : OOB;
        if constexpr (GEN) voff2[j] = ok ? (unsigned)((((b * H + gy) * W + gx) * C2 + 4 * c4) * 4) : OOB;
    }
    const int loff0 = pix0 * LDP + 4 * c4;                                     // LDS element of item j: loff0 + 16 j LDP
    f32x4 st[NLOAD];
    auto load_chunk = [&](int c) __attribute__((always_inline)) {
        if constexpr (GEN) {
            if (c * CK >= C1) {                           // (chunk-uniform: C1 % 64 == 0) this chunk lives in the skip tensor
#pragma unroll
                for (int j = 0; j < NLOAD; ++j)
                    st[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr2, voff2[j], (c * CK - C1) * 4, 0));
                return;
            }
        }
#pragma unroll
        for (int j = 0; j < NLOAD; ++j)
            st[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff[j], c * CK * 4, 0));
    };
    auto store_chunk = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < NLOAD; ++j)
            if (pix0 + 16 * j < NPIX) *reinterpret_cast<bf16x4*>(&sP[buf][loff0 + 16 * j * LDP]) = to_bf16(st[j]);
    };

    // ---- operands
    int a_base[TM];                                                             // my row of my m-tiles (two patch rows each)
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a_base[tm] = (((wm * TM + tm) * 2 + (r >> 4)) * IW + (r & 15)) * LDP + 8 * h;
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, (int)((size_t)9 * K * N * 2), 0x00020000);
    const unsigned b_voff = (unsigned)(((n0 + wn * TN * 32 + r) * 16 + 8 * h) * 2);
    const int tap_stride = K * N * 2, k16_stride = N * 32;                      // bytes of the pack per tap, per 16-k block
    auto load_b = [&](int chunk, int tap, int c16, bf16x8 (&bq)[TN]) __attribute__((always_inline)) {
        const int soff = tap * tap_stride + (chunk * (CK / 16) + c16) * k16_stride;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            bq[tn] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, b_voff, soff + tn * 1024, 0));
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[tm][tn][i] = 0.f;

    const int nchunk = K / CK;
    constexpr int STEPS = 9 * (CK / 16), DEPTH = 3;                             // 16-k steps per chunk; B operands fetched DEPTH steps ahead (6 and 10: same times)
    load_chunk(0);
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        store_chunk(c & 1);
        __syncthreads();                     // (the buffer written next time was last read two chunks ago: one barrier per chunk)
        if (c + 1 < nchunk) load_chunk(c + 1);
        const __bf16* P = sP[c & 1];
        bf16x8 bq[DEPTH + 1][TN];
#pragma unroll
        for (int s = 0; s < DEPTH; ++s) load_b(c, s / (CK / 16), s % (CK / 16), bq[s]);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const int tap = s / (CK / 16), c16 = s % (CK / 16);
#ifdef P16_DBG_NOB            // timing experiment (tools/build_variant.py --flag=-DP16_DBG_NOB): the B operands of the first steps only
            if (s + DEPTH < STEPS && s + DEPTH < 4) load_b(c, (s + DEPTH) / (CK / 16), (s + DEPTH) % (CK / 16), bq[(s + DEPTH) % (DEPTH + 1)]);
#else
            if (s + DEPTH < STEPS) load_b(c, (s + DEPTH) / (CK / 16), (s + DEPTH) % (CK / 16), bq[(s + DEPTH) % (DEPTH + 1)]);
#endif
            bf16x8 a[TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
                a[tm] = *reinterpret_cast<const bf16x8*>(P + a_base[tm] + ((tap / 3) * IW + (tap % 3)) * LDP + 16 * c16);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], bq[s % (DEPTH + 1)][tn], acc[tm][tn], 0, 0, 0);
        }
    }

    // ---- epilogue: [+ residual], store, statistics.  Register i of m-tile mt <-> patch pixel (2 mt + (m >> 4), m & 15), m = (i & 3) + 8 (i >> 2) + 4 h
    const bool want_stats = !GEN && p.stats != nullptr;
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.B * Ho * Wo * N * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res : p.y), 0, p.res ? (int)((size_t)p.B * Ho * Wo * N * 4) : 0, 0x00020000);
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int co = n0 + (wn * TN + tn) * 32 + r;
        const float bv = (GEN && p.bias) ? p.bias[co] : 0.f;
        float ssum = 0.f, ssq = 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int mt = wm * TM + tm;
            unsigned off[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = (i & 3) + 8 * (i >> 2) + 4 * h;
                const int oy = y0 + 2 * mt + (m >> 4), ox = x0 + (m & 15);
                off[i] = (oy < Ho && ox < Wo) ? (unsigned)((((b * Ho + oy) * Wo + ox) * N + co) * 4) : OOB;
            }
            float rv[16];
            if (p.res) {
#pragma unroll
                for (int i = 0; i < 16; ++i) rv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, off[i], 0, 0));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                float v = acc[tm][tn][i];
                if (want_stats && off[i] != OOB) {
                    ssum += v;
                    ssq += v * v;
                }
                if constexpr (GEN) {
                    v += bv;
                    if (p.act == ACT_ELU) v = v > 0.f ? v : expm1f(v);
                    else if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                }
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, p.res ? v + rv[i] : v), yr, off[i], 0, 0);
            }
        }
        if (want_stats) {
            // (copies of the table: thousands of workgroups adding to the same 2 N addresses serialise in the L2, see conv_wino.hip)
            float* stt = p.stats + (size_t)((int)blockIdx.x & p.stat_mask) * p.stat_stride + (b >= p.stat_split ? 2 * N : 0);
            const float a2 = ssum + __shfl_xor(ssum, 32, 64), q2 = ssq + __shfl_xor(ssq, 32, 64);
            if (h == 0) {
                atomicAdd(stt + co, a2);
                atomicAdd(stt + N + co, q2);
            }
        }
    }
}

// ---- thin layers (the decoder's 32- and 16-channel levels, both directions): 32 output channels per workgroup, channel chunks of 32
// or 16.  In fp32 these layers sit at 2 - 5 x their HBM bound because the fp32 matrix rate is the vector rate; on the bf16 cores
// the arithmetic is a tenth of the memory time.  Same patch scheme -- 8 x 32 output pixels (eight 32-pixel m-tiles, two per wave),
// 10 x 34 input pixels -- with the chunk's WEIGHTS in LDS too (9 taps x CKT x 32 columns = 18 KB at CKT = 32): with one 32-column
// fragment per wave a B operand from L2 would be re-fetched by every wave for every step.  One buffer, two barriers per chunk;
// three workgroups per CU hide each other's staging.
constexpr int TPH = 8, TPW = 32, TIH = TPH + 2, TIW = TPW + 2, TNPIX = TIH * TIW;          // 340 input pixels
template <int CKT>
__global__ __launch_bounds__(NT) void conv3x3_p16_thin_kernel(P16Params p) {
    constexpr int LDT = CKT + 8;                                                 // 80 / 48 bytes per pixel: conflict-free b128 reads
    constexpr int VPP = CKT / 4, PSTEP = NT / VPP;                               // 16-byte items per pixel, pixels per staging round
    constexpr int NLD = (TNPIX + PSTEP - 1) / PSTEP;                             // staging rounds (11 / 6)
    constexpr int WROWS = 9 * (CKT / 16) * 32, WLD = (WROWS * 2 + NT - 1) / NT;  // 32-byte weight rows of a chunk, 16-byte items per thread
    __shared__ __attribute__((aligned(16))) __bf16 sP[TNPIX * LDT];
    __shared__ __attribute__((aligned(16))) __bf16 sW[WROWS * 16];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, K = p.K, N = p.N, Np = (N + 31) / 32 * 32;
    const int nblk = Np / 32;
    const int bid = blockIdx.x, nb = bid % nblk, tile = bid / nblk;
    const int tpi = p.tiles_x * p.tiles_y;
    const int b = tile / tpi, trem = tile - b * tpi, ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * TPH, x0 = tx * TPW, n0 = nb * 32;
    const int Ho = p.Ho, Wo = p.Wo, org = p.org;
    const int C1 = p.C1, C2 = K - C1;
    const int Hs = p.up ? H >> 1 : H, Ws = p.up ? W >> 1 : W;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)((size_t)p.B * Hs * Ws * C1 * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t xr2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(C2 ? p.x2 : p.x), 0,
                                                                          C2 ? (int)((size_t)p.B * H * W * C2 * 4) : 0, 0x00020000);
    unsigned voff[NLD], voff2[NLD];
    const int c4 = tid % VPP, pix0 = tid / VPP;
#pragma unroll
    for (int j = 0; j < NLD; ++j) {
        const int pix = pix0 + PSTEP * j;
        const int iy = pix / TIW, ix = pix - iy * TIW;
        int gy = y0 - org + iy, gx = x0 - org + ix;
        if (p.reflect) {
            gy = gy < 0 ? -gy : gy >= H ? 2 * H - 2 - gy : gy;
            gx = gx < 0 ? -gx : gx >= W ? 2 * W - 2 - gx : gx;
        }
        const bool ok = pix < TNPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W;
        const int sy = p.up ? gy >> 1 : gy, sx = p.up ? gx >> 1 : gx;
        voff[j] = ok ? (unsigned)((((b * Hs + sy) * Ws + sx) * C1 + 4 * c4) * 4) : OOB;
        voff2[j] = ok ? (unsigned)((((b * H + gy) * W + gx) * C2 + 4 * c4) * 4) : OOB;
    }
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<__bf16*>(p.w), 0, (int)((size_t)9 * K * Np * 2), 0x00020000);

    int a_base[2];                                                               // m-tile = one patch row of 32 pixels
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) a_base[tm] = ((wave * 2 + tm) * TIW + r) * LDT + 8 * h;
    const int b_base = (r * 16 + 8 * h);                                         // my column's 8 k of a 16-k block of sW

    f32x16 acc[2];
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[tm][i] = 0.f;

    const int nchunk = K / CKT;
#pragma unroll 1
    for (int c = 0; c < nchunk; ++c) {
        f32x4 st[NLD];
        const bool second = c * CKT >= C1;                                       // (chunk-uniform: C1 % CKT == 0)
        if (second) {
#pragma unroll
            for (int j = 0; j < NLD; ++j)
                st[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr2, voff2[j], (c * CKT - C1) * 4, 0));
        } else {
#pragma unroll
            for (int j = 0; j < NLD; ++j) st[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, voff[j], c * CKT * 4, 0));
            if (p.dact) {                                                        // (no second source in this form)
                const __amdgpu_buffer_rsrc_t ar =
                    __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.aux), 0, (int)((size_t)p.B * Hs * Ws * C1 * 4), 0x00020000);
#pragma unroll
                for (int j = 0; j < NLD; ++j) {
                    const f32x4 yv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ar, voff[j], c * CKT * 4, 0));
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        st[j][e] = p.dact == ACT_ELU ? fmaf(st[j][e], fminf(yv[e], 0.f), st[j][e]) : (yv[e] > 0.f ? st[j][e] : 0.f);
                }
            }
        }
        // the chunk's weights: rows (tap, k16 of the chunk, column) of 32 bytes -> sW in the same order
        f32x4 wv[WLD];
#pragma unroll
        for (int j = 0; j < WLD; ++j) {
            const int it = tid + NT * j, row = it >> 1, half = it & 1;
            const int col = row & 31, k16 = (row >> 5) % (CKT / 16), tap = row / (32 * (CKT / 16));
            const unsigned off = it < WROWS * 2 ? (unsigned)((((tap * (K / 16) + c * (CKT / 16) + k16) * Np + n0 + col) * 16 + 8 * half) * 2) : OOB;
            wv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, off, 0, 0));
        }
        if (c > 0) __syncthreads();                                              // everyone is done reading the previous chunk
#pragma unroll
        for (int j = 0; j < NLD; ++j)
            if (pix0 + PSTEP * j < TNPIX) *reinterpret_cast<bf16x4*>(&sP[(pix0 + PSTEP * j) * LDT + 4 * c4]) = to_bf16(st[j]);
#pragma unroll
        for (int j = 0; j < WLD; ++j)
            if (tid + NT * j < WROWS * 2) *reinterpret_cast<f32x4*>(&sW[(tid + NT * j) * 8]) = wv[j];
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 9 * (CKT / 16); ++s) {
            const int tap = s / (CKT / 16), c16 = s % (CKT / 16);
            const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&sW[s * 32 * 16 + b_base]);
#pragma unroll
            for (int tm = 0; tm < 2; ++tm) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(&sP[a_base[tm] + ((tap / 3) * TIW + (tap % 3)) * LDT + 16 * c16]);
                acc[tm] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr, acc[tm], 0, 0, 0);
            }
        }
    }

    // ---- epilogue: + bias, activation, store.  Register i of m-tile mt <-> patch pixel (mt, (i & 3) + 8 (i >> 2) + 4 h)
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)((size_t)p.B * Ho * Wo * N * 4), 0x00020000);
    const int co = n0 + r;
    const bool co_ok = co < N;
    const float bv = (p.bias && co_ok) ? p.bias[co] : 0.f;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm) {
        const int oy = y0 + wave * 2 + tm;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int ox = x0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            const unsigned off = (co_ok && oy < Ho && ox < Wo) ? (unsigned)((((b * Ho + oy) * Wo + ox) * N + co) * 4) : OOB;
            float v = acc[tm][i] + bv;
            if (p.act == ACT_ELU) v = v > 0.f ? v : expm1f(v);
            else if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, off, 0, 0);
        }
    }
}

// ---- weight gradient:  dw[co][tap][ci] += sum over pixels dY[p][co] * x[p + tap][ci]  (K = pixels) ---------------------------------
// A workgroup owns a 32 x 32 block of (output channel, input channel) pairs for all nine taps and a range of 8 x 16 patches; the
// four waves take two patch rows each (a patch row = 16 pixels = one k-step) and keep nine 32 x 32 accumulators (144 registers).
// Both tiles sit in LDS as bf16 [pixel][32 channels] (64-byte pixel rows: the 4 pixels x 32 channels a 32-lane half reads
// transposed are 256 contiguous bytes, every bank once) and both operands are k-strided in that image, so they are fetched with
// ds_read_b64_tr_b16 (see conv_wgrad.hip): A = dY (rows = co) once per k-step, B = the x patch at the tap's shifted pixel, nine
// times.  At the end the waves add their accumulators through LDS and the sum goes into dw with float atomics (36 KB per
// workgroup: the reason the block is 32 x 32 -- a 64 x 64 block per workgroup means four times the atomic volume per launch).
constexpr int WB = 32;                                   // channels per block side
struct P16WgradParams {
    const float* x;          // [B][H][W][Cin]
    const float* dy;         // [B][H][W][Cout]
    float* dw;               // [Cout][3][3][Cin]
    int B, H, W, Cin, Cout;
    int tiles_x, tiles_y, npatch, nblk_ci, nblocks, nsplit, per_split;
    // GEN (the decoder's layers): the convolution's input is cat(x [, x2]) behind ReflectionPad2d(1) (reflect) / a nearest 2x upsample
    // of x (up: x is [B][H/2][W/2][C1]); C1 % 32 == 0.  dact != 0: dy is multiplied by act'(y_out) as it is staged (1 ReLU, 2 ELU) and
    // dbias (optional) += its column sums, taken by the workgroups of input-channel block 0.
    const float* x2;
    const float* y_out;
    float* dbias;
    int C1, up, reflect, dact;
};

using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
__device__ __forceinline__ bf16x8 tr_pair(const __bf16* lo, const __bf16* hi) {
    using lds_ptr = __attribute__((address_space(3))) s16x4*;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lo));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(hi));
    return __builtin_bit_cast(bf16x8, s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}

template <bool GEN>
__global__ __launch_bounds__(NT, 2) void conv3x3_p16_wgrad_kernel(P16WgradParams p) {      // (two waves per SIMD: 144 accumulators + <= 112)
    constexpr int NDY = PH * PW, DY_IT = NDY * (WB / 4) / NT, X_IT = (NPIX * (WB / 4) + NT - 1) / NT;      // 4 and 6 staging items per thread
    __shared__ __attribute__((aligned(16))) __bf16 sD[2][NDY * WB];
    __shared__ __attribute__((aligned(16))) __bf16 sX[2][NPIX * WB];
    __shared__ __attribute__((aligned(16))) float sR[4][16][64];                 // epilogue: one tap's tiles of the four waves
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = p.H, W = p.W, Cin = p.Cin, Cout = p.Cout;
    // the blocks of one patch range run on one XCD (they read the same pixels: L2)
    // (with fewer than eight ranges that rule would leave XCDs empty: plain order then -- workgroup i runs on XCD i % 8, so an XCD gets
    // every eighth block: a few input-channel slices and all output-channel slices)
    int block, split;
    if (p.nsplit >= 8) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        block = slot % p.nblocks;
        split = (slot / p.nblocks) * 8 + xcd;
    } else {
        block = blockIdx.x % p.nblocks;
        split = blockIdx.x / p.nblocks;
    }
    if (split >= p.nsplit) return;
    const int co0 = (block / p.nblk_ci) * WB, ci0 = (block % p.nblk_ci) * WB;
    const int t_begin = split * p.per_split, t_end = min(p.npatch, t_begin + p.per_split);

    // my input-channel block lives in ONE source: x (C1 channels, half resolution when `up`) or the skip x2 (Cin - C1 channels)
    const bool second = GEN && ci0 >= p.C1;
    const int Cs = GEN ? (second ? Cin - p.C1 : p.C1) : Cin, cs0 = second ? ci0 - p.C1 : ci0;
    const bool ups = GEN && p.up && !second;
    const int Hs = ups ? H >> 1 : H, Ws = ups ? W >> 1 : W;
    const __amdgpu_buffer_rsrc_t xr =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(second ? p.x2 : p.x), 0, (int)((size_t)p.B * Hs * Ws * Cs * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dy), 0, (int)((size_t)p.B * H * W * Cout * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>((GEN && p.dact) ? p.y_out : p.dy), 0,
                                                                         (GEN && p.dact) ? (int)((size_t)p.B * H * W * Cout * 4) : 0, 0x00020000);
    const int c4 = tid & 7, pix0 = tid >> 3;                                     // item j: pixel pix0 + 32 j, channels 4 c4 ..
    const int tpi = p.tiles_x * p.tiles_y;
    f32x4 sd[DY_IT], sx[X_IT], bsum = {0.f, 0.f, 0.f, 0.f};
    const bool do_bias = GEN && p.dbias != nullptr && ci0 == 0;
    auto load_patch = [&](int t) __attribute__((always_inline)) {
        const int b = t / tpi, trem = t - b * tpi, ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
        const int y0 = ty * PH, x0 = tx * PW;
#pragma unroll
        for (int j = 0; j < DY_IT; ++j) {
            const int pix = pix0 + 32 * j, gy = y0 + (pix >> 4), gx = x0 + (pix & 15);
            const unsigned off = (gy < H && gx < W && co0 + 4 * c4 < Cout) ? (unsigned)((((b * H + gy) * W + gx) * Cout + co0 + 4 * c4) * 4) : OOB;
            sd[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(dr, off, 0, 0));
            if (GEN && p.dact) {
                const f32x4 yv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ar, off, 0, 0));
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    sd[j][e] = p.dact == ACT_ELU ? fmaf(sd[j][e], fminf(yv[e], 0.f), sd[j][e]) : (yv[e] > 0.f ? sd[j][e] : 0.f);
            }
        }
#pragma unroll
        for (int j = 0; j < X_IT; ++j) {
            const int pix = pix0 + 32 * j, iy = pix / IW, ix = pix - iy * IW;
            int gy = y0 - 1 + iy, gx = x0 - 1 + ix;
            if (GEN && p.reflect) {                        // ReflectionPad2d(1): -1 -> 1, H -> H - 2
                gy = gy < 0 ? -gy : gy >= H ? 2 * H - 2 - gy : gy;
                gx = gx < 0 ? -gx : gx >= W ? 2 * W - 2 - gx : gx;
            }
            const bool ok = pix < NPIX && (unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W && cs0 + 4 * c4 < Cs;
            const int sy = ups ? gy >> 1 : gy, sxx = ups ? gx >> 1 : gx;
            const unsigned off = ok ? (unsigned)((((b * Hs + sy) * Ws + sxx) * Cs + cs0 + 4 * c4) * 4) : OOB;
            sx[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0));
        }
    };
    auto store_patch = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < DY_IT; ++j) {
            *reinterpret_cast<bf16x4*>(&sD[buf][(pix0 + 32 * j) * WB + 4 * c4]) = to_bf16(sd[j]);
            if (do_bias) bsum += sd[j];                  // (pixels outside the image were loaded as zeros)
        }
#pragma unroll
        for (int j = 0; j < X_IT; ++j)
            if (pix0 + 32 * j < NPIX) *reinterpret_cast<bf16x4*>(&sX[buf][(pix0 + 32 * j) * WB + 4 * c4]) = to_bf16(sx[j]);
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

    // my block of a transposed read (see conv_wgrad.hip): pixel (8 h' + q) of the 16-pixel k-step, channels 16 (g & 1) + 4 pp ..
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int trow = 8 * (g >> 1) + q, tcol = 16 * (g & 1) + 4 * pp;

    if (t_begin < t_end) load_patch(t_begin);
    int buf = 0;
#pragma unroll 1
    for (int t = t_begin; t < t_end; ++t, buf ^= 1) {
        store_patch(buf);
        __syncthreads();                     // (buffer buf ^ 1 was last read one patch ago, and every wave has passed this barrier since)
        if (t + 1 < t_end) load_patch(t + 1);
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
            const int py = 2 * wave + rr;                                        // my patch row = my k-step
            const __bf16* dbase = &sD[buf][(py * PW + trow) * WB + tcol];
            const bf16x8 a = tr_pair(dbase, dbase + 4 * WB);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const __bf16* xbase = &sX[buf][((py + tap / 3) * IW + (tap % 3) + trow) * WB + tcol];
                const bf16x8 bfr = tr_pair(xbase, xbase + 4 * WB);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bfr, acc[tap], 0, 0, 0);
            }
        }
    }

    // ---- the four waves' accumulators -> one sum per tap -> dw.  C/D map: column (ci) = lane & 31, row (co) = (i & 3) + 8 (i >> 2) + 4 (lane >> 5)
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) sR[wave][i][lane] = acc[tap][i];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = 4 * wave + k;                                          // wave w adds registers 4 w .. 4 w + 3 of the four waves
            const float v = sR[0][i][lane] + sR[1][i][lane] + sR[2][i][lane] + sR[3][i][lane];
            const int co = co0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            if (co < Cout && ci0 + r < Cin) atomicAdd(p.dw + ((size_t)co * 9 + tap) * Cin + ci0 + r, v);      // (16-channel layers: half a block)
        }
    }
    if (do_bias) {                                       // column sums of the staged dY: threads with equal c4 hold the same four channels
        __syncthreads();
        float* sB = &sR[0][0][0];
        if (tid < WB) sB[tid] = 0.f;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&sB[4 * c4 + e], bsum[e]);
        __syncthreads();
        if (tid < WB && co0 + tid < Cout) atomicAdd(p.dbias + co0 + tid, sB[tid]);
    }
}

}  // namespace

extern "C" {

int dvs_conv3x3_bf16_gen(const float* x, const float* x2, const void* wpack, const float* bias, float* y, int B, int H, int W, int C1, int C2,
                         int N, int Ho, int Wo, int org, int upsample, int reflect, int act, int as_dgrad, const float* y_out, int dact,
                         void* stream) {
    DVS_REQUIRE(x && wpack && y && B > 0 && H > 0 && W > 0 && C1 > 0 && C2 >= 0 && N > 0, "dvs_conv3x3_bf16_gen: bad argument");
    DVS_REQUIRE((C2 == 0) == (x2 == nullptr), "dvs_conv3x3_bf16_gen: x2 and C2 go together");
    const bool big = C1 % CK == 0 && C2 % CK == 0 && N % 64 == 0;
    DVS_REQUIRE(big || (C1 % 16 == 0 && C2 % 16 == 0 && N % 16 == 0),
                "dvs_conv3x3_bf16_gen: C1, C2 and N must be multiples of 16 (got %d, %d, %d)", C1, C2, N);
    DVS_REQUIRE((org == 1 && Ho == H && Wo == W) || (org == 2 && Ho == H + 2 && Wo == W + 2 && !reflect),
                "dvs_conv3x3_bf16_gen: org 1 (same size) or 2 (full correlation, zero padding)");
    DVS_REQUIRE(!reflect || (H >= 2 && W >= 2), "dvs_conv3x3_bf16_gen: ReflectionPad2d(1) needs H, W >= 2");
    DVS_REQUIRE(!upsample || ((H & 1) == 0 && (W & 1) == 0), "dvs_conv3x3_bf16_gen: upsampled input has even H, W");
    DVS_REQUIRE(act == 0 || act == ACT_RELU || act == ACT_ELU, "dvs_conv3x3_bf16_gen: activation %d (0, 1 = ReLU, 2 = ELU)", act);
    const int K = C1 + C2;
    DVS_REQUIRE((double)B * Ho * Wo * (K > N ? K : N) * 4 < 2147483648.0, "dvs_conv3x3_bf16_gen: tensors must be smaller than 2 GiB");
    DVS_REQUIRE(dact == 0 || ((dact == ACT_RELU || dact == ACT_ELU) && y_out && !big && C2 == 0 && !upsample),
                "dvs_conv3x3_bf16_gen: the fused activation derivative (dact, y_out) exists in the thin kernel, one plain source");
    P16Params p{};
    p.x = x; p.x2 = x2; p.w = static_cast<const __bf16*>(wpack); p.bias = bias; p.y = y;
    p.B = B; p.H = H; p.W = W; p.K = K; p.N = N; p.C1 = C1; p.up = upsample; p.reflect = reflect; p.org = org; p.Ho = Ho; p.Wo = Wo; p.act = act;
    p.aux = y_out; p.dact = dact;
    p.stat_split = B;
    dvs::ProfScope prof(as_dgrad ? dvs::SLOT_CONV_DGRAD : dvs::SLOT_CONV_FWD, (hipStream_t)stream);
    prof.work(2.0 * B * H * W * (double)N * 9.0 * K);
    if (!big) {          // thin layers: 32 output channels per workgroup, chunks of 32 (or 16) channels, 8 x 32 patches
        p.tiles_x = (Wo + TPW - 1) / TPW; p.tiles_y = (Ho + TPH - 1) / TPH;
        const int grid = B * p.tiles_x * p.tiles_y * ((N + 31) / 32);
        if (C1 % 32 == 0 && C2 % 32 == 0) hipLaunchKernelGGL((conv3x3_p16_thin_kernel<32>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, p);
        else hipLaunchKernelGGL((conv3x3_p16_thin_kernel<16>), dim3(grid), dim3(NT), 0, (hipStream_t)stream, p);
        return dvs::check_launch("dvs_conv3x3_bf16_gen");
    }
    p.tiles_x = (Wo + PW - 1) / PW; p.tiles_y = (Ho + PH - 1) / PH;
    const int tiles = B * p.tiles_x * p.tiles_y;
    static const bool wide = [] { const char* e = getenv("DVS_BF16_WIDE"); return !(e && e[0] == '0'); }();
    if (N % 128 == 0 && wide) hipLaunchKernelGGL((conv3x3_p16_kernel<1, true, true>), dim3(tiles * (N / 128)), dim3(NT), 0, (hipStream_t)stream, p);
    else if (N % 128 == 0) hipLaunchKernelGGL((conv3x3_p16_kernel<2, true>), dim3(tiles * (N / 128)), dim3(NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv3x3_p16_kernel<1, true>), dim3(tiles * (N / 64)), dim3(NT), 0, (hipStream_t)stream, p);
    return dvs::check_launch("dvs_conv3x3_bf16_gen");
}

int dvs_conv3x3_bf16_wgrad(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Cout, int target_workgroups, void* stream) {
    DVS_REQUIRE(x && dy && dw && B > 0 && H > 0 && W > 0, "dvs_conv3x3_bf16_wgrad: bad argument");
    DVS_REQUIRE(Cin % WB == 0 && Cout % WB == 0 && Cin > 0 && Cout > 0, "dvs_conv3x3_bf16_wgrad: channel counts must be multiples of 32 (got %d, %d)", Cin, Cout);
    DVS_REQUIRE((double)B * H * W * (Cin > Cout ? Cin : Cout) * 4 < 2147483648.0, "dvs_conv3x3_bf16_wgrad: tensors must be smaller than 2 GiB");
    P16WgradParams p{};
    p.x = x; p.dy = dy; p.dw = dw; p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
    p.tiles_x = (W + PW - 1) / PW; p.tiles_y = (H + PH - 1) / PH;
    p.npatch = B * p.tiles_x * p.tiles_y;
    p.nblk_ci = Cin / WB;
    p.nblocks = (Cout / WB) * p.nblk_ci;
    const int target = target_workgroups > 0 ? target_workgroups : 512;
    int nsplit = (target + p.nblocks - 1) / p.nblocks;
    nsplit = nsplit < 1 ? 1 : (nsplit > p.npatch ? p.npatch : nsplit);
    p.per_split = (p.npatch + nsplit - 1) / nsplit;
    p.nsplit = (p.npatch + p.per_split - 1) / p.per_split;
    const int grid = p.nsplit >= 8 ? p.nblocks * ((p.nsplit + 7) / 8) * 8 : p.nblocks * p.nsplit;
    dvs::ProfScope prof(dvs::SLOT_CONV_WGRAD, (hipStream_t)stream);
    prof.work(2.0 * B * H * W * (double)Cout * 9.0 * Cin);
    hipLaunchKernelGGL(conv3x3_p16_wgrad_kernel<false>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, p);
    return dvs::check_launch("dvs_conv3x3_bf16_wgrad");
}

int dvs_conv3x3_bf16_wgrad_gen(const float* x, const float* x2, const float* dy, const float* y_out, float* dw, float* dbias, int B, int H, int W,
                               int C1, int C2, int Cout, int upsample, int reflect, int dact, int target_workgroups, void* stream) {
    DVS_REQUIRE(x && dy && dw && B > 0 && H > 0 && W > 0 && C1 > 0 && C2 >= 0, "dvs_conv3x3_bf16_wgrad_gen: bad argument");
    DVS_REQUIRE((C2 == 0) == (x2 == nullptr), "dvs_conv3x3_bf16_wgrad_gen: x2 and C2 go together");
    DVS_REQUIRE(C1 % 16 == 0 && C2 % 16 == 0 && Cout % 16 == 0 && Cout > 0 && (C2 == 0 || C1 % WB == 0),
                "dvs_conv3x3_bf16_wgrad_gen: channel counts must be multiples of 16, C1 of 32 when there is a second source (got %d, %d, %d)", C1, C2, Cout);
    DVS_REQUIRE(!reflect || (H >= 2 && W >= 2), "dvs_conv3x3_bf16_wgrad_gen: ReflectionPad2d(1) needs H, W >= 2");
    DVS_REQUIRE(!upsample || ((H & 1) == 0 && (W & 1) == 0), "dvs_conv3x3_bf16_wgrad_gen: upsampled input has even H, W");
    DVS_REQUIRE(dact == 0 || ((dact == ACT_RELU || dact == ACT_ELU) && y_out), "dvs_conv3x3_bf16_wgrad_gen: dact 1 (ReLU) / 2 (ELU) needs the forward output");
    DVS_REQUIRE(dbias == nullptr || dact != 0, "dvs_conv3x3_bf16_wgrad_gen: the bias gradient rides on the activation-derivative path");
    const int Cin = C1 + C2;
    DVS_REQUIRE((double)B * H * W * (Cin > Cout ? Cin : Cout) * 4 < 2147483648.0, "dvs_conv3x3_bf16_wgrad_gen: tensors must be smaller than 2 GiB");
    P16WgradParams p{};
    p.x = x; p.x2 = x2; p.dy = dy; p.y_out = y_out; p.dw = dw; p.dbias = dbias;
    p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.C1 = C1; p.up = upsample; p.reflect = reflect; p.dact = dact;
    p.tiles_x = (W + PW - 1) / PW; p.tiles_y = (H + PH - 1) / PH;
    p.npatch = B * p.tiles_x * p.tiles_y;
    p.nblk_ci = (Cin + WB - 1) / WB;
    p.nblocks = ((Cout + WB - 1) / WB) * p.nblk_ci;
    const int target = target_workgroups > 0 ? target_workgroups : 512;
    int nsplit = (target + p.nblocks - 1) / p.nblocks;
    nsplit = nsplit < 1 ? 1 : (nsplit > p.npatch ? p.npatch : nsplit);
    p.per_split = (p.npatch + nsplit - 1) / nsplit;
    p.nsplit = (p.npatch + p.per_split - 1) / p.per_split;
    const int grid = p.nsplit >= 8 ? p.nblocks * ((p.nsplit + 7) / 8) * 8 : p.nblocks * p.nsplit;
    dvs::ProfScope prof(dvs::SLOT_CONV_WGRAD, (hipStream_t)stream);
    prof.work(2.0 * B * H * W * (double)Cout * 9.0 * Cin);
    hipLaunchKernelGGL(conv3x3_p16_wgrad_kernel<true>, dim3(grid), dim3(NT), 0, (hipStream_t)stream, p);
    return dvs::check_launch("dvs_conv3x3_bf16_wgrad_gen");
}

int dvs_conv3x3_bf16_pack(const float* w, void* out, int Cout, int Cin, int flip, void* stream) {
    DVS_REQUIRE(w && out && Cout > 0 && Cin > 0, "dvs_conv3x3_bf16_pack: bad argument");
    DVS_REQUIRE(Cout % 16 == 0 && Cin % 16 == 0, "dvs_conv3x3_bf16_pack: channel counts must be multiples of 16 (got %d, %d)", Cout, Cin);
    const int K = flip ? Cout : Cin, N = ((flip ? Cin : Cout) + 31) / 32 * 32;
    const int rows = 9 * (K / 16) * N;
    hipLaunchKernelGGL(p16_pack_kernel, dim3((rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, static_cast<__bf16*>(out), Cout, Cin, flip);
    return dvs::check_launch("dvs_conv3x3_bf16_pack");
}

int dvs_conv3x3_bf16_fwd(const float* x, const void* wpack, const float* res, float* y, float* stats, int stat_groups, int stat_slots, int B,
                         int H, int W, int K, int N, int as_dgrad, void* stream) {
    DVS_REQUIRE(x && wpack && y && B > 0 && H > 0 && W > 0, "dvs_conv3x3_bf16_fwd: bad argument");
    DVS_REQUIRE(K % CK == 0 && N % 64 == 0 && K > 0 && N > 0, "dvs_conv3x3_bf16_fwd: K %% 64 == 0 and N %% 64 == 0 (got %d, %d)", K, N);
    DVS_REQUIRE(stat_slots >= 1 && stat_slots <= 64 && (stat_slots & (stat_slots - 1)) == 0, "dvs_conv3x3_bf16_fwd: stat_slots must be a power of two <= 64");
    DVS_REQUIRE(stat_groups >= 0 && stat_groups <= 2 && (stat_groups != 2 || (B & 1) == 0), "dvs_conv3x3_bf16_fwd: stat_groups");
    DVS_REQUIRE((double)B * H * W * (K > N ? K : N) * 4 < 2147483648.0, "dvs_conv3x3_bf16_fwd: tensors must be smaller than 2 GiB");
    P16Params p{};
    p.x = x; p.w = static_cast<const __bf16*>(wpack); p.res = res; p.y = y;
    p.stats = stat_groups ? stats : nullptr;
    p.B = B; p.H = H; p.W = W; p.K = K; p.N = N;
    p.tiles_x = (W + PW - 1) / PW; p.tiles_y = (H + PH - 1) / PH;
    p.C1 = K; p.org = 1; p.Ho = H; p.Wo = W;
    p.stat_split = stat_groups == 2 ? B / 2 : B;
    p.stat_mask = stat_slots - 1;
    p.stat_stride = stat_groups * 2 * N;
    const int tiles = B * p.tiles_x * p.tiles_y;
    dvs::ProfScope prof(as_dgrad ? dvs::SLOT_CONV_DGRAD : dvs::SLOT_CONV_FWD, (hipStream_t)stream);
    prof.work(2.0 * B * H * W * (double)N * 9.0 * K);
    static const bool wide = [] { const char* e = getenv("DVS_BF16_WIDE"); return !(e && e[0] == '0'); }();
    if (N % 128 == 0 && wide) hipLaunchKernelGGL((conv3x3_p16_kernel<1, false, true>), dim3(tiles * (N / 128)), dim3(NT), 0, (hipStream_t)stream, p);
    else if (N % 128 == 0) hipLaunchKernelGGL((conv3x3_p16_kernel<2>), dim3(tiles * (N / 128)), dim3(NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv3x3_p16_kernel<1>), dim3(tiles * (N / 64)), dim3(NT), 0, (hipStream_t)stream, p);
    return dvs::check_launch("dvs_conv3x3_bf16_fwd");
}

}  // extern "C"
