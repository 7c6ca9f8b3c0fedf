// a1-a3: forward convolution as an implicit GEMM on the fp32 matrix cores of gfx950.
//
//   Y[m][n] = act( sum_k A[m][k] * Wt[n][k] + bias[n] ),   m = (b,oy,ox), n = co, k = (kh,kw,ci)
//
// A is never materialised: the workgroup gathers its BM x 32 slice of the im2col matrix straight from
// the NHWC activation (16-byte vectors of 4 channels; zero or reflection padding, the decoder's
// nearest-upsample + skip concat, a folded BatchNorm+ReLU of the producer, or the encoder's input
// normalisation are applied in that gather), stages it and the [BN][32] weight slice in LDS (double
// buffered, one barrier per K-step) and runs v_mfma_f32_32x32x2_f32 on 32x32 sub-tiles.
// Epilogue: bias, ReLU/ELU/sigmoid, optional per-channel sum / sum-of-squares of the raw output (the
// batch statistics nn.BatchNorm2d needs in training mode), NHWC store (128-B segments per half wave).
//
// Replaces nn.Conv2d / Conv3x3 / ConvBlock forward of model/resnet_encoder.py:100-111 (torchvision
// BasicBlock convs), model/depthnet.py:64-90, model/layers.py:106-136, model/posenet_single.py:174-202.
#include "conv_common.h"

#include <cstdlib>
#include <type_traits>

namespace {
using namespace dvsconv;

struct FwdParams {
    const float* x;
    const float* w;      // [Cout][Ktot]
    const float* bias;   // [Cout] or NULL
    float* y;            // [B,Ho,Wo,Cout]
    float* stats;        // [2][Cout] running sum / sum of squares of the raw output, or NULL
    int stat_split;      // rows >= stat_split are counted into a second set stats + 2*Cout (0x7fffffff: one set)
    float* y2;           // data gradient of an upsample+concat conv: channels >= split_c1 go here ([B,H,W,Cout-C1]) ...
    int split_c1;        // ... channels < split_c1 are summed 2x2 into the coarse tensor y ([B,H/2,W/2,C1]) with atomics
    int dbg_nobarrier;   // timing experiment only (DVS_CONV_DEBUG_NOBARRIER=1): skip the K-loop barriers -> wrong results
    const float* zero_page;   // 16 bytes of zeros: what the LDS-DMA kernel fetches for padding / tail lanes
    const float* res;         // [B,Ho,Wo,Cout] added before the activation (inference BasicBlock tail), or NULL
    int stat_mask, stat_stride;   // statistics copies: output tile t adds into stats + (t & stat_mask) * stat_stride (0, 0: one table)
    int work_m;               // rows to count as algorithmic work in the profile (0 = all M rows): the padded-domain data
                              // gradient computes a border of rows that the unpadded problem does not have
    int ksplit;               // > 1: split-K launch of the LDS-DMA kernel -- raw partial sums are added into a zero-filled y
                              // with atomics, splitk_finish_kernel applies bias / residual / activation afterwards
    Grid3 g;             // logical grid: M tiles, N tiles, parity classes (launched 1-D, see xcd_logical)
    ConvShape s;
    InXform t;
    int act;
};

// Epilogue shared by the register-staged and the LDS-DMA kernels: bias, activation, BatchNorm statistics,
// the data-gradient row mappings (stride-2 parity classes, upsample+concat split) and the NHWC store.
// Row -> output pixel for the data-gradient modes (stride-2 parity classes, upsample+concat split).  The mapping
// needs two integer divisions per ROW; done per output ELEMENT in the epilogue loop it was ~4500 VALU instructions
// per wave -- more than the whole K loop of the thin decoder layers -- so one thread per row computes it once into
// LDS (the tiles are free after the last stage) and the epilogue reads it back.
template <int BM>
__device__ __forceinline__ void fill_row_table(const FwdParams& p, const ConvShape& s, int* tab, int m0, int M, int Hr, int Wr,
                                               int rstep, int oy0, int ox0) {
    const int r = threadIdx.x;
    if (r < BM) {
        const int m = min(m0 + r, M - 1);
        int pix = m, cp = 0;
        if (rstep == 2) {                                // parity-class row -> pixel of the full grid
            const int b = m / (Hr * Wr), rem = m - b * (Hr * Wr);
            const int oy = rem / Wr, ox = rem - oy * Wr;
            pix = (b * s.Ho + (oy * 2 + oy0)) * s.Wo + (ox * 2 + ox0);
        }
        if (p.split_c1 > 0) {                            // pixel of the half-resolution tensor that receives the 2x2 sum
            const int b = m / (s.Ho * s.Wo), rem = m - b * (s.Ho * s.Wo);
            const int yy = rem / s.Wo, xx = rem - yy * s.Wo;
            cp = (b * (s.Ho >> 1) + (yy >> 1)) * (s.Wo >> 1) + (xx >> 1);
        }
        tab[r] = pix;
        tab[BM + r] = cp;
    }
    __syncthreads();
}

// Epilogue body, specialised at compile time on what a launch needs (the fp32 MFMA shares the vector ALU, so the
// ~10 VALU instructions per output element that a do-everything epilogue spends on statistics selects, activation
// dispatch and 64-bit index arithmetic it does not need are paid in matrix throughput):
//   STATS 0: none   1: whole tile in one statistics group   2: tile straddles the group boundary (per-element test)
//   ACT   false: plain store (no bias, no activation)       true: + bias, runtime activation
template <int TM, int TN, int MODE, int STATS, bool ACT, bool RES = false>
__device__ __forceinline__ void conv_epilogue_body(const FwdParams& p, const ConvShape& s, f32x16 (&acc)[TM][TN], int m0,
                                                   int n0, int wm, int wn, int lane, int M, const int* rowtab, int bm) {
    // C/D map of the 32x32 MFMA: n = lane & 31, m = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int ln = lane & 31, lh = lane >> 5;
    const bool tile_second = m0 >= p.stat_split;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + (wn * TN + tn) * 32 + ln;
        const bool n_ok = n < s.Cout;
        const float bv = (ACT && p.bias && n_ok) ? p.bias[n] : 0.f;
        float ssum = 0.f, ssq = 0.f, ssum1 = 0.f, ssq1 = 0.f;      // second pair: rows >= p.stat_split
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int mb = m0 + (wm * TM + tm) * 32 + 4 * lh;
            // residual values of this 32 x 32 block: sixteen unconditional loads from clamped (always valid) addresses, all in
            // flight before the first use -- inside the bounds test below each load would wait for the one before it
            float rres[RES ? 16 : 1];
            if constexpr (RES) {
                const int nc = min(n, s.Cout - 1);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int m = mb + (i & 3) + 8 * (i >> 2);
                    const size_t pix = (MODE == IN_DGRAD && rowtab) ? (size_t)rowtab[m - m0] : (size_t)min(m, M - 1);
                    rres[i] = p.res[pix * s.Cout + nc];
                }
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int m = mb + (i & 3) + 8 * (i >> 2);
                const float v = acc[tm][tn][i];
                if (m < M && n_ok) {
                    if (STATS == 1) {
                        ssum += v;
                        ssq += v * v;
                    } else if (STATS == 2) {
                        const bool second = m >= p.stat_split;
                        ssum += second ? 0.f : v;
                        ssq += second ? 0.f : v * v;
                        ssum1 += second ? v : 0.f;
                        ssq1 += second ? v * v : 0.f;
                    }
                    if (MODE == IN_DGRAD && p.split_c1 > 0) {
                        // gradient of cat([upsample2x(a), skip]): a gets the 2x2 sum, skip its own channels
                        if (n < p.split_c1) {
                            // rows m (even) and m + 1 are horizontal neighbours of one 2x2 block (Wo, the tile origin and
                            // the register pairing i, i+1 are all even): add them in registers, one atomic per pair
                            if ((i & 1) == 0) {
                                const float pair = (m + 1 < M) ? acc[tm][tn][i + 1] : 0.f;
                                const size_t cp = (size_t)rowtab[bm + m - m0];
                                atomicAdd(p.y + cp * p.split_c1 + n, v + pair);
                            }
                        } else {
                            const size_t pix = rowtab ? (size_t)rowtab[m - m0] : (size_t)m;
                            p.y2[pix * (s.Cout - p.split_c1) + (n - p.split_c1)] = v;
                        }
                    } else {
                        const size_t pix = (MODE == IN_DGRAD && rowtab) ? (size_t)rowtab[m - m0] : (size_t)m;
                        const float vr = RES ? v + rres[i] : v;
                        p.y[pix * s.Cout + n] = ACT ? apply_act(vr + bv, p.act) : vr;
                    }
                }
            }
        }
        if (STATS) {
            float* st = p.stats + (size_t)((m0 / bm) & p.stat_mask) * p.stat_stride;
            if (STATS == 1 && tile_second) {
                ssum1 = ssum;
                ssq1 = ssq;
                ssum = ssq = 0.f;
            }
            if (STATS == 2 || !tile_second) {
                ssum += __shfl_xor(ssum, 32, 64);
                ssq += __shfl_xor(ssq, 32, 64);
                if (lh == 0 && n_ok) {
                    atomicAdd(st + n, ssum);
                    atomicAdd(st + s.Cout + n, ssq);
                }
            }
            if (STATS == 2 || tile_second) {
                ssum1 += __shfl_xor(ssum1, 32, 64);
                ssq1 += __shfl_xor(ssq1, 32, 64);
                if (lh == 0 && n_ok) {
                    atomicAdd(st + 2 * s.Cout + n, ssum1);
                    atomicAdd(st + 3 * s.Cout + n, ssq1);
                }
            }
        }
    }
}

template <int TM, int TN, int MODE>
__device__ __forceinline__ void conv_epilogue(const FwdParams& p, const ConvShape& s, f32x16 (&acc)[TM][TN], int m0, int n0,
                                              int wm, int wn, int lane, int M, int Hr, int Wr, int rstep, int oy0, int ox0,
                                              const int* rowtab, int bm) {
    (void)Hr; (void)Wr; (void)rstep; (void)oy0; (void)ox0;
    const bool act = p.bias != nullptr || p.act != ACT_NONE;          // workgroup-uniform dispatch
    int stats = 0;
    if (MODE != IN_DGRAD && p.stats) {
        const int bm_rows = TM * 32 * (wm + 1);                        // rows of the tile at or below this wave
        (void)bm_rows;
        stats = (m0 < p.stat_split && m0 + bm > p.stat_split) ? 2 : 1;
    }
    if (MODE == IN_DGRAD && p.ksplit <= 1) {
        // p.res: another gradient of the same tensor (a skip / downsample path's), added here instead of by an autograd pass
        if (p.res) conv_epilogue_body<TM, TN, MODE, 0, false, true>(p, s, acc, m0, n0, wm, wn, lane, M, rowtab, bm);
        else conv_epilogue_body<TM, TN, MODE, 0, false>(p, s, acc, m0, n0, wm, wn, lane, M, rowtab, bm);
    } else if (p.ksplit > 1) {       // (data gradients only split when their rows are plain pixels: stride 1, no upsample split)
        const int ln = lane & 31, lh = lane >> 5;
#pragma unroll
        for (int tn = 0; tn < TN; ++tn) {
            const int n = n0 + (wn * TN + tn) * 32 + ln;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int mb = m0 + (wm * TM + tm) * 32 + 4 * lh;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int m = mb + (i & 3) + 8 * (i >> 2);
                    if (m < M && n < s.Cout) atomicAdd(p.y + (size_t)m * s.Cout + n, acc[tm][tn][i]);
                }
            }
        }
    } else if (stats == 0) {
        if (p.res) conv_epilogue_body<TM, TN, MODE, 0, true, true>(p, s, acc, m0, n0, wm, wn, lane, M, rowtab, bm);
        else if (act) conv_epilogue_body<TM, TN, MODE, 0, true>(p, s, acc, m0, n0, wm, wn, lane, M, rowtab, bm);
        else conv_epilogue_body<TM, TN, MODE, 0, false>(p, s, acc, m0, n0, wm, wn, lane, M, rowtab, bm);
    } else if (stats == 1) {
        if (act) conv_epilogue_body<TM, TN, MODE, 1, true>(p, s, acc, m0, n0, wm, wn, lane, M, rowtab, bm);
        else conv_epilogue_body<TM, TN, MODE, 1, false>(p, s, acc, m0, n0, wm, wn, lane, M, rowtab, bm);
    } else {
        conv_epilogue_body<TM, TN, MODE, 2, true>(p, s, acc, m0, n0, wm, wn, lane, M, rowtab, bm);
    }
}

#include "conv_dma.h"

// NBUF = 2: double-buffered LDS, one barrier per K-step (2 workgroups / CU for the 128-wide tiles);
// NBUF = 1: one LDS buffer, the next stage waits in registers, two barriers per K-step but half the LDS,
//           so twice as many workgroups (waves per SIMD) hide each other's gather / barrier phases.
// BF16: the same gathers, the operands rounded to bf16 in store_stage, tiles of LDKH bf16 per row, v_mfma_f32_32x32x16_bf16.
template <int BM, int BN, int WM, int WN, int MODE, bool FOLD, int NBUF, bool BF16 = false>
__global__ __launch_bounds__(NT) void conv_fwd_kernel(FwdParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_VECS = BM / 32, B_VECS = BN / 32;       // 16-byte vectors per thread per stage
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                               // [NBUF][BM][LDK]
    float* Bs = smem + NBUF * BM * LDK;             // [NBUF][BN][LDK]
    __bf16* Ah = reinterpret_cast<__bf16*>(smem);   // BF16: [NBUF][BM][LDKH], [NBUF][BN][LDKH]
    __bf16* Bh = Ah + NBUF * BM * LDKH;

    ConvShape s = p.s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // N tile fastest, then parity class, then M tile: the tiles that share an im2col slice are neighbours
    // N tile fastest, then M tile: the tiles that share an im2col slice are neighbours on one XCD (xcd_logical).
    // The four parity classes of a stride-2 data gradient carry 1, 2, 2 and 4 taps: dealt round-robin in launch order
    // (M tile fastest, class slowest, no remap) they balance across the XCDs; contiguous ranges measured 1.6x slower.
    int bid_x, bid_y, bid_z;
    if (p.g.z > 1) {
        int lg = blockIdx.x;
        bid_x = lg % p.g.x;
        lg /= p.g.x;
        bid_y = lg % p.g.y;
        bid_z = lg / p.g.y;
    } else {
        int lg = xcd_logical(blockIdx.x, p.g.x * p.g.y, p.g.remap);
        bid_y = lg % p.g.y;
        bid_x = lg / p.g.y;
        bid_z = 0;
    }
    const int m0 = bid_x * BM, n0 = bid_y * BN;
    const int c4 = (tid & 7) * 4, r0 = tid >> 3;            // my k-offset inside a stage, my first row

    // Row space.  Forward: output pixels.  Data gradient: input pixels -- for a stride-2 conv they are split
    // into the 4 parity classes (blockIdx.z) of (y + pad, x + pad): a class only ever meets the taps of its
    // own parity (ky = py + 2 jy), so the empty 3/4 of the strided im2col matrix is never multiplied.
    int Hr = s.Ho, Wr = s.Wo, rstep = 1, oy0 = 0, ox0 = 0, ky0 = 0, kx0 = 0, kw_full = s.kw;
    if (MODE == IN_DGRAD && s.stride == 2) {
        const int py = bid_z >> 1, px = bid_z & 1;
        oy0 = (py - s.pad) & 1;
        ox0 = (px - s.pad) & 1;
        Hr = (s.Ho - oy0 + 1) >> 1;
        Wr = (s.Wo - ox0 + 1) >> 1;
        rstep = 2;
        ky0 = py;
        kx0 = px;
        s.kh = py < s.kh ? (s.kh - py + 1) >> 1 : 0;          // taps of this class
        s.kw = px < s.kw ? (s.kw - px + 1) >> 1 : 0;
        s.Ktot = s.kh * s.kw * s.Cin;
    }
    const int M = s.B * Hr * Wr;
    if (m0 >= M) return;

    // pixels of my A rows
    int a_b[A_VECS], a_iy[A_VECS], a_ix[A_VECS];
    bool a_ok[A_VECS];
#pragma unroll
    for (int j = 0; j < A_VECS; ++j) {
        int m = m0 + r0 + 32 * j;
        a_ok[j] = m < M;
        m = min(m, M - 1);
        int b = m / (Hr * Wr), rem = m - b * (Hr * Wr);
        int oy = rem / Wr, ox = rem - oy * Wr;
        a_b[j] = b;
        if (MODE == IN_DGRAD) {      // the gather subtracts the tap from (y + pad, x + pad)
            a_iy[j] = oy * rstep + oy0 + s.pad;
            a_ix[j] = ox * rstep + ox0 + s.pad;
        } else {
            a_iy[j] = oy * s.stride - s.pad;
            a_ix[j] = ox * s.stride - s.pad;
        }
    }
    // weight rows
    const float* b_ptr[B_VECS];
    bool b_ok[B_VECS];
#pragma unroll
    for (int j = 0; j < B_VECS; ++j) {
        int n = n0 + r0 + 32 * j;
        b_ok[j] = n < s.Cout;
        b_ptr[j] = p.w + (size_t)min(n, s.Cout - 1) * p.s.Ktot;
    }

    KPos kp;
    kp.init(c4, s);
    // stage registers: raw loads + validity; zeros / folds are applied in store_stage, AFTER the MFMAs
    f32x4 ra[A_VECS], rb[B_VECS], ry[A_VECS], re[A_VECS], fsc, fsh;
    bool ra_ok[A_VECS], rb_ok;
    unsigned ra_mask[A_VECS];
    float psc = 1.f, psh = 0.f;
    int t_off[A_VECS], t_off2[A_VECS], cur_tap = -1;       // per-tap cache of pixel offsets / validity
    bool t_ok[A_VECS];
    auto load_stage = [&]() {      // issues the loads of the stage kp points at, then advances kp
        const bool k_ok = kp.k < s.Ktot;
        int kc = min(kp.k, p.s.Ktot - 4);
        const int ky = ky0 + rstep * kp.ky, kx = kx0 + rstep * kp.kx;       // actual tap (parity classes skip taps)
        if (MODE == IN_DGRAD) kc = k_ok ? (ky * kw_full + kx) * s.Cin + kp.ci : 0;
        // per-tap work (padding, clamping, pixel offsets) only when my tap changes: every Cin/32 stages
        const int tap = ky * kw_full + kx;
        if (MODE != IN_PLANAR && tap != cur_tap) {
            cur_tap = tap;
#pragma unroll
            for (int j = 0; j < A_VECS; ++j) {
                bool ok = a_ok[j];
                if (MODE == IN_DGRAD) dgrad_tap_setup(p.s, a_b[j], a_iy[j], a_ix[j], ky, kx, ok, t_off[j]);
                else tap_setup<MODE>(s, p.t, a_b[j], a_iy[j] + kp.ky, a_ix[j] + kp.kx, ok, t_off[j], t_off2[j]);
                t_ok[j] = ok;
                if (p.dbg_nobarrier & 2) t_off[j] = t_off2[j] = 0;      // experiment: every row gathers pixel 0 (cache-resident)
            }
        }
#pragma unroll
        for (int j = 0; j < A_VECS; ++j) {
            if (MODE == IN_PLANAR) {
                int ci;
                ra[j] = gather_planar_raw(p.x, s, a_b[j], a_iy[j], a_ix[j], kc, a_ok[j] && k_ok, ra_mask[j], ci);
                if (FOLD && j == 0) {
                    psc = p.t.in_scale[ci];
                    psh = p.t.in_shift[ci];
                }
            } else if (MODE == IN_DGRAD) {
                ra[j] = *reinterpret_cast<const f32x4*>(p.x + (t_off[j] + kp.ci));
                if (p.t.dact) ry[j] = *reinterpret_cast<const f32x4*>(p.t.aux + (t_off[j] + kp.ci));
                re[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (p.s.pad_mode == PAD_REFLECT)
                    re[j] = dgrad_reflect_extra(p.x, p.s, p.t, a_b[j], a_iy[j], a_ix[j], ky, kx, kp.ci, a_ok[j] && k_ok);
                ra_ok[j] = t_ok[j] && k_ok;
            } else {
                ra[j] = load_tap4<MODE>(p.x, p.t, t_off[j], t_off2[j], kp.ci);
                ra_ok[j] = t_ok[j] && k_ok;
            }
        }
        if (FOLD && MODE != IN_PLANAR) {
            fsc = *reinterpret_cast<const f32x4*>(p.t.in_scale + kp.ci);
            fsh = *reinterpret_cast<const f32x4*>(p.t.in_shift + kp.ci);
        }
#pragma unroll
        for (int j = 0; j < B_VECS; ++j) rb[j] = *reinterpret_cast<const f32x4*>(b_ptr[j] + kc);
        rb_ok = k_ok;
        if (MODE == IN_PLANAR) kp.k += BK;
        else kp.advance(s);
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < A_VECS; ++j) {
            f32x4 v;
            if (MODE == IN_PLANAR) v = finalize_planar<FOLD>(ra[j], ra_mask[j], psc, psh);
            else if (MODE == IN_DGRAD) v = finalize_dgrad(ra[j], ry[j], re[j], ra_ok[j], p.t.dact);
            else v = finalize<FOLD>(ra[j], ra_ok[j], fsc, fsh, p.t.in_relu);
            if constexpr (BF16) *reinterpret_cast<bf16x4*>(Ah + (buf * BM + r0 + 32 * j) * LDKH + c4) = to_bf16(v);
            else *reinterpret_cast<f32x4*>(As + (buf * BM + r0 + 32 * j) * LDK + c4) = v;
        }
#pragma unroll
        for (int j = 0; j < B_VECS; ++j) {
            const bool ok = b_ok[j] && rb_ok;
            f32x4 v = rb[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
            if constexpr (BF16) *reinterpret_cast<bf16x4*>(Bh + (buf * BN + r0 + 32 * j) * LDKH + c4) = to_bf16(v);
            else *reinterpret_cast<f32x4*>(Bs + (buf * BN + r0 + 32 * j) * LDK + c4) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    const int KT = (s.Ktot + BK - 1) / BK;
    load_stage();
    store_stage(0);
    __syncthreads();
#pragma unroll 1
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = (NBUF == 2) ? (kt & 1) : 0;
        const bool staged = !(p.dbg_nobarrier & 4);   // experiment bit 4: MFMA + LDS reads only
        if (staged && kt + 1 < KT) load_stage();      // global loads in flight during the MFMAs
        if constexpr (BF16) mfma_stage_bf16<TM, TN>(Ah + buf * BM * LDKH, Bh + buf * BN * LDKH, wm * TM * 32, wn * TN * 32, lane, acc);
        else mfma_stage<TM, TN>(As + buf * BM * LDK, Bs + buf * BN * LDK, wm * TM * 32, wn * TN * 32, lane, acc);
        if (NBUF == 1 && !(p.dbg_nobarrier & 1)) __syncthreads();   // everyone is done reading the single buffer
        if (staged && kt + 1 < KT) store_stage((NBUF == 2) ? (buf ^ 1) : 0);
        if (!(p.dbg_nobarrier & 1)) __syncthreads();
    }

    int* rowtab = nullptr;
    if (MODE == IN_DGRAD && (rstep == 2 || p.split_c1 > 0)) {
        rowtab = reinterpret_cast<int*>(smem);
        fill_row_table<BM>(p, s, rowtab, m0, M, Hr, Wr, rstep, oy0, ox0);
    }
    conv_epilogue<TM, TN, MODE>(p, s, acc, m0, n0, wm, wn, lane, M, Hr, Wr, rstep, oy0, ox0, rowtab, BM);
}

int conv_nbuf() {
    static int v = [] {
        const char* e = getenv("DVS_CONV_NBUF");
        return (e && e[0] == '2') ? 2 : 1;
    }();
    return v;
}

template <int BM, int BN, int WM, int WN, int MODE, bool FOLD, int NBUF, bool BF16 = false>
void launch_buf(const FwdParams& p, hipStream_t st, int slot);

// LDS-DMA kernel (conv_dma.h) whenever the gather needs no per-element arithmetic and a 32-k stage stays
// inside one tap; DVS_CONV_DMA=0 forces the register-staged kernel.
template <int MODE, bool FOLD>
bool dma_eligible(const FwdParams& p) {
    static const bool enabled = [] { const char* e = getenv("DVS_CONV_DMA"); return !(e && e[0] == '0'); }();
    if (!enabled || FOLD || MODE == IN_PLANAR || (p.s.Cin % BK) != 0 || p.s.kh > 3 || p.s.kw > 3) return false;
    // buffer-descriptor addressing with an out-of-range sentinel: every operand below 2 GiB
    const double lim = 2147483648.0 / 4;
    if ((double)p.s.B * p.s.H * p.s.W * p.s.Cin >= lim || (double)p.s.Cout * p.s.Ktot >= lim) return false;
    if (MODE == IN_DGRAD) return p.t.dact == 0 && p.s.pad_mode == PAD_ZERO && p.split_c1 == 0;
    return true;
}

inline int xcd_remap_enabled() {
    static const int on = [] { const char* e = getenv("DVS_CONV_XCD"); return !(e && e[0] == '0') ? 1 : 0; }();
    return on;
}

// y = act(y + bias + residual) over [M][Cout], 4 channels per thread: second half of a split-K forward
__global__ __launch_bounds__(256) void splitk_finish_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                            const float* __restrict__ res, size_t nvec, int cout4, int act) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= nvec) return;
    f32x4 v = reinterpret_cast<f32x4*>(y)[i];
    if (bias) v += reinterpret_cast<const f32x4*>(bias)[i % cout4];
    if (res) v += reinterpret_cast<const f32x4*>(res)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = apply_act(v[e], act);
    reinterpret_cast<f32x4*>(y)[i] = v;
}

// Zero fill of a split-K output.  A kernel, not hipMemsetAsync: inside a captured HIP graph the memset node was not
// reliably ordered before the kernel node that accumulates into the buffer (replays of a graph captured after another
// one lost partial sums, tools/dbg_graph.py); kernel -> kernel edges are.
__global__ __launch_bounds__(256) void splitk_zero_kernel(float* __restrict__ y, size_t nvec) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < nvec) reinterpret_cast<f32x4*>(y)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int BM, int BN, int WM, int WN, int MODE>
void launch_dma(const FwdParams& p, hipStream_t st, int slot) {
    int M = p.s.B * p.s.Ho * p.s.Wo;
    dim3 grid((M + BM - 1) / BM, (p.s.Cout + BN - 1) / BN);
    if (MODE == IN_DGRAD && p.s.stride == 2) {
        int mc = p.s.B * ((p.s.Ho + 1) / 2) * ((p.s.Wo + 1) / 2);
        grid = dim3((mc + BM - 1) / BM, (p.s.Cout + BN - 1) / BN, 4);
    }
    size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(float);
    auto kern = conv_dma_kernel<BM, BN, WM, WN, MODE>;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024 - 256) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dvs::ProfScope prof(slot, st);
    const double eff = (MODE == IN_DGRAD) ? 1.0 / (p.s.stride * p.s.stride) : 1.0;
    prof.work(2.0 * (p.work_m ? p.work_m : M) * p.s.Cout * (double)p.s.Ktot * eff);
    FwdParams q = p;
    q.g = Grid3{(int)grid.x, (int)grid.y, (int)grid.z, xcd_remap_enabled()};
    q.ksplit = 1;
    if (MODE != IN_DGRAD || (p.s.stride == 1 && p.split_c1 == 0)) {
        // Few output pixels (batch-1 inference: 300 ... 4800 rows in layers 2-4 and the coarse decoder levels) leave most
        // of the 256 CUs without a tile while each tile walks a K of thousands: split the channel blocks over several
        // workgroups.  Also taken by small training launches without a statistics epilogue (decoder level 4, batch-4 data
        // gradients of layer 4).
        static const bool splitk = [] { const char* e = getenv("DVS_CONV_SPLITK"); return !(e && e[0] == '0'); }();
        const int tiles = grid.x * grid.y, nC = p.s.Cin / BK;
        // (200 ... 320 tiles: one workgroup per CU and a K of thousands -- upconv_4_0's data gradient, 284 tiles of 72 stages: 148 us;
        // split towards 1 024 workgroups: 111 us.  DVS_SPLITK_TILES / DVS_SPLITK_TARGET for experiments)
        static const int sk_tiles = [] { const char* e = getenv("DVS_SPLITK_TILES"); return e ? atoi(e) : 320; }();
        static const int sk_target = [] { const char* e = getenv("DVS_SPLITK_TARGET"); return e ? atoi(e) : 0; }();
        if (splitk && !dvs::deterministic() && !p.stats && tiles < sk_tiles && nC >= 2 && (p.s.Cout & 3) == 0) {
            const int target = sk_target > 0 ? sk_target : (tiles < 200 ? 256 : 1024);
            int want = min(nC, (target + tiles - 1) / tiles);
            const int per = (nC + want - 1) / want;
            q.ksplit = (nC + per - 1) / per;
        }
    }
    static const float* zp = [] {
        void* d = nullptr;
        (void)hipGetSymbolAddress(&d, HIP_SYMBOL(g_dvs_zero_page));
        return static_cast<const float*>(d);
    }();
    q.zero_page = zp;
    if (q.ksplit > 1) {
        const size_t nvec = (size_t)M * p.s.Cout / 4;
        hipLaunchKernelGGL(splitk_zero_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st, p.y, nvec);
        hipLaunchKernelGGL(kern, dim3(grid.x * grid.y * q.ksplit), dim3(NT), lds, st, q);
        if (p.bias || p.res || p.act != ACT_NONE)
            hipLaunchKernelGGL(splitk_finish_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, st, p.y, p.bias, p.res,
                               nvec, p.s.Cout / 4, p.act);
        return;
    }
    hipLaunchKernelGGL(kern, dim3(grid.x * grid.y * grid.z), dim3(NT), lds, st, q);
}

template <int BM, int BN, int WM, int WN, int MODE, bool FOLD>
void launch_cfg(const FwdParams& p0, hipStream_t st, int slot) {
    FwdParams p = p0;
    if constexpr (MODE == IN_PLANAR) {
        if (dvs::precision_bf16()) {            // the stems in the bf16 mode: the generic planar gather beside the bf16 matrix pipe
            p.dbg_nobarrier = 0;
            launch_buf<BM, BN, WM, WN, MODE, FOLD, 2, true>(p, st, slot);
            return;
        }
    }
    if constexpr (!FOLD && MODE != IN_PLANAR) {
        // dvs_set_precision(1): the register-staged kernel with bf16 tiles (the LDS-DMA path cannot convert on the way)
        if (dvs::precision_bf16() && (p.s.Cin & 3) == 0) {
            p.dbg_nobarrier = 0;
            launch_buf<BM, BN, WM, WN, MODE, FOLD, 2, true>(p, st, slot);
            return;
        }
        if (dma_eligible<MODE, FOLD>(p)) {
            launch_dma<BM, BN, WM, WN, MODE>(p, st, slot);
            return;
        }
    }
    static const int nobar = dvs::experiment_flags("DVS_CONV_DEBUG_NOBARRIER");
    p.dbg_nobarrier = nobar;
    if (conv_nbuf() == 2) launch_buf<BM, BN, WM, WN, MODE, FOLD, 2>(p, st, slot);
    else launch_buf<BM, BN, WM, WN, MODE, FOLD, 1>(p, st, slot);
}

template <int BM, int BN, int WM, int WN, int MODE, bool FOLD, int NBUF, bool BF16>
void launch_buf(const FwdParams& p, hipStream_t st, int slot) {
    int M = p.s.B * p.s.Ho * p.s.Wo;
    dim3 grid((M + BM - 1) / BM, (p.s.Cout + BN - 1) / BN);
    if (MODE == IN_DGRAD && p.s.stride == 2) {      // 4 parity classes of input pixels (see the kernel)
        int mc = p.s.B * ((p.s.Ho + 1) / 2) * ((p.s.Wo + 1) / 2);
        grid = dim3((mc + BM - 1) / BM, (p.s.Cout + BN - 1) / BN, 4);
    }
    size_t lds = BF16 ? (size_t)NBUF * (BM + BN) * LDKH * 2 : (size_t)NBUF * (BM + BN) * LDK * sizeof(float);
    if (BF16 && MODE == IN_DGRAD && lds < (size_t)2 * BM * sizeof(int)) lds = (size_t)2 * BM * sizeof(int);      // the row table of the epilogue
    auto kern = conv_fwd_kernel<BM, BN, WM, WN, MODE, FOLD, NBUF, BF16>;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    dvs::ProfScope prof(slot, st);
    // algorithmic flops of the convolution (IN_PLANAR: the 7 real taps of a row, not the 8 padded ones)
    const double k_real = (MODE == IN_PLANAR) ? (double)p.s.Cin * p.s.kh * p.s.kw : (double)p.s.Ktot;
    const double eff = (MODE == IN_DGRAD) ? 1.0 / (p.s.stride * p.s.stride) : 1.0;   // strided taps are empty
    prof.work(2.0 * (p.work_m ? p.work_m : M) * p.s.Cout * k_real * eff);
    FwdParams q = p;
    q.g = Grid3{(int)grid.x, (int)grid.y, (int)grid.z, xcd_remap_enabled()};
    hipLaunchKernelGGL(kern, dim3(grid.x * grid.y * grid.z), dim3(NT), lds, st, q);
}

template <int MODE, bool FOLD>
void launch_mode(const FwdParams& p, hipStream_t st, int slot) {
    const ConvShape& s = p.s;
    const int M = s.B * s.Ho * s.Wo;
    // DVS_CONV_TILE=small: 64-row tiles everywhere (more, lighter workgroups; tuning experiments)
    static const bool small = [] { const char* e = getenv("DVS_CONV_TILE"); return e && e[0] == 's'; }();
    if constexpr (MODE == IN_DGRAD && !FOLD) {
        // upconv_1_1's data gradient (N = 96 input channels): three 32-column tiles per wave instead of a 128-wide tile
        // that is one quarter padding
        static const bool n96 = [] { const char* e = getenv("DVS_CONV_N96"); return !(e && e[0] == '0'); }();
        if (n96 && s.Cout == 96 && (dvs::precision_bf16() || !dma_eligible<MODE, FOLD>(p))) {
            FwdParams q = p;
            q.dbg_nobarrier = 0;
            if (dvs::precision_bf16()) launch_buf<128, 96, 4, 1, MODE, FOLD, 2, true>(q, st, slot);
            else if (conv_nbuf() == 2) launch_buf<128, 96, 4, 1, MODE, FOLD, 2>(q, st, slot);
            else launch_buf<128, 96, 4, 1, MODE, FOLD, 1>(q, st, slot);
            return;
        }
    }
    if (s.Cout > 64) {
        // few output pixels (layer3/4, pose decoder): halve the M tile so the grid still covers the 256 CUs
        // ... and once more (64 x 64 tiles) when even that leaves half the chip without a tile and the launch cannot split K
        // (statistics epilogue): layers 3 / 4 of a batch-4 step
        static const bool small_m = [] { const char* e = getenv("DVS_CONV_SMALLM"); return !(e && e[0] == '0'); }();
        const int t64 = ((M + 63) / 64) * ((s.Cout + 127) / 128);
        // 128 x 128 tiles run two workgroups per CU (512 at a time): when the last of at most three rounds is mostly empty (the ViT
        // token GEMMs: 1 032 tiles = 2.02 rounds -> 3), 64-row tiles waste half as much (fc1 62 -> 92 TF, qkv 72 -> 92 TF)
        const int t128 = ((M + 127) / 128) * ((s.Cout + 127) / 128);
        const double rounds = t128 / 512.0, full = ceil(rounds);
        const bool tail_waste = rounds <= 3.2 && (full - rounds) / full > 0.2;
        // ... and for plain 1x1 products (token GEMMs of a single ViT frame: 1 370 rows) that leave CUs without a 64 x 128 tile
        const bool gemm_small = s.kh == 1 && s.kw == 1 && MODE != IN_DGRAD && t64 < 224;
        if (small_m && MODE != IN_DGRAD && ((p.stats && t64 < 160) || gemm_small)) launch_cfg<64, 64, 2, 2, MODE, FOLD>(p, st, slot);
        else if (small || t128 < 448 || tail_waste) launch_cfg<64, 128, 1, 4, MODE, FOLD>(p, st, slot);
        else launch_cfg<128, 128, 2, 2, MODE, FOLD>(p, st, slot);
    } else if (s.Cout > 32) {
        if (small) launch_cfg<64, 64, 2, 2, MODE, FOLD>(p, st, slot);
        else launch_cfg<128, 64, 2, 2, MODE, FOLD>(p, st, slot);
    } else {
        launch_cfg<128, 32, 4, 1, MODE, FOLD>(p, st, slot);
    }
}

// Wt[ci][tap][co] = W[co][tap][ci]: the [N][K] operand of the data-gradient GEMM (32x32 LDS tile transpose).
__global__ __launch_bounds__(256) void pack_wt_kernel(const float* __restrict__ w, float* __restrict__ wt, int Cout,
                                                      int Cin, int T) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z, ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int co = co0 + ty + 8 * j, ci = ci0 + tx;
        tile[ty + 8 * j][tx] = (co < Cout && ci < Cin) ? w[((size_t)co * T + tap) * Cin + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ci = ci0 + ty + 8 * j, co = co0 + tx;
        if (ci < Cin && co < Cout) wt[((size_t)ci * T + tap) * Cout + co] = tile[tx][ty + 8 * j];
    }
}

// dZ = dY * act'(Y): the pre-activation gradient as a tensor, for the layers whose data- and weight-gradient kernels
// would otherwise both re-derive it in their gathers (9 taps x N tiles times per element in the data gradient).
// dbias (optional): += column sums of dZ -- the bias gradient, so that the weight-gradient kernel needs no bias path.
// A thread's 4-channel vector is the same in every grid-stride iteration (C/4 divides 256 when dbias is given).
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      float* __restrict__ dz, size_t nvec, int act, float* __restrict__ dbias,
                                                      int C) {
    __shared__ f32x4 red[256];
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += stride) {
        f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
        const f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
        if (act == ACT_ELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] = fmaf(g[e], fminf(v[e], 0.f), g[e]);      // 1 + min(y, 0)
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) g[e] *= act_grad_from_out(v[e], act);
        }
        reinterpret_cast<f32x4*>(dz)[i] = g;
        bsum += g;
    }
    if (dbias) {
        const int cv = C / 4;
        red[threadIdx.x] = bsum;
        __syncthreads();
        if ((int)threadIdx.x < cv) {
            for (int k = threadIdx.x + cv; k < 256; k += cv) bsum += red[k];
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(dbias + threadIdx.x * 4 + e, bsum[e]);
        }
    }
}

// Reflection fold + upsample / concat split of a data gradient computed on the PADDED input domain.
// G [B][H+2][W+2][C] = gradient w.r.t. ReflectionPad2d(1)(input) (a plain zero-padded "full" correlation, which the
// LDS-DMA kernel runs at full speed).  The mirrored border folds back: padded row 0 -> row 1, row H+1 -> row H-2, same
// for columns.  One thread per (output pixel or 2x2 block, 4-channel vector):
//   C1 == 0: dx [B][H][W][C] = fold(G);
//   C1 > 0 : channels < C1 are summed over each 2x2 block into dx [B][H/2][W/2][C1] (gradient of the nearest upsample),
//            channels >= C1 go to dx_skip [B][H][W][C-C1].
__device__ __forceinline__ f32x4 folded_at(const float* __restrict__ G, int b, int y, int x, int c, int H, int W, int C) {
    const int Wp = W + 2;
    const float* base = G + ((size_t)b * (H + 2)) * Wp * C + c;
    auto at = [&](int py, int px) { return *reinterpret_cast<const f32x4*>(base + ((size_t)py * Wp + px) * C); };
    const int ey = (y == 1) ? 0 : ((y == H - 2) ? H + 1 : -1), ex = (x == 1) ? 0 : ((x == W - 2) ? W + 1 : -1);
    f32x4 v = at(y + 1, x + 1);
    if (ey >= 0) v += at(ey, x + 1);
    if (ex >= 0) v += at(y + 1, ex);
    if (ey >= 0 && ex >= 0) v += at(ey, ex);
    if (H == 3 && y == 1) v += at(H + 1, x + 1) + ((ex >= 0) ? at(H + 1, ex) : f32x4{0.f, 0.f, 0.f, 0.f});   // row 1 == row H-2
    if (W == 3 && x == 1) v += at(y + 1, W + 1) + ((ey >= 0) ? at(ey, W + 1) : f32x4{0.f, 0.f, 0.f, 0.f});
    if (H == 3 && W == 3 && y == 1 && x == 1) v += at(H + 1, W + 1);
    return v;
}
__global__ __launch_bounds__(256) void reflect_fold_kernel(const float* __restrict__ G, float* __restrict__ dx,
                                                           float* __restrict__ dx_skip, int B, int H, int W, int C, int C1) {
    const int cv = C / 4;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (C1 == 0) {
        if (i >= (size_t)B * H * W * cv) return;
        const int c = (int)(i % cv) * 4;
        size_t m = i / cv;
        const int x = (int)(m % W);
        m /= W;
        const int y = (int)(m % H), b = (int)(m / H);
        *reinterpret_cast<f32x4*>(dx + (((size_t)b * H + y) * W + x) * C + c) = folded_at(G, b, y, x, c, H, W, C);
        return;
    }
    const int H2 = H >> 1, W2 = W >> 1;
    if (i >= (size_t)B * H2 * W2 * cv) return;
    const int c = (int)(i % cv) * 4;
    size_t m = i / cv;
    const int x2 = (int)(m % W2);
    m /= W2;
    const int y2 = (int)(m % H2), b = (int)(m / H2);
    f32x4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = folded_at(G, b, 2 * y2 + (q >> 1), 2 * x2 + (q & 1), c, H, W, C);
    if (c < C1) {
        *reinterpret_cast<f32x4*>(dx + (((size_t)b * H2 + y2) * W2 + x2) * C1 + c) = (v[0] + v[1]) + (v[2] + v[3]);
    } else {
        const int cs = C - C1;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            *reinterpret_cast<f32x4*>(dx_skip + (((size_t)b * H + 2 * y2 + (q >> 1)) * W + 2 * x2 + (q & 1)) * cs + (c - C1)) = v[q];
    }
}

// Every data-gradient weight pack of a network in ONE launch (the weights only change at the optimiser step, so
// dp.FusedAdam repacks them all right after it instead of 51 small launches inside the backward pass).
struct PackEntry {
    const float* w;
    float* wt;
    int Cout, Cin, T, wg_begin;      // wg_begin: first workgroup of this entry (entries sorted, cumulative)
};
__global__ __launch_bounds__(256) void pack_wt_batch_kernel(const PackEntry* __restrict__ tab, int n) {
    __shared__ float tile[32][33];
    __shared__ int s_e;
    if (threadIdx.x == 0) {
        int e = 0;
        while (e + 1 < n && (int)blockIdx.x >= tab[e + 1].wg_begin) ++e;
        s_e = e;
    }
    __syncthreads();
    const PackEntry en = tab[s_e];
    const int tci = (en.Cin + 31) / 32, tco = (en.Cout + 31) / 32;
    int local = blockIdx.x - en.wg_begin;
    const int tap = local / (tci * tco);
    local -= tap * tci * tco;
    const int co0 = (local / tci) * 32, ci0 = (local % tci) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int co = co0 + ty + 8 * j, ci = ci0 + tx;
        tile[ty + 8 * j][tx] = (co < en.Cout && ci < en.Cin) ? en.w[((size_t)co * en.T + tap) * en.Cin + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        int ci = ci0 + ty + 8 * j, co = co0 + tx;
        if (ci < en.Cin && co < en.Cout) en.wt[((size_t)ci * en.T + tap) * en.Cout + co] = tile[tx][ty + 8 * j];
    }
}

}  // namespace

extern "C" {

int dvs_reflect_fold(const float* g_padded, float* dx, float* dx_skip, int B, int H, int W, int C, int C1, void* stream) {
    DVS_REQUIRE(g_padded && dx && B > 0 && H >= 3 && W >= 3 && C > 0 && (C & 3) == 0, "dvs_reflect_fold: bad argument");
    DVS_REQUIRE(C1 == 0 || ((C1 & 3) == 0 && C1 <= C && !((H | W) & 1) && (C1 == C || dx_skip)), "dvs_reflect_fold: bad split");
    const size_t n = (C1 == 0 ? (size_t)B * H * W : (size_t)B * (H / 2) * (W / 2)) * (C / 4);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dvs::ProfScope prof(dvs::SLOT_CONV_DGRAD, st);       // part of the data gradient's time (no flops of its own)
    hipLaunchKernelGGL(reflect_fold_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g_padded, dx, dx_skip, B, H, W, C, C1);
    return dvs::check_launch("dvs_reflect_fold");
}

int dvs_act_bwd(const float* dy, const float* y, float* dz, size_t n, int act, float* dbias, int C, void* stream) {
    DVS_REQUIRE(dy && y && dz && n > 0 && (n & 3) == 0 && act >= 0 && act <= 3, "dvs_act_bwd: bad argument");
    DVS_REQUIRE(!dbias || (C >= 4 && (C & 3) == 0 && C / 4 <= 256 && 256 % (C / 4) == 0 && n % C == 0),
                "dvs_act_bwd: the bias gradient needs C/4 to divide 256 (C=%d)", C);
    const size_t nvec = n / 4;
    size_t blocks = (nvec + 255) / 256;
    if (blocks > 512) blocks = 512;                      // grid-stride: few same-address atomics for the bias gradient
    if (dbias) {
        // every workgroup ends with one atomic per channel: 512 of them on a small tensor (PoseNet's decoder: 7 MB) serialise
        // for ~45 us behind 6 us of streaming -- at least 32 grid-stride iterations per workgroup there
        size_t few = nvec / (256 * 32);
        few = few < 32 ? 32 : few;
        if (blocks > few) blocks = few;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    dvs::ProfScope prof(dvs::SLOT_CONV_DGRAD, st);       // counted with the data gradient (it serves both gradients)
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, st, dy, y, dz, nvec, act, dbias, C);
    return dvs::check_launch("dvs_act_bwd");
}

int dvs_conv2d_pack_wt_batch(const void* table, int n_entries, int total_workgroups, void* stream) {
    DVS_REQUIRE(table && n_entries > 0 && total_workgroups > 0, "dvs_conv2d_pack_wt_batch: bad argument");
    hipLaunchKernelGGL(pack_wt_batch_kernel, dim3(total_workgroups), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const PackEntry*>(table), n_entries);
    return dvs::check_launch("dvs_conv2d_pack_wt_batch");
}

int dvs_conv2d_pack_wt(const float* w, float* wt, int Cout, int Cin, int kh, int kw, void* stream) {
    DVS_REQUIRE(w && wt && Cout > 0 && Cin > 0 && kh > 0 && kw > 0, "dvs_conv2d_pack_wt: bad argument");
    dim3 grid((Cin + 31) / 32, (Cout + 31) / 32, kh * kw);
    hipLaunchKernelGGL(pack_wt_kernel, grid, dim3(256), 0, static_cast<hipStream_t>(stream), w, wt, Cout, Cin, kh * kw);
    return dvs::check_launch("dvs_conv2d_pack_wt");
}

int dvs_conv2d_dgrad(const float* dy, const float* wt, float* dx, const dvs_conv_desc* d, const float* y_out,
                     int dact, float* dx_skip, int C1, void* stream) {
    return dvs_conv2d_dgrad_res(dy, wt, dx, d, y_out, dact, dx_skip, C1, nullptr, stream);
}

int dvs_conv2d_dgrad_res(const float* dy, const float* wt, float* dx, const dvs_conv_desc* d, const float* y_out,
                         int dact, float* dx_skip, int C1, const float* residual, void* stream) {
    DVS_REQUIRE(dy && wt && dx && d, "dvs_conv2d_dgrad: null pointer");
    DVS_REQUIRE(!residual || (C1 == 0 && residual != dx), "dvs_conv2d_dgrad: a residual excludes the upsample+concat split and may not alias dx");
    DVS_REQUIRE(C1 == 0 || (d->stride == 1 && (d->H & 1) == 0 && (d->W & 1) == 0 && C1 <= d->Cin &&
                            (C1 == d->Cin || dx_skip != nullptr)),
                "dvs_conv2d_dgrad: bad upsample+concat split");
    DVS_REQUIRE(d->stride == 1 || d->stride == 2, "dvs_conv2d_dgrad: stride %d (1 or 2 supported)", d->stride);
    DVS_REQUIRE((d->Cout & 3) == 0 && (d->Cin & 3) == 0, "dvs_conv2d_dgrad: channel counts must be multiples of 4");
    DVS_REQUIRE(d->pad_mode == PAD_ZERO || (d->pad_mode == 2 && d->pad == 0 && d->stride == 1 && d->H > 2 && d->W > 2 && !dact && !C1) ||
                    (d->pad_mode == PAD_REFLECT && d->pad == 1 && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->H >= 2 && d->W >= 2),
                "dvs_conv2d_dgrad: reflect mode is ReflectionPad2d(1) + 3x3 stride 1 only; pad_mode 2 = pre-padded input, pad 0");
    DVS_REQUIRE(!dact || y_out, "dvs_conv2d_dgrad: activation gradient needs the forward output");
    DVS_REQUIRE((double)d->B * d->H * d->W * (d->Cin > d->Cout ? d->Cin : d->Cout) < 2147483648.0,
                "dvs_conv2d_dgrad: tensors must have fewer than 2^31 elements (32-bit gather offsets)");
    FwdParams p{};
    p.x = dy; p.w = wt; p.y = dx;
    p.stat_split = 0x7fffffff;
    ConvShape& s = p.s;
    int Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1, Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    s.B = d->B;
    s.Ho = d->H; s.Wo = d->W; s.Cout = d->Cin;        // GEMM rows = input pixels, N = input channels
    s.H = Ho; s.W = Wo; s.Cin = d->Cout;              // gathered tensor = dY
    s.kh = d->kh; s.kw = d->kw; s.stride = d->stride; s.pad = d->pad; s.pad_mode = d->pad_mode;
    if (d->pad_mode == 2) {      // input = an already reflection-padded tensor (dvs_reflect_fold follows): zero padding here,
        s.pad_mode = PAD_ZERO;   // and the profile counts the unpadded problem's flops
        p.work_m = d->B * (d->H - 2) * (d->W - 2);
    }
    s.Ktot = d->kh * d->kw * d->Cout;
    p.t.aux = y_out;
    p.t.dact = dact;
    p.y2 = dx_skip;
    p.split_c1 = C1;
    p.res = residual;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!residual && d->pad_mode == PAD_REFLECT && d->kh == 3 && d->kw == 3 && d->stride == 1 && d->pad == 1 &&
        thin_dgrad(dy, wt, dx, y_out, dact, d->B, d->H, d->W, d->Cin, d->Cout, C1, dx_skip, st))
        return dvs::check_launch("dvs_conv2d_dgrad");      // 16-output-channel decoder layers: conv_thin.hip
    launch_mode<IN_DGRAD, false>(p, st, dvs::SLOT_CONV_DGRAD);
    return dvs::check_launch("dvs_conv2d_dgrad");
}

int dvs_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, const dvs_conv_desc* d,
                   const dvs_conv_fusion* f, void* stream) {
    DVS_REQUIRE(x && w && y && d, "dvs_conv2d_fwd: null pointer");
    DVS_REQUIRE(d->B > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0 && d->kh > 0 && d->kw > 0 &&
                    d->stride > 0 && d->pad >= 0,
                "dvs_conv2d_fwd: bad descriptor");
    FwdParams p{};
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.stat_split = 0x7fffffff;
    ConvShape& s = p.s;
    s.B = d->B; s.H = d->H; s.W = d->W; s.Cin = d->Cin; s.Cout = d->Cout;
    s.kh = d->kh; s.kw = d->kw; s.stride = d->stride; s.pad = d->pad; s.pad_mode = d->pad_mode;
    s.Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    s.Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    DVS_REQUIRE(s.Ho > 0 && s.Wo > 0, "dvs_conv2d_fwd: empty output");
    DVS_REQUIRE((double)d->B * d->H * d->W * d->Cin < 2147483648.0 && (double)d->B * s.Ho * s.Wo * d->Cout < 2147483648.0,
                "dvs_conv2d_fwd: tensors must have fewer than 2^31 elements (32-bit gather offsets)");
    DVS_REQUIRE(d->pad_mode == PAD_ZERO || (d->pad < d->H && d->pad < d->W), "dvs_conv2d_fwd: reflect pad too large");
    int planar = 0;
    if (f) {
        p.t.x2 = f->x2; p.t.C1 = f->C1; p.t.in_scale = f->in_scale; p.t.in_shift = f->in_shift;
        p.t.in_relu = f->in_relu; planar = f->nchw_planar;
        p.act = f->act; p.stats = f->stats; p.res = f->residual;
        DVS_REQUIRE(f->act >= 0 && f->act <= ACT_GELU, "dvs_conv2d_fwd: unknown activation %d", f->act);
        DVS_REQUIRE(!(f->residual && f->stats), "dvs_conv2d_fwd: residual and stats are exclusive (inference vs training)");
        DVS_REQUIRE(f->stat_groups >= 0 && f->stat_groups <= 2 && (f->stat_groups != 2 || (d->B % 2) == 0),
                    "dvs_conv2d_fwd: stat_groups is 0, 1 or 2 (2 needs an even batch)");
        p.stat_split = (f->stat_groups == 2) ? (d->B / 2) * s.Ho * s.Wo : 0x7fffffff;
        if (f->stats && f->stat_slots > 1) {
            DVS_REQUIRE(f->stat_slots <= 64 && (f->stat_slots & (f->stat_slots - 1)) == 0 && !f->nchw_planar,
                        "dvs_conv2d_fwd: stat_slots must be a power of two <= 64 (NHWC input only)");
            p.stat_mask = f->stat_slots - 1;
            p.stat_stride = (f->stat_groups == 2 ? 2 : 1) * 2 * d->Cout;
        }
        // C1 == Cin: upsample only (x2 is never read); otherwise the concat boundary must not split a 32-k stage
        DVS_REQUIRE(!(f->x2) || (f->C1 > 0 && (d->H & 1) == 0 && (d->W & 1) == 0 &&
                                 ((f->C1 == d->Cin && (f->C1 & 3) == 0) || (f->C1 < d->Cin && (f->C1 % BK) == 0))),
                    "dvs_conv2d_fwd: upsample+concat needs C1 %% 32 == 0 (or C1 == Cin) and even H, W");
        DVS_REQUIRE(!(f->x2 && planar), "dvs_conv2d_fwd: planar input cannot be concatenated");
        DVS_REQUIRE((f->in_scale == nullptr) == (f->in_shift == nullptr), "dvs_conv2d_fwd: in_scale/in_shift come together");
    }
    if (planar) {
        DVS_REQUIRE(d->kw <= 8 && d->pad_mode == PAD_ZERO, "dvs_conv2d_fwd: planar input supports kw <= 8, zero padding");
        s.Ktot = d->Cin * d->kh * 8;      // weights packed [Cout][Cin][kh][8]
    } else {
        DVS_REQUIRE((d->Cin & 3) == 0, "dvs_conv2d_fwd: NHWC input needs Cin %% 4 == 0 (got %d)", d->Cin);
        s.Ktot = d->kh * d->kw * d->Cin;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool fold = p.t.in_scale != nullptr;
    // (DVS_BF16_STEM=1: the stems on the generic planar kernel with bf16 tiles -- measured SLOWER than the fp32 stem kernels, 17.0 vs 16.2 ms per step)
    static const bool stem16 = [] { const char* e = getenv("DVS_BF16_STEM"); return e && e[0] == '1'; }();
    if (planar && stem_shape(s) && !p.t.in_relu && bias == nullptr && p.act == ACT_NONE && !p.res && !(stem16 && dvs::precision_bf16())) {
        dvs::ProfScope prof(dvs::SLOT_CONV_FWD, st);
        prof.work(2.0 * s.B * s.Ho * s.Wo * s.Cout * (double)s.Cin * s.kh * s.kw);
        stem_fwd(x, w, y, p.stats, f && f->stat_groups == 2 ? 2 : 1, s, p.t.in_scale, p.t.in_shift, st);
    } else if (!planar && !p.stats && !p.res && thin_fwd(x, w, bias, y, s, p.t, p.act, st)) {
        // 16-output-channel decoder layers: conv_thin.hip
    } else if (planar) {
        if (fold) launch_mode<IN_PLANAR, true>(p, st, dvs::SLOT_CONV_FWD);
        else launch_mode<IN_PLANAR, false>(p, st, dvs::SLOT_CONV_FWD);
    } else if (p.t.x2) {
        if (fold) launch_mode<IN_UPCAT, true>(p, st, dvs::SLOT_CONV_FWD);
        else launch_mode<IN_UPCAT, false>(p, st, dvs::SLOT_CONV_FWD);
    } else {
        if (fold) launch_mode<IN_NHWC, true>(p, st, dvs::SLOT_CONV_FWD);
        else launch_mode<IN_NHWC, false>(p, st, dvs::SLOT_CONV_FWD);
    }
    return dvs::check_launch("dvs_conv2d_fwd");
}

}  // extern "C"
