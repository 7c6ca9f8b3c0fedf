// a1-a3 backward: weight gradient of the convolutions as a split-K implicit GEMM on the fp32 matrix cores.
//
//   dW[co][k] = sum_m dY[m][co] * A[m][k],   m = (b,oy,ox) over all output pixels, k = (kh,kw,ci)
//
// The reduction dimension is the pixel index: a workgroup owns a BM(co) x BN(k) tile of dW and a slice of
// the pixels (blockIdx.z), stages 32 pixels per step -- the dY rows (optionally multiplied by the
// activation derivative act'(Y)) and the matching im2col rows, gathered exactly as in the forward kernel
// (padding, upsample+concat, BatchNorm fold) -- into LDS as [pixel][channel] and feeds
// v_mfma_f32_32x32x2_f32 with one ds_read_b32 per operand (lanes run along the contiguous channel
// dimension: conflict-free).  Partial tiles are combined with fp32 global atomics (128-B segments), the
// bias gradient (column sums of dY) rides along in the k-tile-0 workgroups.
//
// Replaces the weight/bias gradient of nn.Conv2d (autograd of the modules cited in conv_fwd.hip).
#include "conv_common.h"

#include <cstdlib>

namespace {
using namespace dvsconv;

constexpr int BP = 32;   // pixels per stage

struct WgradParams {
    const float* x;      // forward input (gather source)
    const float* dy;     // [B,Ho,Wo,Cout]
    float* dw;           // [Cout][Ktot], accumulated with atomics (caller zero-fills)
    float* dbias;        // [Cout] or NULL
    ConvShape s;
    InXform t;           // forward input transform + (aux = Y, dact) for the dY side
    int m_per_split;     // pixels per split, multiple of BP
    Grid3 g;             // logical grid: Cout tiles, (tap,ci) tiles, pixel splits (launched 1-D, see xcd_logical)
    int dbg;             // timing experiment (DVS_CONV_DEBUG_NOBARRIER & 4): MFMA + LDS reads only -> wrong results
    float* part;         // null, or the slab workspace [split][tile][wave][register][lane]: every workgroup stores its accumulators
                         // there with plain coalesced stores and wgrad_reduce_kernel adds the splits into dw in a fixed order
    size_t part_bytes;
};

// Epilogue of the three kernels: the workgroup's BM x BN tile either goes into dw with float atomics (which execute at the
// memory side at ~1.3 TB/s chip-wide, a fifth of the plain-store rate: MI355X_MICROARCH.md, Global float atomics) or, with a slab
// workspace, into its own slab.  C/D map: column (k) = lane & 31, row (co) = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5).
template <int TM, int TN, int WN>
__device__ __forceinline__ void wgrad_store(const WgradParams& p, const f32x16 (&acc)[TM][TN], int tile, int bid_z, int co0, int k0, int wave,
                                            int lane) {
    const ConvShape& s = p.s;
    const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
    if (p.part) {
        float* o = p.part + (((size_t)bid_z * (p.g.x * p.g.y) + tile) * 4 + wave) * (TM * TN * 16 * 64) + lane;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[((tm * TN + tn) * 16 + i) * 64] = acc[tm][tn][i];
        return;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int kk = k0 + (wn * TN + tn) * 32 + r;
        if (kk >= s.Ktot) continue;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int cb = co0 + (wm * TM + tm) * 32 + 4 * h;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                int c = cb + (i & 3) + 8 * (i >> 2);
                if (c < s.Cout) atomicAdd(p.dw + (size_t)c * s.Ktot + kk, acc[tm][tn][i]);
            }
        }
    }
}

// dw += sum over the splits of the slabs, in split order.  One thread per accumulator element of a tile; grid = tiles x
// (BM x BN / 256) workgroups.  Reads and writes are coalesced (a wave's 64 lanes = two 32-column rows of dw).
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(NT) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int Cout, int Ktot, int gx,
                                                          int tiles, int splits) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32, PER_WAVE = TM * TN * 16 * 64, PER_TILE = 4 * PER_WAVE;
    const int tile = blockIdx.x / (PER_TILE / NT), e = (blockIdx.x % (PER_TILE / NT)) * NT + threadIdx.x;
    const int wave = e / PER_WAVE, rem = e % PER_WAVE, reg = rem >> 6, lane = rem & 63;
    const int tm = reg / (TN * 16), tn = (reg / 16) % TN, i = reg & 15;
    const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
    const int bx = tile % gx, by = tile / gx;
    const int c = bx * BM + (wm * TM + tm) * 32 + 4 * h + (i & 3) + 8 * (i >> 2), kk = by * BN + (wn * TN + tn) * 32 + r;
    if (c >= Cout || kk >= Ktot) return;
    const float* src = part + (size_t)tile * PER_TILE + e;
    float sum = 0.f;
    for (int z = 0; z < splits; ++z) sum += src[(size_t)z * tiles * PER_TILE];
    dw[(size_t)c * Ktot + kk] += sum;
}


// BF16 (dvs_set_precision(1)): the tiles hold bf16, still [pixel][channel] -- the reduction index of this GEMM is the pixel, so both
// operands of v_mfma_f32_32x32x16_bf16 (A[row co][k = 8 h + j], B[k = 8 h + j][col]) are k-strided in that image; they are fetched
// with gfx950's transposed read `ds_read_b64_tr_b16` (per 16 lanes a block of 4 pixels x 16 channels, lane i receives channel i of
// the four pixels; lane 4 q + p supplies the address of pixel q, channels 4 p ..): two reads per operand and 16-pixel step.  Row
// strides are padded to 64 or 192 bytes modulo 256 so that the four pixel rows a 32-lane half reads land on disjoint banks.
constexpr int tr_pad(int n) { return ((2 * n) % 256 == 64 || (2 * n) % 256 == 192) ? n : n + 32; }
using s16x4 = __attribute__((ext_vector_type(4))) short;
using s16x8 = __attribute__((ext_vector_type(8))) short;
__device__ __forceinline__ bf16x8 tr_frag(const __bf16* lo, const __bf16* hi) {
    using lds_ptr = __attribute__((address_space(3))) s16x4*;
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(lo));
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(hi));
    return __builtin_bit_cast(bf16x8, s16x8{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]});
}

template <int BM, int BN, int WM, int WN, int MODE, bool FOLD, bool BF16 = false>
__global__ __launch_bounds__(NT) void conv_wgrad_kernel(WgradParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int D_VECS = BM / 32, X_VECS = BN / 32;       // 16-byte vectors per thread per stage
    constexpr int DV = BM / 4, XV = BN / 4;                 // vectors per pixel row
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves");
    constexpr int LDM = tr_pad(BM), LDN = tr_pad(BN);       // BF16: row strides (elements) of the two images
    __shared__ __attribute__((aligned(16))) float Ds[2][BP][BF16 ? LDM / 2 : BM];      // (BF16: the same bytes hold [2][BP][LDM] bf16)
    __shared__ __attribute__((aligned(16))) float Xs[2][BP][BF16 ? LDN / 2 : BN];
    __bf16* const Dh = reinterpret_cast<__bf16*>(&Ds[0][0][0]);
    __bf16* const Xh = reinterpret_cast<__bf16*>(&Xs[0][0][0]);
    // Gather-offset ring (all modes but the planar stem): the two divisions that turn a pixel index into (b, oy, ox)
    // and the padding / reflection / upsample arithmetic of a tap depend on the PIXEL and the TAP only, yet every
    // one of the Cin/4 lanes that fetch a slice of that pixel used to redo them every stage (~100 VALU per 16-byte
    // vector; the fp32 MFMA shares the vector ALU, so that was most of the kernel).  Now each (pixel, tap) offset is
    // computed once, by one thread, into LDS: 2 halves x RP pixels x up to 9 taps, refilled every RP/BP stages.
    constexpr int RP = 64, RTAPS = 9;
    constexpr bool RING = MODE != IN_PLANAR;
    constexpr int ROFF = (MODE == IN_UPCAT) ? 2 : 1;        // IN_UPCAT: one offset per source
    __shared__ int rtab[RING ? 2 * RP * RTAPS * ROFF : 1];
    constexpr int NO_TAP = -2147483647 - 1;
    float* sBias = &Ds[0][0][0];    // reused after the pixel loop (the loop ends on a barrier)

    const ConvShape& s = p.s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int M = s.B * s.Ho * s.Wo;
    // Cout tile fastest, then (tap,ci) tile, then pixel split: the tiles of one split read the same pixels
    int lg = xcd_logical(blockIdx.x, p.g.x * p.g.y * p.g.z, p.g.remap);
    const int bid_x = lg % p.g.x;
    lg /= p.g.x;
    const int bid_y = lg % p.g.y, bid_z = lg / p.g.y;
    const int co0 = bid_x * BM, k0 = bid_y * BN;
    const int m_begin = bid_z * p.m_per_split, m_end = min(M, m_begin + p.m_per_split);
    const bool do_bias = p.dbias != nullptr && bid_y == 0;

    // dY side: my channel vector and pixel rows
    const int d_c = (tid % DV) * 4, d_p0 = tid / DV;        // rows d_p0 + (NT/DV) * j
    const int co = co0 + d_c;
    const bool co_ok = co < s.Cout;                          // Cout % 4 == 0: whole vector in or out
    // X side: my k vector (fixed for the whole kernel) and pixel rows
    const int x_c = (tid % XV) * 4, x_p0 = tid / XV;
    const int k = k0 + x_c;
    const bool k_ok = k < s.Ktot;
    int ky = 0, kx = 0, ci = 0;
    if (MODE != IN_PLANAR) {
        int kc = min(k, s.Ktot - 4);
        int tap = kc / s.Cin;
        ci = kc - tap * s.Cin;
        ky = tap / s.kw;
        kx = tap - ky * s.kw;
    }

    // stage registers: raw loads + validity; zeros / folds are applied in store_stage, AFTER the MFMAs
    f32x4 rd[D_VECS], rdy[D_VECS], rx[X_VECS], bsum = {0.f, 0.f, 0.f, 0.f};
    f32x4 fsc = {1.f, 1.f, 1.f, 1.f}, fsh = {0.f, 0.f, 0.f, 0.f};
    bool rd_ok[D_VECS], rx_ok[X_VECS];
    unsigned rx_mask[X_VECS];
    float psc = 1.f, psh = 0.f;
    if (FOLD && MODE != IN_PLANAR) {     // my k-slice (hence its channels) is fixed for the whole kernel
        fsc = *reinterpret_cast<const f32x4*>(p.t.in_scale + ci);
        fsh = *reinterpret_cast<const f32x4*>(p.t.in_shift + ci);
    }
    // ---- ring fill: thread -> (slot = tid % RP, taps tid / RP, + NT/RP, ...) of one half --------------------------
    const int ntap = (MODE == IN_PLANAR) ? 0 : s.kh * s.kw;
    const bool ring_ok = RING && ntap <= RTAPS;              // otherwise the per-vector path below
    int f_b = 0, f_oy = 0, f_ox = 0, f_m = 0;               // my ring pixel: slot tid % RP of the next block of RP pixels
    auto ring_seek = [&](int m) {                           // (divisions: once per kernel)
        f_m = m;
        const int mc = min(m, M - 1);
        f_b = mc / (s.Ho * s.Wo);
        const int rem = mc - f_b * (s.Ho * s.Wo);
        f_oy = rem / s.Wo;
        f_ox = rem - f_oy * s.Wo;
    };
    auto ring_fill = [&](int half) {                        // writes the entries of pixel f_m, then advances it
        if (RING) {
            const int slot = tid % RP;
            for (int t = tid / RP; t < ntap; t += NT / RP) {
                const int tky = t / s.kw, tkx = t - tky * s.kw;
                bool ok = f_m < m_end;
                int off = 0, off2 = 0;
                tap_setup<MODE == IN_PLANAR ? IN_NHWC : MODE>(s, p.t, f_b, f_oy * s.stride - s.pad + tky,
                                                              f_ox * s.stride - s.pad + tkx, ok, off, off2);
                int* e = rtab + ((half * RP + slot) * RTAPS + t) * ROFF;
                e[0] = ok ? off : NO_TAP;
                if (ROFF == 2) e[1] = ok ? off2 : NO_TAP;
            }
            f_m += RP;                                       // the halves are filled alternately: next block of RP pixels
            f_ox += RP;
            while (f_ox >= s.Wo) {
                f_ox -= s.Wo;
                if (++f_oy == s.Ho) {
                    f_oy = 0;
                    f_b = min(f_b + 1, s.B - 1);
                }
            }
        }
    };
    const int my_tap = ky * s.kw + kx;
    const bool src2 = MODE == IN_UPCAT && ci >= p.t.C1;     // my channel slice lives in the second (skip) source
    const float* const x_src = src2 ? p.t.x2 : p.x;
    // dY rows: pointer of my slice in the first stage, advanced by BP rows per stage
    const float* d_ptr[D_VECS];
    int d_m[D_VECS];
#pragma unroll
    for (int j = 0; j < D_VECS; ++j) {
        d_m[j] = m_begin + d_p0 + (NT / DV) * j;
        d_ptr[j] = p.dy + (size_t)d_m[j] * s.Cout + min(co, s.Cout - 4);
    }
    const ptrdiff_t aux_delta = p.t.dact ? (p.t.aux - p.dy) : 0;
    const size_t d_step = (size_t)BP * s.Cout;
    auto load_stage_ring = [&](int mb) {
#pragma unroll
        for (int j = 0; j < D_VECS; ++j) {
            rd_ok[j] = d_m[j] < m_end && co_ok;
            const float* a = rd_ok[j] ? d_ptr[j] : p.dy;            // always a valid address, no branch around the load
            rd[j] = *reinterpret_cast<const f32x4*>(a);
            if (p.t.dact) rdy[j] = *reinterpret_cast<const f32x4*>(a + aux_delta);
            d_m[j] += BP;
            d_ptr[j] += d_step;
        }
        const int ring = (mb - m_begin) & (2 * RP - 1);
#pragma unroll
        for (int j = 0; j < X_VECS; ++j) {
            const int slot = ring + x_p0 + (NT / XV) * j;
            const int off = rtab[(slot * RTAPS + my_tap) * ROFF + (src2 ? 1 : 0)];
            const bool ok = k_ok && off != NO_TAP;
            const float* a = ok ? x_src + (off + ci) : p.x;
            rx[j] = *reinterpret_cast<const f32x4*>(a);
            rx_ok[j] = ok;
        }
    };
    auto load_stage = [&](int mb) {
#pragma unroll
        for (int j = 0; j < D_VECS; ++j) {
            int m = mb + d_p0 + (NT / DV) * j;
            rd_ok[j] = m < m_end && co_ok;
            size_t o = (size_t)min(m, M - 1) * s.Cout + min(co, s.Cout - 4);
            rd[j] = *reinterpret_cast<const f32x4*>(p.dy + o);
            if (p.t.dact) rdy[j] = *reinterpret_cast<const f32x4*>(p.t.aux + o);
        }
#pragma unroll
        for (int j = 0; j < X_VECS; ++j) {
            int m = mb + x_p0 + (NT / XV) * j;
            bool ok = m < m_end && k_ok;
            m = min(m, M - 1);
            int b = m / (s.Ho * s.Wo), rem = m - b * (s.Ho * s.Wo);
            int oy = rem / s.Wo, ox = rem - oy * s.Wo;
            int iy = oy * s.stride - s.pad, ix = ox * s.stride - s.pad;
            if (MODE == IN_PLANAR) {
                int pci;
                rx[j] = gather_planar_raw(p.x, s, b, iy, ix, min(k, s.Ktot - 4), ok, rx_mask[j], pci);
                if (FOLD && j == 0) {
                    psc = p.t.in_scale[pci];
                    psh = p.t.in_shift[pci];
                }
            } else {
                rx[j] = gather_raw<MODE>(p.x, s, p.t, b, iy + ky, ix + kx, ci, ok);
            }
            rx_ok[j] = ok;
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < D_VECS; ++j) {
            f32x4 v = rd[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float u = p.t.dact ? v[e] * act_grad_from_out(rdy[j][e], p.t.dact) : v[e];
                v[e] = rd_ok[j] ? u : 0.f;
            }
            if constexpr (BF16) *reinterpret_cast<bf16x4*>(Dh + (buf * BP + d_p0 + (NT / DV) * j) * LDM + d_c) = to_bf16(v);
            else *reinterpret_cast<f32x4*>(&Ds[buf][d_p0 + (NT / DV) * j][d_c]) = v;
            bsum += v;
        }
#pragma unroll
        for (int j = 0; j < X_VECS; ++j) {
            f32x4 v;
            if (MODE == IN_PLANAR) v = finalize_planar<FOLD>(rx[j], rx_mask[j], psc, psh);
            else v = finalize<FOLD>(rx[j], rx_ok[j], fsc, fsh, p.t.in_relu);
            if constexpr (BF16) *reinterpret_cast<bf16x4*>(Xh + (buf * BP + x_p0 + (NT / XV) * j) * LDN + x_c) = to_bf16(v);
            else *reinterpret_cast<f32x4*>(&Xs[buf][x_p0 + (NT / XV) * j][x_c]) = v;
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    const int r = lane & 31, h = lane >> 5;
    const int a_col = wm * TM * 32 + r, b_col = wn * TN * 32 + r;
    if (ring_ok) {
        ring_seek(m_begin + tid % RP);
        ring_fill(0);                                         // pixels [0, RP) of my range
        ring_fill(1);                                         // pixels [RP, 2 RP)
        __syncthreads();
    }
    if (m_begin < m_end) {
        if (ring_ok) load_stage_ring(m_begin);
        else load_stage(m_begin);
        store_stage(0);
    }
    __syncthreads();
    int buf = 0, stage = 0;
    constexpr int SPH = RP / BP;                              // stages per ring half
#pragma unroll 1
    for (int mb = m_begin; mb < m_end; mb += BP, ++stage) {
        const bool more = mb + BP < m_end;
        // at stage q*SPH (q >= 1) the loads issued from now on read half q & 1 (and later); refill the other one with
        // the pixels of half-index q + 1
        if (ring_ok && stage > 0 && (stage % SPH) == 0) {
            if (((stage / SPH) & 1) == 0) ring_fill(1);       // q even: half 1 is free  (it held q - 1)
            else ring_fill(0);
        }
        if (more) {
            if (ring_ok) load_stage_ring(mb + BP);
            else load_stage(mb + BP);
        }
        if constexpr (BF16) {
            // my block of a transposed read: pixel row (8 h + q) of the 16-pixel step, channels 16 (g & 1) + 4 pp .. of a 32-column tile
            const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
            const int prow = 8 * (g >> 1) + q, pcol = 16 * (g & 1) + 4 * pp;
#pragma unroll
            for (int t = 0; t < BP / 16; ++t) {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int m = 0; m < TM; ++m) {
                    const __bf16* base = Dh + (buf * BP + 16 * t + prow) * LDM + wm * TM * 32 + m * 32 + pcol;
                    a[m] = tr_frag(base, base + 4 * LDM);
                }
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    const __bf16* base = Xh + (buf * BP + 16 * t + prow) * LDN + wn * TN * 32 + n * 32 + pcol;
                    b[n] = tr_frag(base, base + 4 * LDN);
                }
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
            }
        } else {
#pragma unroll
        for (int t = 0; t < BP / 2; ++t) {
            float a[TM], b[TN];
#pragma unroll
            for (int m = 0; m < TM; ++m) a[m] = Ds[buf][2 * t + h][a_col + m * 32];
#pragma unroll
            for (int n = 0; n < TN; ++n) b[n] = Xs[buf][2 * t + h][b_col + n * 32];
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        }
        if (more) store_stage(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // epilogue: dW[co][k] += acc.  C/D map: column (k) = lane & 31, row (co) = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
    wgrad_store<TM, TN, WN>(p, acc, bid_y * p.g.x + bid_x, bid_z, co0, k0, wave, lane);
    if (do_bias) {
        if (tid < BM) sBias[tid] = 0.f;
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(&sBias[d_c + e], bsum[e]);
        __syncthreads();
        if (tid < BM && co0 + tid < s.Cout) atomicAdd(p.dbias + co0 + tid, sBias[tid]);
    }
}

// LDS-DMA variant (plain NHWC / upsample+concat input, no BatchNorm fold, no activation derivative, no bias):
// the [pixel][channel] tiles are filled by `global_load_lds_dwordx4` (1 KB = 256 consecutive floats of a tile per
// instruction, padding / tails served from a 16-byte zero page), so a stage costs no VGPR staging, no padding
// selects and no ds_write pass.  Operand fetch is the same conflict-free ds_read_b32 as above (lanes run along
// the channel dimension), hence no swizzle is needed here.
__device__ __attribute__((aligned(16))) float g_dvs_zero_page_w[4] = {0.f, 0.f, 0.f, 0.f};

// VALU diet (the fp32 MFMA shares the vector ALU, see conv_dma.h): a lane's tap is fixed for the whole kernel and
// only the pixel advances, so everything about a PIXEL -- image, window origin, base offset: two divisions -- is
// computed once per pixel by one thread into a small LDS ring (256 pixels per half, refilled every 8 stages), and a
// stage's per-row work is a ds_read_b64 plus two bounds compares, one add, one 64-bit shift-add and two selects
// (~10 VALU instead of ~40 of incremental coordinate updates + padding arithmetic).  Zero padding, NHWC input only:
// the encoder convolutions, which are all this kernel serves.
constexpr int PT = 256;          // pixels per half of the pixel ring (8 or 16 stages)

template <int BM, int BN, int WM, int WN, int MODE, int BPD>
__global__ __launch_bounds__(NT) void conv_wgrad_dma_kernel(WgradParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int D_INS = BM * BPD / 1024, X_INS = BN * BPD / 1024;   // DMA instructions (1 KB each) per wave per stage
    constexpr int SPR = PT / BPD;                            // stages per ring half
    static_assert(D_INS >= 1 && X_INS >= 1 && (SPR & (SPR - 1)) == 0, "stage shape");
    constexpr int DV = BM / 4, XV = BN / 4;                  // 16-byte slots per pixel row
    constexpr int D_RPI = 64 / DV, X_RPI = 64 / XV;          // pixel rows per instruction
    static_assert(WM * WN == 4 && DV <= 64 && XV <= 64, "tile");
    static_assert(MODE == IN_NHWC, "plain NHWC input");
    __shared__ __attribute__((aligned(16))) float Ds[2][BPD][BM];
    __shared__ __attribute__((aligned(16))) float Xs[2][BPD][BN];
    __shared__ int2 ptab[2][PT];                             // .x: offset of the window origin, .y: y0 << 16 | x0 & 0xffff

    const ConvShape& s = p.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int M = s.B * s.Ho * s.Wo;
    // Cout tile fastest, then (tap,ci) tile, then pixel split: the tiles of one split read the same pixels
    int lg = xcd_logical(blockIdx.x, p.g.x * p.g.y * p.g.z, p.g.remap);
    const int bid_x = lg % p.g.x;
    lg /= p.g.x;
    const int bid_y = lg % p.g.y, bid_z = lg / p.g.y;
    const int co0 = bid_x * BM, k0 = bid_y * BN;
    const int m_begin = bid_z * p.m_per_split, m_end = min(M, m_begin + p.m_per_split);

    // one pixel of the ring per thread: (b, oy, ox) -> byte offset of the window origin and a 9-bit mask of the taps that
    // fall inside the image (the 2-D bounds test, done once per pixel instead of once per lane and stage)
    auto fill_ring = [&](int half, int m_first) {
        const int m = m_first + tid;
        int2 e;
        e.x = 0;
        e.y = 0;                                             // no tap valid
        if (m < m_end) {
            const int b = m / (s.Ho * s.Wo), rem = m - b * (s.Ho * s.Wo);
            const int oy = rem / s.Wo, ox = rem - oy * s.Wo;
            const int y0 = oy * s.stride - s.pad, x0 = ox * s.stride - s.pad;
            e.x = ((b * s.H + y0) * s.W + x0) * s.Cin * 4;
            int mask = 0;
            for (int t = 0; t < s.kh * s.kw; ++t) {
                const int tky = t / s.kw, tkx = t - tky * s.kw;
                if ((unsigned)(y0 + tky) < (unsigned)s.H && (unsigned)(x0 + tkx) < (unsigned)s.W) mask |= 1 << t;
            }
            e.y = mask;
        }
        ptab[half][tid] = e;
    };

    // Buffer-addressed LDS-DMA (see conv_dma.h): per-lane byte offsets are loop-invariant, the stage's pixel block is a
    // scalar offset, lanes that must read zeros carry an out-of-range offset.  The dY descriptor ends at this split's
    // last pixel, so the tail rows of the last stage are zero-filled by the range check.
    constexpr int OOB = OOB_OFF;
    const auto rdy = dma_rsrc(p.dy, (size_t)m_end * s.Cout * 4);
    const auto rxs = dma_rsrc(p.x, (size_t)s.B * s.H * s.W * s.Cin * 4);
    const int d_c = (lane % DV) * 4, d_r = lane / DV;
    int d_off[D_INS];
#pragma unroll
    for (int j = 0; j < D_INS; ++j)
        d_off[j] = (co0 + d_c < s.Cout) ? (((wave * D_INS + j) * D_RPI + d_r) * s.Cout + co0 + d_c) * 4 : OOB;
    // X rows: my tap and channel slice are fixed; the pixel comes from the ring
    const int x_c = (lane % XV) * 4, x_r = lane / XV;
    const int k = k0 + x_c;
    const bool k_ok = k < s.Ktot;
    const int kc = min(k, s.Ktot - 4), tap = kc / s.Cin, ci = kc - tap * s.Cin, ky = tap / s.kw, kx = tap - ky * s.kw;
    const int lane_off = ((ky * s.W + kx) * s.Cin + ci) * 4;
    const int tap_bit = k_ok ? (1 << tap) : 0;

    auto issue_stage = [&](int mb, int buf) {
        const int ring = (mb - m_begin) & (2 * PT - 1);               // position of the stage's first pixel in the ring
        const int d_soff = mb * s.Cout * 4;                           // scalar: first pixel row of the stage
#pragma unroll
        for (int j = 0; j < D_INS; ++j)
            dma16_buf(rdy, d_off[j], d_soff, &Ds[buf][(wave * D_INS + j) * D_RPI][0]);
#pragma unroll
        for (int j = 0; j < X_INS; ++j) {
            const int slot = ring + (wave * X_INS + j) * X_RPI + x_r;
            const int2 e = (&ptab[0][0])[slot];
            const int voff = (e.y & tap_bit) ? e.x + lane_off : OOB;
            dma16_buf(rxs, voff, 0, &Xs[buf][(wave * X_INS + j) * X_RPI][0]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    const int r = lane & 31, h = lane >> 5;
    const int a_col = wm * TM * 32 + r, b_col = wn * TN * 32 + r;
    fill_ring(0, m_begin);
    fill_ring(1, m_begin + PT);
    __syncthreads();
    if (m_begin < m_end) issue_stage(m_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0, stage = 0;
#pragma unroll 1
    for (int mb = m_begin; mb < m_end; mb += BPD, ++stage) {
        // refill the half of the ring that the stages issued from now on no longer touch: at stage 8q (q >= 1) the
        // next issue reads pixels of stage 8q+1, which live in half q & 1; half (q-1) & 1 ... is the one to reuse
        if ((stage & (SPR - 1)) == 0 && stage > 0) fill_ring(((stage / SPR) + 1) & 1, m_begin + ((stage / SPR) + 1) * PT);
        if (!p.dbg && mb + BPD < m_end) issue_stage(mb + BPD, buf ^ 1);
#pragma unroll
        for (int t = 0; t < BPD / 2; ++t) {
            float a[TM], b[TN];
#pragma unroll
            for (int m = 0; m < TM; ++m) a[m] = Ds[buf][2 * t + h][a_col + m * 32];
#pragma unroll
            for (int n = 0; n < TN; ++n) b[n] = Xs[buf][2 * t + h][b_col + n * 32];
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
    }
    wgrad_store<TM, TN, WN>(p, acc, bid_y * p.g.x + bid_x, bid_z, co0, k0, wave, lane);
}

// LDS-DMA weight gradient for the decoder's gathers (reflection padding, nearest-upsample + concat), once the activation
// derivative has been applied to dY beforehand (dvs_act_bwd) and the bias gradient taken there: same tiles and MFMA loop
// as above, but a pixel's tap offsets (reflected / upsampled coordinates, one per source tensor) come from a ring of
// per-(pixel, tap) element offsets -- computed once per pixel by one thread, as in the register-staged kernel -- and the
// X rows are fetched with flat `global_load_lds_dwordx4` (a lane picks its source tensor by its channel slice; rows
// past the end of the split fetch the zero page).
constexpr int PTG = 128;         // pixels per half of the offset ring (8 stages of 16 pixels)

__device__ __forceinline__ void dma16_flat(const float* gp, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(NT) void conv_wgrad_dma_gen_kernel(WgradParams p) {
    constexpr int BPD = 16;
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int D_INS = BM * BPD / 1024, X_INS = BN * BPD / 1024;
    constexpr int DV = BM / 4, XV = BN / 4;
    constexpr int D_RPI = 64 / DV, X_RPI = 64 / XV;
    constexpr int SPR = PTG / BPD;
    constexpr int NSRC = (MODE == IN_UPCAT) ? 2 : 1, RT = 9;
    constexpr int NO_TAP = -2147483647 - 1;
    static_assert(WM * WN == 4 && DV <= 64 && XV <= 64 && D_INS >= 1 && X_INS >= 1, "tile");
    __shared__ __attribute__((aligned(16))) float Ds[2][BPD][BM];
    __shared__ __attribute__((aligned(16))) float Xs[2][BPD][BN];
    __shared__ int otab[2 * PTG * RT * NSRC];

    const ConvShape& s = p.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int M = s.B * s.Ho * s.Wo;
    int lg = xcd_logical(blockIdx.x, p.g.x * p.g.y * p.g.z, p.g.remap);
    const int bid_x = lg % p.g.x;
    lg /= p.g.x;
    const int bid_y = lg % p.g.y, bid_z = lg / p.g.y;
    const int co0 = bid_x * BM, k0 = bid_y * BN;
    const int m_begin = bid_z * p.m_per_split, m_end = min(M, m_begin + p.m_per_split);

    // ring fill: thread -> (slot = tid % PTG, taps tid / PTG, + NT / PTG, ...) of one half
    auto fill_ring = [&](int half, int m_first) {
        const int slot = tid % PTG, m = m_first + slot;
        const int mc = min(m, M - 1);
        const int b = mc / (s.Ho * s.Wo), rem = mc - b * (s.Ho * s.Wo);
        const int oy = rem / s.Wo, ox = rem - oy * s.Wo;
        for (int t = tid / PTG; t < RT; t += NT / PTG) {
            const int tky = t / 3, tkx = t - tky * 3;
            bool ok = m < m_end;
            int off = 0, off2 = 0;
            tap_setup<MODE>(s, p.t, b, oy - s.pad + tky, ox - s.pad + tkx, ok, off, off2);
            int* e = otab + ((half * PTG + slot) * RT + t) * NSRC;
            e[0] = ok ? off : NO_TAP;
            if (NSRC == 2) e[1] = ok ? off2 : NO_TAP;
        }
    };

    constexpr int OOB = OOB_OFF;
    const auto rdy = dma_rsrc(p.dy, (size_t)m_end * s.Cout * 4);
    const int d_c = (lane % DV) * 4, d_r = lane / DV;
    int d_off[D_INS];
#pragma unroll
    for (int j = 0; j < D_INS; ++j)
        d_off[j] = (co0 + d_c < s.Cout) ? (((wave * D_INS + j) * D_RPI + d_r) * s.Cout + co0 + d_c) * 4 : OOB;
    // X rows: my tap, channel slice and source tensor are fixed; the pixel's offsets come from the ring
    const int x_c = (lane % XV) * 4, x_r = lane / XV;
    const int k = k0 + x_c;
    const bool k_ok = k < s.Ktot;
    const int kc = min(k, s.Ktot - 4), tap = kc / s.Cin, ci = kc - tap * s.Cin;
    const bool src2 = MODE == IN_UPCAT && ci >= p.t.C1;
    const float* const x_src = (src2 ? p.t.x2 : p.x) + ci;          // tap_setup's second offset is rebased by -C1
    const int my_ent = tap * NSRC + (src2 ? 1 : 0);

    auto issue_stage = [&](int mb, int buf) {
        const int ring = (mb - m_begin) & (2 * PTG - 1);
        const int d_soff = mb * s.Cout * 4;
#pragma unroll
        for (int j = 0; j < D_INS; ++j)
            dma16_buf(rdy, d_off[j], d_soff, &Ds[buf][(wave * D_INS + j) * D_RPI][0]);
#pragma unroll
        for (int j = 0; j < X_INS; ++j) {
            const int slot = ring + (wave * X_INS + j) * X_RPI + x_r;
            const int e = otab[slot * RT * NSRC + my_ent];
            const float* a = (k_ok && e != NO_TAP) ? x_src + e : g_dvs_zero_page_w;
            dma16_flat(a, &Xs[buf][(wave * X_INS + j) * X_RPI][0]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    const int r = lane & 31, h = lane >> 5;
    const int a_col = wm * TM * 32 + r, b_col = wn * TN * 32 + r;
    fill_ring(0, m_begin);
    fill_ring(1, m_begin + PTG);
    __syncthreads();
    if (m_begin < m_end) issue_stage(m_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0, stage = 0;
#pragma unroll 1
    for (int mb = m_begin; mb < m_end; mb += BPD, ++stage) {
        if ((stage & (SPR - 1)) == 0 && stage > 0) fill_ring(((stage / SPR) + 1) & 1, m_begin + ((stage / SPR) + 1) * PTG);
        if (mb + BPD < m_end) issue_stage(mb + BPD, buf ^ 1);
#pragma unroll
        for (int t = 0; t < BPD / 2; ++t) {
            float a[TM], b[TN];
#pragma unroll
            for (int m = 0; m < TM; ++m) a[m] = Ds[buf][2 * t + h][a_col + m * 32];
#pragma unroll
            for (int n = 0; n < TN; ++n) b[n] = Xs[buf][2 * t + h][b_col + n * 32];
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        buf ^= 1;
    }
    wgrad_store<TM, TN, WN>(p, acc, bid_y * p.g.x + bid_x, bid_z, co0, k0, wave, lane);
}

template <int BM, int BN, int WM, int WN, int MODE, bool FOLD>
void launch_cfg(WgradParams p, hipStream_t st, size_t* need = nullptr) {
    const int M = p.s.B * p.s.Ho * p.s.Wo;
    const int tiles = ((p.s.Cout + BM - 1) / BM) * ((p.s.Ktot + BN - 1) / BN);
    static const int wg_target = [] { const char* e = getenv("DVS_WGRAD_WGS"); return e ? atoi(e) : 1024; }();
    int splits = (wg_target + tiles - 1) / tiles;            // ~4 workgroups per CU in flight
    splits = max(1, min(splits, (M + 255) / 256));           // at least 8 stages per workgroup
    int mps = ((M + splits - 1) / splits + BP - 1) / BP * BP;
    splits = (M + mps - 1) / mps;
    p.m_per_split = mps;
    dim3 grid((p.s.Cout + BM - 1) / BM, (p.s.Ktot + BN - 1) / BN, splits);
    static const int xcd_on = [] { const char* e = getenv("DVS_CONV_XCD"); return !(e && e[0] == '0') ? 1 : 0; }();
    p.g = Grid3{(int)grid.x, (int)grid.y, (int)grid.z, xcd_on};
    const size_t slab = (size_t)grid.x * grid.y * grid.z * BM * BN * sizeof(float);
    if (need) {                                            // workspace query: nothing is launched
        *need = slab;
        return;
    }
    if (p.part && p.part_bytes < slab) p.part = nullptr;   // (checked by the caller; never write past a short workspace)
    struct Reduce {                                        // after whichever kernel ran: the ordered second pass
        const WgradParams& p; dim3 g; hipStream_t st;
        ~Reduce() {
            if (p.part)
                hipLaunchKernelGGL((wgrad_reduce_kernel<BM, BN, WM, WN>), dim3(g.x * g.y * (BM * BN / NT)), dim3(NT), 0, st, p.part, p.dw,
                                   p.s.Cout, p.s.Ktot, (int)g.x, (int)(g.x * g.y), (int)g.z);
        }
    } reduce_after{p, grid, st};
    dvs::ProfScope prof(dvs::SLOT_CONV_WGRAD, st);
    const double k_real = (MODE == IN_PLANAR) ? (double)p.s.Cin * p.s.kh * p.s.kw : (double)p.s.Ktot;
    prof.work(2.0 * M * p.s.Cout * k_real);
    if constexpr (MODE == IN_PLANAR) {
        if (dvs::precision_bf16()) {       // the stems in the bf16 mode
            hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, MODE, FOLD, true>), dim3(grid.x * grid.y * grid.z), dim3(NT), 0, st, p);
            return;
        }
    }
    if constexpr (!FOLD && MODE != IN_PLANAR) {
        if (dvs::precision_bf16()) {       // bf16 tiles: the register-staged kernel (the LDS-DMA ones cannot convert on the way)
            hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, MODE, FOLD, true>), dim3(grid.x * grid.y * grid.z), dim3(NT), 0, st, p);
            return;
        }
        static const bool dma = [] { const char* e = getenv("DVS_CONV_DMA"); return !(e && e[0] == '0'); }();
        static const int dbg = dvs::experiment_flags("DVS_CONV_DEBUG_NOBARRIER");
        p.dbg = dbg & 4;
        if (dma && MODE == IN_NHWC && p.s.pad_mode == PAD_ZERO && p.t.dact == 0 && p.dbias == nullptr && p.s.kh * p.s.kw <= 30 &&
            (double)p.s.B * p.s.H * p.s.W * p.s.Cin < 536870912.0 && (double)M * p.s.Cout < 536870912.0) {
            // 16-pixel stages: half the LDS of 32-pixel ones, so four workgroups per CU instead of two hide each other's
            // vmcnt(0) + barrier at the end of a stage (83 -> 87 TF over the step's launches; DVS_WGRAD_BP=32 for the old shape)
            static const int bpd = [] { const char* e = getenv("DVS_WGRAD_BP"); return e ? atoi(e) : 16; }();
            if constexpr (BM >= 64) {
                if (bpd == 16) {
                    hipLaunchKernelGGL((conv_wgrad_dma_kernel<BM, BN, WM, WN, IN_NHWC, 16>), dim3(grid.x * grid.y * grid.z), dim3(NT), 0, st, p);
                    return;
                }
            }
            hipLaunchKernelGGL((conv_wgrad_dma_kernel<BM, BN, WM, WN, IN_NHWC, 32>), dim3(grid.x * grid.y * grid.z), dim3(NT), 0, st, p);
            return;
        }
    }
    if constexpr (!FOLD && MODE != IN_PLANAR && BM >= 64) {
        static const bool gen = [] { const char* e = getenv("DVS_WGRAD_GEN"); return !(e && e[0] == '0'); }();
        static const bool dma2 = [] { const char* e = getenv("DVS_CONV_DMA"); return !(e && e[0] == '0'); }();
        if (gen && dma2 && (MODE == IN_UPCAT || p.s.pad_mode == PAD_REFLECT) && p.t.dact == 0 && p.dbias == nullptr &&
            p.s.kh == 3 && p.s.kw == 3 && p.s.stride == 1 && (double)M * p.s.Cout < 536870912.0) {
            hipLaunchKernelGGL((conv_wgrad_dma_gen_kernel<BM, BN, WM, WN, MODE>), dim3(grid.x * grid.y * grid.z), dim3(NT), 0, st, p);
            return;
        }
    }
    hipLaunchKernelGGL((conv_wgrad_kernel<BM, BN, WM, WN, MODE, FOLD>), dim3(grid.x * grid.y * grid.z), dim3(NT), 0, st, p);
}

template <int MODE, bool FOLD>
void launch_mode(const WgradParams& p, hipStream_t st, size_t* need = nullptr) {
    if (p.s.Cout > 64) launch_cfg<128, 128, 2, 2, MODE, FOLD>(p, st, need);
    else if (p.s.Cout > 32) launch_cfg<64, 128, 1, 4, MODE, FOLD>(p, st, need);
    else launch_cfg<32, 128, 1, 4, MODE, FOLD>(p, st, need);
}

}  // namespace

extern "C" {

// workspace != null: ordered (slab) reduction; need != null: only report the slab size of this problem (0: a path without slabs)
static int wgrad_impl(const float* x, const float* dy, float* dw, float* dbias, const dvs_conv_desc* d, const dvs_conv_fusion* f,
                      const float* y_out, int dact, float* workspace, size_t workspace_bytes, size_t* need, void* stream) {
    DVS_REQUIRE(need || (x && dy && dw), "dvs_conv2d_wgrad: null pointer");
    DVS_REQUIRE(d, "dvs_conv2d_wgrad: null descriptor");
    DVS_REQUIRE((d->Cout & 3) == 0, "dvs_conv2d_wgrad: Cout must be a multiple of 4 (got %d)", d->Cout);
    DVS_REQUIRE(!dact || y_out, "dvs_conv2d_wgrad: activation gradient needs the forward output");
    WgradParams p{};
    p.x = x; p.dy = dy; p.dw = dw; p.dbias = dbias;
    p.part = workspace; p.part_bytes = workspace_bytes;
    ConvShape& s = p.s;
    s.B = d->B; s.H = d->H; s.W = d->W; s.Cin = d->Cin; s.Cout = d->Cout;
    s.kh = d->kh; s.kw = d->kw; s.stride = d->stride; s.pad = d->pad; s.pad_mode = d->pad_mode;
    s.Ho = (d->H + 2 * d->pad - d->kh) / d->stride + 1;
    s.Wo = (d->W + 2 * d->pad - d->kw) / d->stride + 1;
    int planar = 0;
    if (f) {
        p.t.x2 = f->x2; p.t.C1 = f->C1; p.t.in_scale = f->in_scale; p.t.in_shift = f->in_shift;
        p.t.in_relu = f->in_relu; planar = f->nchw_planar;
    }
    p.t.aux = y_out;
    p.t.dact = dact;
    if (planar) {
        DVS_REQUIRE(d->kw <= 8 && d->pad_mode == PAD_ZERO, "dvs_conv2d_wgrad: planar input supports kw <= 8, zero padding");
        s.Ktot = d->Cin * d->kh * 8;
    } else {
        DVS_REQUIRE((d->Cin & 3) == 0, "dvs_conv2d_wgrad: NHWC input needs Cin %% 4 == 0 (got %d)", d->Cin);
        s.Ktot = d->kh * d->kw * d->Cin;
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool fold = p.t.in_scale != nullptr;
    if (need) *need = 0;
    // (DVS_BF16_STEM=1: the stems on the generic planar kernel with bf16 tiles -- measured SLOWER than the fp32 stem kernels, 17.0 vs 16.2 ms per step)
    static const bool stem16 = [] { const char* e = getenv("DVS_BF16_STEM"); return e && e[0] == '1'; }();
    if (planar && stem_shape(s) && dact == 0 && dbias == nullptr && !p.t.in_relu && !(stem16 && dvs::precision_bf16())) {
        if (need) return DVS_OK;
        dvs::ProfScope prof(dvs::SLOT_CONV_WGRAD, st);
        prof.work(2.0 * s.B * s.Ho * s.Wo * s.Cout * (double)s.Cin * s.kh * s.kw);
        stem_wgrad(x, dy, dw, s, p.t.in_scale, p.t.in_shift, st);
    } else if (!planar && thin_wgrad_shape(s, p.t)) {
        // decoder layers with 16 / 32 output channels: conv_thin.hip (persistent workgroups, no slabs)
        if (need) return DVS_OK;
        thin_wgrad(x, dy, dw, dbias, s, p.t, st);
    } else if (planar) {
        if (fold) launch_mode<IN_PLANAR, true>(p, st, need);
        else launch_mode<IN_PLANAR, false>(p, st, need);
    } else if (p.t.x2) {
        if (fold) launch_mode<IN_UPCAT, true>(p, st, need);
        else launch_mode<IN_UPCAT, false>(p, st, need);
    } else {
        if (fold) launch_mode<IN_NHWC, true>(p, st, need);
        else launch_mode<IN_NHWC, false>(p, st, need);
    }
    return need ? DVS_OK : dvs::check_launch("dvs_conv2d_wgrad");
}

int dvs_conv2d_wgrad(const float* x, const float* dy, float* dw, float* dbias, const dvs_conv_desc* d,
                     const dvs_conv_fusion* f, const float* y_out, int dact, void* stream) {
    return wgrad_impl(x, dy, dw, dbias, d, f, y_out, dact, nullptr, 0, nullptr, stream);
}

size_t dvs_conv2d_wgrad_workspace(const dvs_conv_desc* d, const dvs_conv_fusion* f, int dact, int with_bias) {
    size_t need = 0;
    static float dummy;
    if (wgrad_impl(nullptr, nullptr, nullptr, with_bias ? &dummy : nullptr, d, f, dact ? &dummy : nullptr, dact, nullptr, 0, &need, nullptr) != DVS_OK)
        return 0;
    return need;
}

int dvs_conv2d_wgrad_ws(const float* x, const float* dy, float* dw, float* dbias, const dvs_conv_desc* d, const dvs_conv_fusion* f,
                        const float* y_out, int dact, float* workspace, size_t workspace_bytes, void* stream) {
    if (workspace) {
        const size_t need = dvs_conv2d_wgrad_workspace(d, f, dact, dbias != nullptr);
        DVS_REQUIRE(workspace_bytes >= need, "dvs_conv2d_wgrad_ws: workspace of %zu bytes, %zu needed", workspace_bytes, need);
    }
    return wgrad_impl(x, dy, dw, dbias, d, f, y_out, dact, workspace, workspace_bytes, nullptr, stream);
}

}  // extern "C"
