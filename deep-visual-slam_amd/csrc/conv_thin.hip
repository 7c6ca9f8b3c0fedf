// a2: row-ring kernels for the depth decoder's thin full-resolution layers (16 / 32 output channels: upconv_1_0, _1_1,
// _0_0, _0_1 of model/depthnet.py:49-62 -- ReflectionPad2d(1) + 3x3 conv + ELU on 120x160 ... 480x640 maps; the fused
// layers.py:106-136 ConvBlock / Conv3x3 with the upsample + concat of depthnet.py:79-85 in front).  Three kernels, one
// structure: thin_wgrad_kernel, thin_fwd_kernel, thin_dgrad_kernel.  First the weight gradient:
//
//   dW[co][ky][kx][ci] = sum_{b,y,x} dZ[b,y,x,co] * X[b, refl(y+ky-1), refl(x+kx-1), ci],   dZ = dY * ELU'(Y)
//
// As a GEMM this is M = Cout (16 / 32) x N = 9 Cin (144 ... 864) with millions of pixels to reduce over.  The generic
// split-K kernel (conv_wgrad.hip) stages a [pixel][32 ... 128 k] im2col slice per step: with Cout = 16 half of every
// 32x32 MFMA tile is padding, each input pixel is gathered 9 times, and a stage holds 16 MFMAs per wave -- too few to
// hide its own loads (17 ... 51 TF).  Here one workgroup keeps the WHOLE dW of the layer in accumulators and walks
// down a column of output-row segments:
//   * a ring of 4 input rows (segment + halo, reflection / nearest-upsample / concat resolved while staging) lives
//     in LDS as [row][pixel][channel]; a new output row costs ONE new input row, the 9 taps are just address
//     offsets of the B-operand ds_read_b32 (lanes run along the channels: conflict-free, pixel stride = 16 mod 32
//     banks for the 16-wide form);
//   * Cout = 16 layers run on v_mfma_f32_16x16x4_f32 (same peak rate as 32x32x2, no padded rows): the 4 waves split
//     the pixels of a stage and each holds all 9 Cin/16 tiles; Cout = 32 layers run on 32x32x2 with the tiles dealt
//     to the waves;
//   * the activation derivative and the bias gradient ride on the (small) dY staging.
// Partial dW of the workgroups are combined with fp32 atomics into the gradient arena, as in conv_wgrad.hip.
#include "conv_common.h"

#include <cstdlib>
#include <type_traits>

#ifndef THIN_DBG          // timing cuts of the row loops (tools/build_variant.py --flag=-DTHIN_DBG=<bits>; wrong results): 1 no global loads,
#define THIN_DBG 0        // 2 no MFMAs, 4 no LDS stores, 8 no barrier, 16 no output stores (data gradient)
#endif

namespace dvsconv {
namespace {

struct ThinParams {
    const float* x;      // [B,H,W,C1], or [B,H/2,W/2,C1] when up
    const float* x2;     // [B,H,W,C2] skip tensor (channels C1 ...), or NULL
    const float* dy;     // [B,H,W,Cout]
    const float* y;      // forward output (activation derivative), or NULL
    float* dw;           // [Cout][9][Cin]
    float* dbias;        // [Cout] or NULL
    int B, H, W, C1, C2, up, dact;
    int nseg, rows_per_wg, row_chunks;
};

template <int MT>
struct Acc {
    using type = typename std::conditional<MT == 32, f32x16, f32x4>::type;
};

template <int I, int N, class F>
__device__ __forceinline__ void static_for_thin(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for_thin<I + 1, N>(f);
    }
}

__device__ __forceinline__ f32x16 mma(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mma(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// MT = Cout = MFMA tile edge (16: 16x16x4, 32: 32x32x2); CIN = C1 + C2; SEG = output pixels of a stage (one row segment)
template <int MT, int CIN, int SEG>
__global__ __launch_bounds__(NT) void thin_wgrad_kernel(ThinParams p) {
    constexpr int KS = 64 / MT;                          // pixels (k) per MFMA
    constexpr int CT = CIN / MT;                         // channel tiles per tap
    constexpr int NTILE = 9 * CT;                        // (tap, channel tile) tiles of dW
    constexpr int PWAVES = (MT == 16) ? 4 : 1;           // waves along the pixels of a stage
    constexpr int TWAVES = 4 / PWAVES;                   // waves along the tiles
    constexpr int TPW = (NTILE + TWAVES - 1) / TWAVES;   // tiles per wave
    constexpr int PXW = SEG / PWAVES;                    // pixels of a stage per wave
    constexpr int STEPS = PXW / KS;
    constexpr int CS = (MT == 16 && CIN % 32 == 0) ? CIN + 16 : CIN;   // LDS pixel stride: 16 mod 32 for the 16-lane groups
    constexpr int COLS = SEG + 2;
    constexpr int ROWF = COLS * CS;
    constexpr int XV = CIN / 4, ROW_VECS = COLS * XV, X_LOADS = (ROW_VECS + NT - 1) / NT;
    constexpr int DV = MT / 4, D_VECS = SEG * DV, D_LOADS = (D_VECS + NT - 1) / NT;
    static_assert(CIN % MT == 0 && SEG % (PWAVES * KS) == 0 && D_VECS % 64 == 0, "shape");
    using acc_t = typename Acc<MT>::type;
    constexpr int NR = (MT == 32) ? 16 : 4;

    __shared__ __attribute__((aligned(16))) float Ps[4 * ROWF];
    __shared__ __attribute__((aligned(16))) float Ds[2][SEG][MT];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pw = wave / TWAVES, tw = wave % TWAVES;
    int lg = xcd_logical(blockIdx.x, gridDim.x, 1);
    const int chunk = lg % p.row_chunks;
    lg /= p.row_chunks;
    const int seg = lg % p.nseg, b = lg / p.nseg;
    const int oy_begin = chunk * p.rows_per_wg, oy_end = min(p.H, oy_begin + p.rows_per_wg);
    const int x0 = seg * SEG;
    const int Cout = MT, Ktot = 9 * CIN;
    const int H1 = p.up ? p.H >> 1 : p.H, W1 = p.up ? p.W >> 1 : p.W;

    // ---- input-row staging: my 16-byte vectors of a (segment + halo) row; everything but the row is loop-invariant
    int x_goff[X_LOADS], x_loff[X_LOADS];
    bool x_ok[X_LOADS], x_is2[X_LOADS];
#pragma unroll
    for (int j = 0; j < X_LOADS; ++j) {
        const int idx = tid + NT * j;
        x_ok[j] = idx < ROW_VECS;
        const int px = min(idx, ROW_VECS - 1) / XV, c = (min(idx, ROW_VECS - 1) % XV) * 4;
        const int sx = reflect_i(x0 - 1 + px, p.W);
        x_is2[j] = c >= p.C1;
        x_goff[j] = x_is2[j] ? sx * p.C2 + (c - p.C1) : (p.up ? sx >> 1 : sx) * p.C1 + c;
        x_loff[j] = px * CS + c;
    }
    // NS register sets: a row is loaded NS stages before the stage that stores it into the ring (a stage is ~1 us of MFMAs, an HBM
    // round trip under load several: with one set every stage ended waiting for the row it had just asked for)
#ifdef THIN_NS
    constexpr int NS = THIN_NS;
#else
    constexpr int NS = 2;                                // (three and four sets measured the same; they cost a workgroup per CU)
#endif
    f32x4 rx[NS > 3 ? NS : 3][X_LOADS];                  // (three sets for the prologue's three rows; the loop uses the first NS)
    auto load_row = [&](int pr, auto set) {              // padded row pr = source row refl(pr - 1)
        constexpr int S = decltype(set)::value;
        const int sr = reflect_i(min(pr, p.H + 1) - 1, p.H);
        const float* r1 = p.x + ((size_t)b * H1 + (p.up ? sr >> 1 : sr)) * W1 * p.C1;
        const float* r2 = p.C2 > 0 ? p.x2 + ((size_t)b * p.H + sr) * p.W * p.C2 : r1;
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j) rx[S][j] = *reinterpret_cast<const f32x4*>((x_is2[j] ? r2 : r1) + x_goff[j]);
    };
    auto store_row = [&](int pr, auto set) {
        constexpr int S = decltype(set)::value;
        float* dst = Ps + (pr & 3) * ROWF;
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j)
            if (x_ok[j]) *reinterpret_cast<f32x4*>(dst + x_loff[j]) = rx[S][j];
    };

    // ---- dY staging (x activation derivative) + bias gradient
    f32x4 rd[NS][D_LOADS], ry[NS][D_LOADS], bsum = {0.f, 0.f, 0.f, 0.f};
    const int d_c = (tid % DV) * 4;
    const bool d_ok = tid < D_VECS;                       // wave-uniform (D_VECS is a multiple of 64)
    auto load_d = [&](int oy, auto set) {
        constexpr int S = decltype(set)::value;
        if (d_ok) {
#pragma unroll
            for (int j = 0; j < D_LOADS; ++j) {
                const size_t o = (((size_t)b * p.H + min(oy, p.H - 1)) * p.W + x0 + (tid + NT * j) / DV) * Cout + d_c;
                rd[S][j] = *reinterpret_cast<const f32x4*>(p.dy + o);
                if (p.dact) ry[S][j] = *reinterpret_cast<const f32x4*>(p.y + o);
            }
        }
    };
    auto store_d = [&](int buf, auto set) {
        constexpr int S = decltype(set)::value;
        if (d_ok) {
#pragma unroll
            for (int j = 0; j < D_LOADS; ++j) {
                f32x4 v = rd[S][j];
                if (p.dact == ACT_ELU) {                  // 1 + min(y, 0)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], fminf(ry[S][j][e], 0.f), v[e]);
                } else if (p.dact) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= act_grad_from_out(ry[S][j][e], p.dact);
                }
                *reinterpret_cast<f32x4*>(&Ds[buf][(tid + NT * j) / DV][d_c]) = v;
                bsum += v;
            }
        }
    };

    // ---- my tiles: q = tw + TWAVES * i -> (tap, channel tile); a wave past the last tile recomputes tile 0 and drops it
    const int col = lane % MT, kidx = lane / MT;
    int t_ky[TPW], t_off[TPW];
    bool t_ok[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int q = tw + TWAVES * i;
        t_ok[i] = q < NTILE;
        const int qq = t_ok[i] ? q : 0;
        const int tap = qq / CT, ct = qq % CT;
        t_ky[i] = tap / 3;
        t_off[i] = (tap % 3) * CS + ct * MT;
    }
    const int b_lane = (pw * PXW + kidx) * CS + col;      // + tile offset + row slot + t * KS * CS
    const int a_lane = (pw * PXW + kidx) * MT + col;      // + t * KS * MT

    acc_t acc[TPW];
#pragma unroll
    for (int i = 0; i < TPW; ++i)
#pragma unroll
        for (int r = 0; r < NR; ++r) acc[i][r] = 0.f;

    using S0 = std::integral_constant<int, 0>;
    if (oy_begin < oy_end) {
        // the three rows of the first window and the first dY row: all loads first (one HBM round trip instead of four -- every
        // workgroup of the one resident round is here at the same time, so nothing else covers it)
        static_for_thin<0, 3>([&](auto k) { load_row(oy_begin + k.value, k); });
        load_d(oy_begin, S0{});
        static_for_thin<0, 3>([&](auto k) { store_row(oy_begin + k.value, k); });
        store_d(0, S0{});
        static_for_thin<0, NS - 1>([&](auto k) {          // set k: the row stage oy_begin + k stores
            load_row(oy_begin + 3 + k.value, k);
            load_d(oy_begin + 1 + k.value, k);
        });
        // the ring slot and the dY buffer that a one- or two-row chunk never writes are still read (times zero) by the dead stages
        for (int i = tid; i < ROWF; i += NT) Ps[((oy_begin + 3) & 3) * ROWF + i] = 0.f;
        for (int i = tid; i < SEG * MT; i += NT) (&Ds[1][0][0])[i] = 0.f;
    }
    __syncthreads();
    int buf = 0;
    // stage oy with register set S = (oy - oy_begin) % NS: ask for the row / dY row that stage oy + NS - 1 will store (into the set
    // the previous stage emptied), multiply, store what set S holds (asked for NS - 1 stages ago) for stage oy + 1.  Rows past the
    // end of the chunk are never asked for or stored; the dead stages that fill the last group multiply zeros.
    auto stage = [&](int oy, auto set) {
        constexpr int S = decltype(set)::value;
        using Free = std::integral_constant<int, (S + NS - 1) % NS>;
        const bool more = oy + 1 < oy_end;
        if (oy + NS < oy_end && !(THIN_DBG & 1)) {
            load_row(oy + 2 + NS, Free{});                // global loads stay in flight across NS - 1 stages of MFMAs
            load_d(oy + NS, Free{});
        }
        const float* Ab = &Ds[buf][0][0] + a_lane;
        const float* Bb[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) Bb[i] = Ps + ((oy + t_ky[i]) & 3) * ROWF + t_off[i] + b_lane;
        const float live = oy < oy_end ? 1.f : 0.f;
        // operands of k-step t + 1 are read before the MFMAs of step t, and the order is pinned: left alone the compiler puts every
        // ds_read right in front of the two MFMAs it feeds and waits lgkmcnt(0) there -- one LDS round trip per 64 cycles of matrix
        // work (the 16-channel forms ran at 0.36 of the MFMA rate for it)
        float aq[2], bq[2][TPW];
        aq[0] = Ab[0];
#pragma unroll
        for (int i = 0; i < TPW; ++i) bq[0][i] = Bb[i][0];
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
            if (t + 1 < STEPS) {
                aq[(t + 1) & 1] = Ab[(t + 1) * KS * MT];
#pragma unroll
                for (int i = 0; i < TPW; ++i) bq[(t + 1) & 1][i] = Bb[i][(t + 1) * KS * CS];
            }
            __builtin_amdgcn_sched_barrier(0);
            const float a = aq[t & 1] * live;
            if constexpr (THIN_DBG & 2) {
#pragma unroll
                for (int i = 0; i < TPW; ++i) acc[i][0] += a * bq[t & 1][i];
            } else {
#pragma unroll
                for (int i = 0; i < TPW; ++i) acc[i] = mma(a, bq[t & 1][i], acc[i]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more && !(THIN_DBG & 4)) {
            store_row(oy + 3, set);                       // slot (oy - 1) & 3: not read by this stage
            store_d(buf ^ 1, set);
        }
        if (!(THIN_DBG & 8)) __syncthreads();
        buf ^= 1;
    };
    // (stages come in groups of NS, one per register set, without a branch between them: a conditional stage made the compiler keep
    // two copies of the accumulators)
#pragma unroll 1
    for (int oy = oy_begin; oy < oy_end; oy += NS) static_for_thin<0, NS>([&](auto k) { stage(oy + k.value, k); });

    // ---- epilogue: dW[co][tap][ci] += acc.  C/D maps: 32x32: co = (r&3) + 8 (r>>2) + 4 (lane>>5), ci = lane & 31;
    //      16x16: co = 4 (lane>>4) + r, ci = lane & 15
    if constexpr (PWAVES > 1) {
        // The waves of the 16-channel forms split the PIXELS: each holds a partial of the whole dW.  Added to the arena wave by wave
        // that was 4 x 480 same-address atomics per weight (~85 ns each: 160 us, more than the kernel's arithmetic -- with every
        // phase of the row loop cut out it still took 195 us); the waves now add into one LDS copy first (the ring is free: the row
        // loop ended on a barrier) and the workgroup adds that once.
        static_assert(Cout * 9 * CIN <= 4 * ROWF, "dW fits the ring");
        float* sw = Ps;
        for (int i = tid; i < Cout * Ktot; i += NT) sw[i] = 0.f;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if (!t_ok[i]) continue;
            const int q = tw + TWAVES * i;
            const int kk = (q / CT) * CIN + (q % CT) * MT + col;
#pragma unroll
            for (int r = 0; r < NR; ++r) atomicAdd(sw + (4 * (lane >> 4) + r) * Ktot + kk, acc[i][r]);
        }
        __syncthreads();
        // every workgroup finishes at the same time: each starts its pass over dW somewhere else, so that the atomics in flight at any
        // moment are spread over the addresses instead of queueing on the same ones
        constexpr int NW = Cout * 9 * CIN;
        const int rot = (int)((blockIdx.x * 7u) % (unsigned)(NW / NT)) * NT;
        for (int i = tid; i < NW; i += NT) {
            const int j = i + rot < NW ? i + rot : i + rot - NW;
            atomicAdd(p.dw + j, sw[j]);
        }
    } else {
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            if (!t_ok[i]) continue;
            const int q = tw + TWAVES * i;
            const int kk = (q / CT) * CIN + (q % CT) * MT + col;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int co = (MT == 32) ? (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) : 4 * (lane >> 4) + r;
                atomicAdd(p.dw + (size_t)co * Ktot + kk, acc[i][r]);
            }
        }
    }
    if (p.dbias) {
        float* sb = &Ds[0][0][0];                         // the row loop ended on a barrier
        if (tid < MT) sb[tid] = 0.f;
        __syncthreads();
        if (d_ok) {
#pragma unroll
            for (int e = 0; e < 4; ++e) atomicAdd(&sb[d_c + e], bsum[e]);
        }
        __syncthreads();
        if (tid < MT) atomicAdd(p.dbias + tid, sb[tid]);
    }
}

template <int MT, int CIN, int SEG>
void launch_thin(ThinParams p, hipStream_t st) {
    p.nseg = p.W / SEG;
    // one resident round of equal workgroups: (image, segment) columns x row chunks <= CUs x workgroups per CU.  Few, long
    // workgroups also keep the final atomics (a whole dW each) a small part of the work.
    static const int slots = [] {
        int occ = 0, dev = 0, cus = 256;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, thin_wgrad_kernel<MT, CIN, SEG>, NT, 0) != hipSuccess || occ < 1) occ = 1;
        return cus * occ;
    }();
    const int cols = p.B * p.nseg;
    int chunks = max(1, min(slots / cols, p.H / 8));
    p.rows_per_wg = (p.H + chunks - 1) / chunks;
    p.row_chunks = (p.H + p.rows_per_wg - 1) / p.rows_per_wg;
    dvs::ProfScope prof(dvs::SLOT_CONV_WGRAD, st);
    prof.work(2.0 * p.B * p.H * p.W * MT * 9.0 * CIN);
    hipLaunchKernelGGL((thin_wgrad_kernel<MT, CIN, SEG>), dim3(cols * p.row_chunks), dim3(NT), 0, st, p);
}


// ---------------------------------------------------------------------------------------------------------------------
// Forward of the 16-output-channel layers (upconv_0_0: 32 -> 16 at 240x320, upconv_0_1: upsample(16) -> 16 at 480x640).
// N = 16 fills half of the generic kernel's narrowest (32-column) tile and a 16- / 32-channel K stage cannot use the
// LDS-DMA path, so those two layers ran at 32 / 41 TF.  Same row ring as the weight gradient above, roles swapped:
// A = pixels x (tap, channel) read from the ring (16 pixel lanes at stride CIN + 2 floats: conflict-free), B = the
// layer's whole weight matrix held in registers (9 CIN / 4 values per lane), v_mfma_f32_16x16x4_f32, bias +
// activation on the accumulators, direct NHWC store.
struct ThinFwdParams {
    const float* x;      // [B,H,W,CIN], or [B,H/2,W/2,CIN] when up
    const float* w;      // [16][9][CIN]
    const float* bias;   // [16] or NULL
    float* y;            // [B,H,W,16]
    int B, H, W, up, act;
    int nseg, rows_per_wg, row_chunks;
};

__device__ __forceinline__ float thin_act(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_ELU: return v > 0.f ? v : __expf(v) - 1.f;     // absolute error ~1e-7 (one v_exp_f32 instead of expm1f's polynomial)
        case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        default: return v;
    }
}

template <int CIN, int SEG>
__global__ __launch_bounds__(NT) void thin_fwd_kernel(ThinFwdParams p) {
    constexpr int CO = 16;
    constexpr int CS = CIN + 2;                          // pixel stride: 16 pixel lanes x 2 channel groups = 32 distinct banks
    constexpr int COLS = SEG + 2, ROWF = COLS * CS;
    constexpr int XV = CIN / 4, ROW_VECS = COLS * XV, X_LOADS = (ROW_VECS + NT - 1) / NT;
    constexpr int PXW = SEG / 4, PT = PXW / 16;          // pixels / 16-pixel tiles of a stage per wave
    constexpr int C4 = CIN / 4, KSTEPS = 9 * C4;
    static_assert(PXW % 16 == 0 && (ROWF % 2) == 0, "shape");
    __shared__ __attribute__((aligned(16))) float Ps[4 * ROWF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lg = xcd_logical(blockIdx.x, gridDim.x, 1);
    const int chunk = lg % p.row_chunks;
    lg /= p.row_chunks;
    const int seg = lg % p.nseg, b = lg / p.nseg;
    const int oy_begin = chunk * p.rows_per_wg, oy_end = min(p.H, oy_begin + p.rows_per_wg);
    const int x0 = seg * SEG;
    const int H1 = p.up ? p.H >> 1 : p.H, W1 = p.up ? p.W >> 1 : p.W;

    int x_goff[X_LOADS], x_loff[X_LOADS];
    bool x_ok[X_LOADS];
#pragma unroll
    for (int j = 0; j < X_LOADS; ++j) {
        const int idx = tid + NT * j;
        x_ok[j] = idx < ROW_VECS;
        const int px = min(idx, ROW_VECS - 1) / XV, c = (min(idx, ROW_VECS - 1) % XV) * 4;
        const int sx = reflect_i(x0 - 1 + px, p.W);
        x_goff[j] = (p.up ? sx >> 1 : sx) * CIN + c;
        x_loff[j] = px * CS + c;
    }
    f32x4 rx[X_LOADS];
    auto load_row_to = [&](int pr, f32x4 (&r)[X_LOADS]) {
        const int sr = reflect_i(pr - 1, p.H);
        const float* r1 = p.x + ((size_t)b * H1 + (p.up ? sr >> 1 : sr)) * W1 * CIN;
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j) r[j] = *reinterpret_cast<const f32x4*>(r1 + x_goff[j]);
    };
    auto store_row_from = [&](int pr, const f32x4 (&r)[X_LOADS]) {
        float* dst = Ps + (pr & 3) * ROWF;
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j)
            if (x_ok[j]) {                               // CS is even, not a multiple of 4: two 8-byte stores
                float2* d2 = reinterpret_cast<float2*>(dst + x_loff[j]);
                d2[0] = float2{r[j][0], r[j][1]};
                d2[1] = float2{r[j][2], r[j][3]};
            }
    };
    auto load_row = [&](int pr) { load_row_to(pr, rx); };
    auto store_row = [&](int pr) { store_row_from(pr, rx); };

    // B operand: lane (n = lane & 15, kidx = lane >> 4) holds W[n][tap][4 c4 + kidx] for every k-step (tap, c4)
    const int n = lane & 15, kidx = lane >> 4;
    float wreg[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) wreg[s] = p.w[(size_t)n * (9 * CIN) + (s / C4) * CIN + (s % C4) * 4 + kidx];
    const float bv = p.bias ? p.bias[n] : 0.f;
    const int a_lane = (wave * PXW + n) * CS + kidx;     // pixel lane & 15 of my first tile, channel kidx

    if (oy_begin < oy_end) {
        // the three rows of the first window: all loads first (one HBM round trip, not three -- every workgroup of the one resident
        // round starts here at the same time, so nothing else covers it)
        f32x4 r0[X_LOADS], r1[X_LOADS];
        load_row_to(oy_begin, r0);
        load_row_to(oy_begin + 1, r1);
        load_row(oy_begin + 2);
        store_row_from(oy_begin, r0);
        store_row_from(oy_begin + 1, r1);
        store_row(oy_begin + 2);
    }
    __syncthreads();
#pragma unroll 1
    for (int oy = oy_begin; oy < oy_end; ++oy) {
        const bool more = oy + 1 < oy_end;
        if (more) load_row(oy + 3);
        const float* Ar[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) Ar[ky] = Ps + ((oy + ky) & 3) * ROWF + a_lane;
        f32x4 acc[PT];
#pragma unroll
        for (int t = 0; t < PT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < KSTEPS; ++s) {
            const int tap = s / C4, c4 = s % C4, ky = tap / 3, kx = tap % 3;
#pragma unroll
            for (int t = 0; t < PT; ++t) acc[t] = mma(Ar[ky][(t * 16 + kx) * CS + c4 * 4], wreg[s], acc[t]);
        }
        // C/D map: pixel = 4 (lane >> 4) + r, channel = lane & 15
        float* yrow = p.y + (((size_t)b * p.H + oy) * p.W + x0 + wave * PXW + 4 * kidx) * CO + n;
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) yrow[(t * 16 + r) * CO] = thin_act(acc[t][r] + bv, p.act);
        if (more) store_row(oy + 3);
        __syncthreads();
    }
}

template <int CIN, int SEG>
void launch_thin_fwd(ThinFwdParams p, hipStream_t st) {
    p.nseg = p.W / SEG;
    static const int slots = [] {
        int occ = 0, dev = 0, cus = 256;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, thin_fwd_kernel<CIN, SEG>, NT, 0) != hipSuccess || occ < 1) occ = 1;
        return cus * occ;
    }();
    const int cols = p.B * p.nseg;
    int chunks = max(1, min(slots / cols, p.H / 8));
    p.rows_per_wg = (p.H + chunks - 1) / chunks;
    p.row_chunks = (p.H + p.rows_per_wg - 1) / p.rows_per_wg;
    dvs::ProfScope prof(dvs::SLOT_CONV_FWD, st);
    prof.work(2.0 * p.B * p.H * p.W * 16 * 9.0 * CIN);
    hipLaunchKernelGGL((thin_fwd_kernel<CIN, SEG>), dim3(cols * p.row_chunks), dim3(NT), 0, st, p);
}


// ---------------------------------------------------------------------------------------------------------------------
// Data gradient of the same two layers (dY has 16 channels; dX has 32 channels at 240x320 / 16 channels summed 2x2 into
// the 240x320 tensor that was upsampled).  The generic kernel multiplied dY by ELU'(Y) nine times per element, padded N
// to 32 and, for upconv_0_1, reduced the 2x2 blocks with 29 M global atomics (22 TF).  Row ring again:
//   dX[y][x] = sum_{ky,kx} W[.][ky][kx][.]^T dZ[y + 1 - ky][x + 1 - kx]     (dZ = dY * act'(Y), zero outside the image)
//   + the reflection fold: what the mirrored border of the padded input received comes back to row 1 / H-2 and column
//     1 / W-2.  Those are EXTRA k-steps, not epilogue work: row 1 also multiplies dZ row 0 by the ky = 0 taps (row H-2:
//     dZ row H-1, ky = 2) -- extra k-steps of the whole stage, a uniform branch; column 1 also receives dZ column 0
//     through the kx = 0 taps (column W-2: column W-1, kx = 2) -- added to the A operand of that one pixel lane of one
//     edge tile (every other lane of the tile adds the ring's zero halo entry).
//   * dZ rows are staged once per workgroup (activation derivative applied there), weights live in registers;
//   * the 2x2 sum of the upsample gradient is two rows accumulated into the same accumulators plus an in-lane add of
//     register pairs (the 16x16 C/D map keeps 4 consecutive pixels in a lane): plain stores, no atomics.
struct ThinDgradParams {
    const float* dy;     // [B,H,W,CK]
    const float* y;      // forward output or NULL
    const float* wt;     // packed [nout][9][CK] (dvs_conv2d_pack_wt)
    float* dx;           // [B,H,W,nout] (C1 == 0), or the coarse [B,H/2,W/2,C1] tensor of an upsample(+concat) input
    float* dx_skip;      // [B,H,W,nout-C1]: the skip tensor's gradient (C1 < nout), or NULL
    int B, H, W, dact, nout, C1;
    int nseg, rows_per_wg, row_chunks, nsplit;
};

// CK = channels of dY (the forward's Cout: 16 / 32); a workgroup computes 16 NTN of the nout input channels (split index
// from the block id: weights for more than 32 channels do not fit a wave's registers, and dZ rows are cheap to re-stage);
// channels below C1 belong to the upsampled operand (2x2-summed), the rest to the skip tensor / the plain input.
template <int CK, int NTN, int SEG>
__device__ __forceinline__ void thin_dgrad_body(const ThinDgradParams& p) {
    constexpr int CS = CK + 2, C4 = CK / 4;
    constexpr int COLS = SEG + 2, ROWF = COLS * CS;
    constexpr int ROW_VECS = COLS * C4, X_LOADS = (ROW_VECS + NT - 1) / NT;
    constexpr int PXW = SEG / 4, PT = PXW / 16;
    static_assert(PXW % 16 == 0, "shape");
    __shared__ __attribute__((aligned(16))) float Ps[4 * ROWF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int lg = xcd_logical(blockIdx.x, gridDim.x, 1);
    const int split = lg % p.nsplit;                      // the channel groups of one pixel column are neighbours: shared dZ rows
    lg /= p.nsplit;
    const int chunk = lg % p.row_chunks;
    lg /= p.row_chunks;
    const int seg = lg % p.nseg, b = lg / p.nseg;
    const int iy_begin = chunk * p.rows_per_wg, iy_end = min(p.H, iy_begin + p.rows_per_wg);
    const int x0 = seg * SEG;
    const int n_base = split * 16 * NTN;
    bool coarse[NTN];                                     // per 16-channel tile, workgroup-uniform
    bool any_coarse = false, any_direct = false;
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt) {
        coarse[nt] = n_base + nt * 16 < p.C1;
        any_coarse = any_coarse || coarse[nt];
        any_direct = any_direct || !coarse[nt];
    }

    // ---- dZ row staging: entry e of a ring row is image column x0 - 1 + e; columns / rows outside the image are zeros
    int x_goff[X_LOADS], x_loff[X_LOADS];
    bool x_ok[X_LOADS], x_in[X_LOADS];
#pragma unroll
    for (int j = 0; j < X_LOADS; ++j) {
        const int idx = tid + NT * j;
        x_ok[j] = idx < ROW_VECS;
        const int e = min(idx, ROW_VECS - 1) / C4, c = (min(idx, ROW_VECS - 1) % C4) * 4;
        const int col = x0 - 1 + e;
        x_in[j] = (unsigned)col < (unsigned)p.W;
        x_goff[j] = clampi(col, p.W) * CK + c;
        x_loff[j] = e * CS + c;
    }
    // NS register sets: a dZ row is asked for NS stages before the stage that stores it into the ring (see thin_wgrad_kernel).  Two sets
    // only for the one-wave-per-SIMD form (upconv_1_1): the others hide the round trip behind their second and third workgroup, and
    // the second copy of the stage costs them one
    constexpr int NS = (NTN == 3) ? 2 : 1;
    f32x4 rd[3][X_LOADS], ry[3][X_LOADS];                // (three sets for the prologue's three rows; the loop uses the first NS)
    const bool elu = p.dact == ACT_ELU;
    auto load_row = [&](int d, auto set) {                // dZ row d (may be -1 or H: zeros)
        constexpr int S = decltype(set)::value;
        const size_t ro = ((size_t)b * p.H + clampi(d, p.H)) * p.W * CK;
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j) {
            rd[S][j] = *reinterpret_cast<const f32x4*>(p.dy + ro + x_goff[j]);
            if (p.dact) ry[S][j] = *reinterpret_cast<const f32x4*>(p.y + ro + x_goff[j]);
        }
    };
    auto store_row = [&](int d, auto set) {
        constexpr int S = decltype(set)::value;
        const bool row_in = (unsigned)d < (unsigned)p.H;
        float* dst = Ps + ((d + 1) & 3) * ROWF;
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j)
            if (x_ok[j]) {
                f32x4 v = rd[S][j];
                const bool in = row_in && x_in[j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // ELU'(y) = 1 + min(y, 0): two instructions, no branch (the generic form is a switch per element)
                    const float g = elu ? fmaf(v[e], fminf(ry[S][j][e], 0.f), v[e]) : (p.dact ? v[e] * act_grad_from_out(ry[S][j][e], p.dact) : v[e]);
                    v[e] = in ? g : 0.f;
                }
                float2* d2 = reinterpret_cast<float2*>(dst + x_loff[j]);
                d2[0] = float2{v[0], v[1]};
                d2[1] = float2{v[2], v[3]};
            }
    };

    // ---- B operand: lane (n = lane & 15 [+ 16 nt], kidx) holds W[co = 4 c4 + kidx][tap][ci] = wt[(ci * 9 + tap) * CK + co]
    const int n = lane & 15, kidx = lane >> 4;
    float wreg[NTN][9 * C4];
#pragma unroll
    for (int nt = 0; nt < NTN; ++nt)
#pragma unroll
        for (int s = 0; s < 9 * C4; ++s)
            wreg[nt][s] = p.wt[((size_t)(n_base + nt * 16 + n) * 9 + s / C4) * CK + (s % C4) * 4 + kidx];
    const int a_lane = (wave * PXW + n) * CS + kidx;       // + (16 t + 2 - kx) CS: entry of pixel (lane & 15) of tile t, tap kx
    // column fold: the one lane whose pixel is image column 1 (first segment, wave 0, tile 0) / W - 2 (last segment, last
    // wave, last tile) reads entry 1 (column 0) / SEG (column W - 1); the other lanes of that tile read the zero halo entry
    // the last segment may be narrower than SEG (W a multiple of 16 only: upconv_1_0's 160 columns = 64 + 64 + 32): its columns past
    // the image are staged as zeros, the waves past them store nothing, and the right-hand fold moves to the wave that owns column W - 1
    const int wseg = min(SEG, p.W - x0);
    const bool edge_l = seg == 0 && wave == 0, edge_r = seg == p.nseg - 1 && wave == wseg / PXW - 1;
    const bool wave_live = wave * PXW < wseg;
    const int fold_l = ((n == 1) ? 1 : 0) * CS + kidx;
    const int fold_r = ((n == 14) ? wseg : wseg + 1) * CS + kidx;

    f32x4 acc[PT][NTN];
    auto zero_acc = [&]() {
#pragma unroll
        for (int t = 0; t < PT; ++t)
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt) acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    zero_acc();

    using S0 = std::integral_constant<int, 0>;
    if (iy_begin < iy_end) {
        static_for_thin<0, 3>([&](auto k) { load_row(iy_begin - 1 + k.value, k); });           // one round trip for the first window
        static_for_thin<0, 3>([&](auto k) { store_row(iy_begin - 1 + k.value, k); });
        static_for_thin<0, NS - 1>([&](auto k) { load_row(iy_begin + 2 + k.value, k); });      // set k: the row stage iy_begin + k stores
    }
    __syncthreads();
    auto stage = [&](int iy, auto set) {
        constexpr int S = decltype(set)::value;
        using Free = std::integral_constant<int, (S + NS - 1) % NS>;
        const bool more = iy + 1 < iy_end;
        if (iy + NS < iy_end && !(THIN_DBG & 1)) load_row(iy + 1 + NS, Free{});      // in flight across NS stages (NS = 1: this one)
        // operands of one tap (all k-steps, all pixel tiles) / its MFMAs.  The LDS reads of tap i + 1 are issued before the
        // MFMAs of tap i and fenced there (sched_barrier): left alone, the scheduler sinks every ds_read next to its MFMA and
        // the wave pays one LDS round trip per k-step.
        auto lda = [&](float (&a)[PT][C4], const float* row, int kx) {
#pragma unroll
            for (int t = 0; t < PT; ++t)
#pragma unroll
                for (int c4 = 0; c4 < C4; ++c4) a[t][c4] = row[a_lane + (t * 16 + 2 - kx) * CS + c4 * 4];
            // column fold (edge tiles, wave-uniform): image column 1 also receives dZ column 0 through the kx = 0 taps, column
            // W - 2 receives column W - 1 through kx = 2 -- added to the OPERAND of that one pixel lane (the other lanes add
            // the ring's zero halo entry), not multiplied separately
            if (kx == 0 && edge_l) {
#pragma unroll
                for (int c4 = 0; c4 < C4; ++c4) a[0][c4] += row[fold_l + c4 * 4];
            }
            if (kx == 2 && edge_r) {
#pragma unroll
                for (int c4 = 0; c4 < C4; ++c4) a[PT - 1][c4] += row[fold_r + c4 * 4];
            }
        };
        auto mfmas = [&](const float (&a)[PT][C4], int tap) {
#pragma unroll
            for (int c4 = 0; c4 < C4; ++c4)
#pragma unroll
                for (int t = 0; t < PT; ++t)
#pragma unroll
                    for (int nt = 0; nt < NTN; ++nt) {
                        if constexpr (THIN_DBG & 2) acc[t][nt][0] += a[t][c4] * wreg[nt][tap * C4 + c4];
                        else acc[t][nt] = mma(a[t][c4], wreg[nt][tap * C4 + c4], acc[t][nt]);
                    }
        };
        auto taps = [&](const float* row, int ky) {       // the three kx taps of weight row ky on one dZ row (fold rows only)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                float a[PT][C4];
                lda(a, row, kx);
                mfmas(a, ky * 3 + kx);
            }
        };
        const float* rows[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) rows[ky] = Ps + ((iy + 2 - ky) & 3) * ROWF;   // dZ row iy + 1 - ky lives in slot (d + 1) & 3
        float abuf[2][PT][C4];
        lda(abuf[0], rows[0], 0);
        static_for_thin<0, 9>([&](auto ic) {
            constexpr int I = decltype(ic)::value;
            if constexpr (I + 1 < 9) lda(abuf[(I + 1) & 1], rows[(I + 1) / 3], (I + 1) % 3);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(abuf[I & 1], I);
            __builtin_amdgcn_sched_barrier(0);
        });
        if (iy == 1) {                                    // + what padded row 0 received: dZ row 0 through the ky = 0 taps
            const float* row = Ps + 1 * ROWF;
            taps(row, 0);
        }
        if (iy == p.H - 2) {                              // + padded row H + 1: dZ row H - 1 through the ky = 2 taps
            const float* row = Ps + (p.H & 3) * ROWF;
            taps(row, 2);
        }
        // C/D map: pixel = 4 (lane >> 4) + r, channel = lane & 15
        if (any_coarse && (iy & 1) && wave_live && !(THIN_DBG & 16)) { // rows iy - 1 and iy are in the accumulators: 2x2 sums, one coarse row
            float* orow = p.dx + (((size_t)b * (p.H >> 1) + (iy >> 1)) * (p.W >> 1) + ((x0 + wave * PXW + 4 * kidx) >> 1)) * p.C1 +
                          n_base + n;
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
                if (coarse[nt]) {
#pragma unroll
                    for (int t = 0; t < PT; ++t) {
                        orow[(size_t)(t * 8 + 0) * p.C1 + nt * 16] = acc[t][nt][0] + acc[t][nt][1];
                        orow[(size_t)(t * 8 + 1) * p.C1 + nt * 16] = acc[t][nt][2] + acc[t][nt][3];
                        acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
        }
        if (any_direct && wave_live && !(THIN_DBG & 16)) {
            const int cs = p.nout - p.C1;                 // channels of the tensor these tiles belong to (C1 == 0: the input itself)
            float* base = p.C1 > 0 ? p.dx_skip : p.dx;
            float* orow = base + (((size_t)b * p.H + iy) * p.W + x0 + wave * PXW + 4 * kidx) * cs + (n_base - p.C1) + n;
#pragma unroll
            for (int nt = 0; nt < NTN; ++nt)
                if (!coarse[nt]) {
#pragma unroll
                    for (int t = 0; t < PT; ++t) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) orow[(size_t)(t * 16 + r) * cs + nt * 16] = acc[t][nt][r];
                        acc[t][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
                    }
                }
        }
        if (more && !(THIN_DBG & 4)) store_row(iy + 2, set);
        if (!(THIN_DBG & 8)) __syncthreads();
    };
#pragma unroll 1
    for (int iy = iy_begin; iy < iy_end; iy += NS)
        static_for_thin<0, NS>([&](auto k) {
            if (k.value == 0 || iy + k.value < iy_end) stage(iy + k.value, k);
        });
}

template <int CK, int NTN, int SEG>
__global__ __launch_bounds__(NT) void thin_dgrad_kernel(ThinDgradParams p) {
    thin_dgrad_body<CK, NTN, SEG>(p);
}
// 48 input channels per workgroup keep 216 weight values per lane: one wave per SIMD with the whole register file (the
// register allocator otherwise squeezes under 256 VGPRs by re-reading every LDS operand right before its MFMA)
template <int CK, int NTN, int SEG>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 1))) void thin_dgrad_kernel_w1(ThinDgradParams p) {
    thin_dgrad_body<CK, NTN, SEG>(p);
}

template <int CK, int NTN, int SEG, bool W1 = false>
void launch_thin_dgrad(ThinDgradParams p, hipStream_t st) {
    auto kern = W1 ? thin_dgrad_kernel_w1<CK, NTN, SEG> : thin_dgrad_kernel<CK, NTN, SEG>;
    p.nseg = (p.W + SEG - 1) / SEG;
    p.nsplit = p.nout / (16 * NTN);
    static const int slots = [] {
        int occ = 0, dev = 0, cus = 256;
        hipGetDevice(&dev);
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        auto k = W1 ? thin_dgrad_kernel_w1<CK, NTN, SEG> : thin_dgrad_kernel<CK, NTN, SEG>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, NT, 0) != hipSuccess || occ < 1) occ = 1;
        return cus * occ;
    }();
    const int cols = p.B * p.nseg * p.nsplit;
    int chunks = max(1, min(slots / cols, p.H / 8));
    p.rows_per_wg = ((p.H + chunks - 1) / chunks + 1) & ~1;   // even: a 2x2 block never straddles two workgroups
    p.row_chunks = (p.H + p.rows_per_wg - 1) / p.rows_per_wg;
    dvs::ProfScope prof(dvs::SLOT_CONV_DGRAD, st);
    prof.work(2.0 * p.B * p.H * p.W * CK * 9.0 * p.nout);
    hipLaunchKernelGGL(kern, dim3(cols * p.row_chunks), dim3(NT), 0, st, p);
}

}  // namespace

// the shapes thin_wgrad takes (it launches nothing for any other)
bool thin_wgrad_shape(const ConvShape& s, const InXform& t) {
    static const bool enabled = [] { const char* e = getenv("DVS_CONV_THIN"); return !(e && e[0] == '0'); }();
    if (!enabled || s.kh != 3 || s.kw != 3 || s.stride != 1 || s.pad != 1 || s.pad_mode != PAD_REFLECT || t.in_scale) return false;
    if (s.H < 8 || (t.x2 && ((s.H | s.W) & 1))) return false;
    const int C1 = t.x2 != nullptr ? t.C1 : s.Cin;
    if (C1 <= 0 || (C1 & 3) || s.Cin - C1 < 0) return false;
    return (s.Cout == 32 && s.Cin == 96 && s.W % 32 == 0) || (s.Cout == 32 && s.Cin == 64 && s.W % 32 == 0) ||
           (s.Cout == 16 && s.Cin == 32 && s.W % 64 == 0) || (s.Cout == 16 && s.Cin == 16 && s.W % 128 == 0);
}

bool thin_wgrad(const float* x, const float* dy, float* dw, float* dbias, const ConvShape& s, const InXform& t,
                hipStream_t st) {
    if (!thin_wgrad_shape(s, t)) return false;
    ThinParams p{};
    p.x = x; p.dy = dy; p.y = t.aux; p.dw = dw; p.dbias = dbias;
    p.B = s.B; p.H = s.H; p.W = s.W; p.dact = t.dact;
    p.up = t.x2 != nullptr;
    p.C1 = p.up ? t.C1 : s.Cin;
    p.C2 = s.Cin - p.C1;
    p.x2 = p.C2 > 0 ? t.x2 : nullptr;
    if (p.C1 <= 0 || (p.C1 & 3) || p.C2 < 0) return false;
    if (s.Cout == 32 && s.Cin == 96 && s.W % 32 == 0) launch_thin<32, 96, 32>(p, st);
    else if (s.Cout == 32 && s.Cin == 64 && s.W % 32 == 0) launch_thin<32, 64, 32>(p, st);
    else if (s.Cout == 16 && s.Cin == 32 && s.W % 64 == 0) launch_thin<16, 32, 64>(p, st);
    else if (s.Cout == 16 && s.Cin == 16 && s.W % 128 == 0) launch_thin<16, 16, 128>(p, st);
    else return false;
    return true;
}

}  // namespace dvsconv

namespace dvsconv {

bool thin_fwd(const float* x, const float* w, const float* bias, float* y, const ConvShape& s, const InXform& t, int act,
              hipStream_t st) {
    static const bool enabled = [] { const char* e = getenv("DVS_CONV_THIN"); return !(e && e[0] == '0'); }();
    if (!enabled || s.kh != 3 || s.kw != 3 || s.stride != 1 || s.pad != 1 || s.pad_mode != PAD_REFLECT || t.in_scale) return false;
    if (s.Cout != 16 || s.H < 8 || (t.x2 && (t.C1 != s.Cin || ((s.H | s.W) & 1)))) return false;     // upsample-only or plain
    ThinFwdParams p{};
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    p.B = s.B; p.H = s.H; p.W = s.W; p.up = t.x2 != nullptr; p.act = act;
    if (s.Cin == 16 && s.W % 128 == 0) launch_thin_fwd<16, 128>(p, st);
    else if (s.Cin == 32 && s.W % 64 == 0) launch_thin_fwd<32, 64>(p, st);
    else return false;
    return true;
}

}  // namespace dvsconv

namespace dvsconv {

// arguments as dvs_conv2d_dgrad receives them: H, W, Cin = the forward conv's (logical, full-resolution) input
bool thin_dgrad(const float* dy, const float* wt, float* dx, const float* y_out, int dact, int B, int H, int W, int Cin,
                int Cout, int split_c1, float* dx_skip, hipStream_t st) {
    static const bool enabled = [] { const char* e = getenv("DVS_CONV_THIN"); return !(e && e[0] == '0'); }();
    if (!enabled || H < 8) return false;
    if (split_c1 > 0 && (((H | W) & 1) || (split_c1 < Cin && !dx_skip))) return false;
    ThinDgradParams p{};
    p.dy = dy; p.y = y_out; p.wt = wt; p.dx = dx; p.dx_skip = dx_skip; p.B = B; p.H = H; p.W = W; p.dact = dact;
    p.nout = Cin; p.C1 = split_c1;
    if (Cout == 16 && Cin == 16 && split_c1 == 16 && W % 128 == 0) launch_thin_dgrad<16, 1, 128>(p, st);            // upconv_0_1
    else if (Cout == 16 && Cin == 32 && split_c1 == 0 && W % 64 == 0) launch_thin_dgrad<16, 2, 64>(p, st);          // upconv_0_0
    else if (Cout == 32 && Cin % 48 == 0 && Cin <= 192 && split_c1 % 16 == 0 && W % 64 == 0)
        launch_thin_dgrad<32, 3, 64, true>(p, st);   // upconv_1_1 (the two-tile form at two workgroups per CU: 696 against 565 us)
    else if (Cout == 32 && Cin % 32 == 0 && Cin <= 128 && split_c1 % 16 == 0 && (W % 64 == 0 || (W % 16 == 0 && W > 64)))
        launch_thin_dgrad<32, 2, 64>(p, st);          // (upconv_1_0: 160 columns, a ragged last segment)
    else return false;
    return true;
}

}  // namespace dvsconv
