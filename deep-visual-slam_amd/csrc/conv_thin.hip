// a2 backward: weight gradient of the decoder's thin full-resolution layers (Cout = 16 / 32; upconv_1_0, _1_1, _0_0,
// _0_1 of model/depthnet.py:49-62 -- ReflectionPad2d(1) + 3x3 conv + ELU on 120x160 ... 480x640 maps).
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
    f32x4 rx[X_LOADS];
    auto load_row = [&](int pr) {                        // padded row pr = source row refl(pr - 1)
        const int sr = reflect_i(pr - 1, p.H);
        const float* r1 = p.x + ((size_t)b * H1 + (p.up ? sr >> 1 : sr)) * W1 * p.C1;
        const float* r2 = p.C2 > 0 ? p.x2 + ((size_t)b * p.H + sr) * p.W * p.C2 : r1;
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j) rx[j] = *reinterpret_cast<const f32x4*>((x_is2[j] ? r2 : r1) + x_goff[j]);
    };
    auto store_row = [&](int pr) {
        float* dst = Ps + (pr & 3) * ROWF;
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j)
            if (x_ok[j]) *reinterpret_cast<f32x4*>(dst + x_loff[j]) = rx[j];
    };

    // ---- dY staging (x activation derivative) + bias gradient
    f32x4 rd[D_LOADS], ry[D_LOADS], bsum = {0.f, 0.f, 0.f, 0.f};
    const int d_c = (tid % DV) * 4;
    const bool d_ok = tid < D_VECS;                       // wave-uniform (D_VECS is a multiple of 64)
    auto load_d = [&](int oy) {
        if (d_ok) {
#pragma unroll
            for (int j = 0; j < D_LOADS; ++j) {
                const size_t o = (((size_t)b * p.H + oy) * p.W + x0 + (tid + NT * j) / DV) * Cout + d_c;
                rd[j] = *reinterpret_cast<const f32x4*>(p.dy + o);
                if (p.dact) ry[j] = *reinterpret_cast<const f32x4*>(p.y + o);
            }
        }
    };
    auto store_d = [&](int buf) {
        if (d_ok) {
#pragma unroll
            for (int j = 0; j < D_LOADS; ++j) {
                f32x4 v = rd[j];
                if (p.dact) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= act_grad_from_out(ry[j][e], p.dact);
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

    if (oy_begin < oy_end) {
        for (int pr = oy_begin; pr < oy_begin + 3; ++pr) {
            load_row(pr);
            store_row(pr);
        }
        load_d(oy_begin);
        store_d(0);
    }
    __syncthreads();
    int buf = 0;
#pragma unroll 1
    for (int oy = oy_begin; oy < oy_end; ++oy) {
        const bool more = oy + 1 < oy_end;
        if (more) {
            load_row(oy + 3);                             // global loads stay in flight across the MFMAs
            load_d(oy + 1);
        }
        const float* Ab = &Ds[buf][0][0] + a_lane;
        const float* Bb[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) Bb[i] = Ps + ((oy + t_ky[i]) & 3) * ROWF + t_off[i] + b_lane;
#pragma unroll
        for (int t = 0; t < STEPS; ++t) {
            const float a = Ab[t * KS * MT];
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc[i] = mma(a, Bb[i][t * KS * CS], acc[i]);
        }
        if (more) {
            store_row(oy + 3);                            // slot (oy - 1) & 3: not read by this stage
            store_d(buf ^ 1);
        }
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue: dW[co][tap][ci] += acc.  C/D maps: 32x32: co = (r&3) + 8 (r>>2) + 4 (lane>>5), ci = lane & 31;
    //      16x16: co = 4 (lane>>4) + r, ci = lane & 15
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
    auto load_row = [&](int pr) {
        const int sr = reflect_i(pr - 1, p.H);
        const float* r1 = p.x + ((size_t)b * H1 + (p.up ? sr >> 1 : sr)) * W1 * CIN;
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j) rx[j] = *reinterpret_cast<const f32x4*>(r1 + x_goff[j]);
    };
    auto store_row = [&](int pr) {
        float* dst = Ps + (pr & 3) * ROWF;
#pragma unroll
        for (int j = 0; j < X_LOADS; ++j)
            if (x_ok[j]) {                               // CS is even, not a multiple of 4: two 8-byte stores
                float2* d2 = reinterpret_cast<float2*>(dst + x_loff[j]);
                d2[0] = float2{rx[j][0], rx[j][1]};
                d2[1] = float2{rx[j][2], rx[j][3]};
            }
    };

    // B operand: lane (n = lane & 15, kidx = lane >> 4) holds W[n][tap][4 c4 + kidx] for every k-step (tap, c4)
    const int n = lane & 15, kidx = lane >> 4;
    float wreg[KSTEPS];
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) wreg[s] = p.w[(size_t)n * (9 * CIN) + (s / C4) * CIN + (s % C4) * 4 + kidx];
    const float bv = p.bias ? p.bias[n] : 0.f;
    const int a_lane = (wave * PXW + n) * CS + kidx;     // pixel lane & 15 of my first tile, channel kidx

    if (oy_begin < oy_end) {
        for (int pr = oy_begin; pr < oy_begin + 3; ++pr) {
            load_row(pr);
            store_row(pr);
        }
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

}  // namespace

bool thin_wgrad(const float* x, const float* dy, float* dw, float* dbias, const ConvShape& s, const InXform& t,
                hipStream_t st) {
    static const bool enabled = [] { const char* e = getenv("DVS_CONV_THIN"); return !(e && e[0] == '0'); }();
    if (!enabled || s.kh != 3 || s.kw != 3 || s.stride != 1 || s.pad != 1 || s.pad_mode != PAD_REFLECT || t.in_scale) return false;
    if (s.H < 8 || (t.x2 && ((s.H | s.W) & 1))) return false;
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
