// Shared device code of the implicit-GEMM convolution kernels (conv_fwd.hip, conv_bwd.hip).
//
// Data layout (HBM): activations NHWC fp32 (torch "channels_last": logical [B,C,H,W]); weights
// [Cout][kh][kw][Cin] fp32 (torch channels_last of the reference's [Cout,Cin,kh,kw] parameter), i.e. a
// row-major [N][K] matrix with K = (kh,kw,ci) contiguous -- exactly the im2col K order.
// Matrix core: v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles / instruction / SIMD = the fp32 peak).
// LDS tiles are [rows][BK + 4] floats: the 16-byte row pad makes ds_read_b128 conflict-free (slot index
// 9*row mod 16 is a permutation).  Each lane fetches 4 consecutive k per read; the MFMA's two k-slots
// (lane halves) are mapped to k = 8j + 4h + t, the same permutation for A and B, so the sum is exact.
//
// Gathers are branch-free: every lane always issues its 16-byte load from a clamped (always valid)
// address and the padding / tail zeros are selected afterwards, so the loads of a stage go out back to
// back instead of being serialised behind exec-mask branches.
#pragma once
#include "common.h"

namespace dvsconv {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BK = 32;          // k per LDS stage
constexpr int LDK = BK + 4;     // padded LDS row (floats)
constexpr int NT = 256;         // 4 waves

enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_ELU = 2, ACT_SIGMOID = 3, ACT_GELU = 4 };   // GELU: forward only (ViT MLP)
enum Pad { PAD_ZERO = 0, PAD_REFLECT = 1 };
// how the logical input is stored
enum InMode {
    IN_NHWC = 0,     // one NHWC tensor
    IN_UPCAT = 1,    // concat(upsample_nearest2x(x [B,H/2,W/2,C1]), x2 [B,H,W,Cin-C1])  (model/depthnet.py:81-85)
    IN_PLANAR = 2,   // planar [B,Cin,H,W] image, K ordered (ci,ky,kx8): encoder conv1 (model/resnet_encoder.py:102-103)
    IN_DGRAD = 3     // data-gradient gather: the "input" is dY [B,Ho,Wo,Cout] read at ((y+pad-ky)/stride, (x+pad-kx)/stride)
};

struct ConvShape {
    int B, H, W, Cin;        // logical input  [B,H,W,Cin]
    int Ho, Wo, Cout;        // output [B,Ho,Wo,Cout]
    int kh, kw, stride, pad; // pad = implicit padding on each side (zero or reflect)
    int pad_mode;
    int Ktot;                // GEMM K: kh*kw*Cin (IN_PLANAR: Cin*kh*8 with kw padded to 8)
};

struct InXform {
    const float* x2;
    int C1;
    const float* in_scale;   // per-channel affine (folded BatchNorm, or the input normalisation) ...
    const float* in_shift;
    int in_relu;             // ... followed by ReLU
    const float* aux;        // IN_DGRAD / wgrad: forward output Y of the conv (same layout as dY) ...
    int dact;                // ... whose activation derivative multiplies dY: 1 ReLU (y>0), 2 ELU (y>0 ? 1 : y+1),
                             //     3 sigmoid (y(1-y))
};

// Workgroup -> tile mapping.  Workgroups are dealt to the 8 XCDs round-robin in launch order (linear id L runs on
// XCD L % 8) and every XCD has its own 4 MB L2, so two tiles that read the same operand rows -- the output rows r
// and r+1 of a 3x3 conv share two input rows, the N tiles of one M tile share the whole im2col slice, the N tiles
// of a weight-gradient split share its pixels -- only meet in an L2 if they run on the SAME XCD close in time.
// The grids are therefore launched 1-D and XCD x works through the contiguous logical range [x*q, (x+1)*q) of
// tile ids (bijective for any workgroup count).  grid3.w = 0 switches the remap off (DVS_CONV_XCD=0).
struct Grid3 {
    int x, y, z, remap;
};
__device__ __forceinline__ int xcd_logical(int L, int nwg, int remap) {
    if (!remap) return L;
    const int q = nwg >> 3, r = nwg & 7, xcd = L & 7, idx = L >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// d act(u) / du expressed through the activation's OUTPUT y = act(u)
__device__ __forceinline__ float act_grad_from_out(float y, int act) {
    switch (act) {
        case ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case ACT_ELU: return y > 0.f ? 1.f : y + 1.f;
        case ACT_SIGMOID: return y * (1.f - y);
        default: return 1.f;
    }
}

__device__ __forceinline__ int reflect_i(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}
__device__ __forceinline__ int clampi(int i, int n) { return min(max(i, 0), n - 1); }

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_ELU: return v > 0.f ? v : expm1f(v);
        case ACT_SIGMOID: return 1.f / (1.f + __expf(-v));
        case ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));      // nn.GELU() (exact erf form)
        default: return v;
    }
}

// ---------------------------------------------------------------------------------------------
// K-position iterator: decodes the stage's k = kt*32 + c4 into (tap ky,kx ; channel ci) incrementally.
// ---------------------------------------------------------------------------------------------
struct KPos {
    int k, ci, ky, kx;
    __device__ __forceinline__ void init(int c4, const ConvShape& s) {
        k = c4;
        int tap = c4 / s.Cin;
        ci = c4 - tap * s.Cin;
        ky = tap / s.kw;
        kx = tap - ky * s.kw;
    }
    __device__ __forceinline__ void advance(const ConvShape& s) {
        k += BK;
        ci += BK;
        while (ci >= s.Cin) {
            ci -= s.Cin;
            if (++kx == s.kw) {
                kx = 0;
                ++ky;
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
// Gathers come in two halves so that a stage's global loads stay in flight across the MFMAs of the
// previous stage: `*_raw` only issues the (unconditional, clamped-address) loads and returns the validity
// flag; `finalize*` -- called right before the LDS store, after the MFMAs -- applies the padding zeros,
// the folded BatchNorm / activation-derivative arithmetic.  (Selecting the zeros next to the load would put
// an s_waitcnt vmcnt(0) in front of the MFMAs and serialise HBM latency with the matrix pipe.)
// ---------------------------------------------------------------------------------------------

// 16-byte slice (4 channels at ci) of input pixel (b, iy, ix); `ok` comes in with row validity & k < Ktot.
template <int MODE>
__device__ __forceinline__ f32x4 gather_raw(const float* __restrict__ x, const ConvShape& s, const InXform& t, int b,
                                            int iy, int ix, int ci, bool& ok) {
    if (s.pad_mode == PAD_REFLECT) {
        iy = reflect_i(iy, s.H);
        ix = reflect_i(ix, s.W);
    } else {
        ok = ok && (unsigned)iy < (unsigned)s.H && (unsigned)ix < (unsigned)s.W;
    }
    iy = clampi(iy, s.H);
    ix = clampi(ix, s.W);
    if (MODE == IN_UPCAT) {
        if (ci < t.C1) {     // block-uniform: C1 is a multiple of the 32-channel stage
            int h2 = s.H >> 1, w2 = s.W >> 1;
            return *reinterpret_cast<const f32x4*>(x + (((size_t)b * h2 + (iy >> 1)) * w2 + (ix >> 1)) * t.C1 + ci);
        }
        int c2 = s.Cin - t.C1;
        return *reinterpret_cast<const f32x4*>(t.x2 + (((size_t)b * s.H + iy) * s.W + ix) * c2 + (ci - t.C1));
    }
    return *reinterpret_cast<const f32x4*>(x + (((size_t)b * s.H + iy) * s.W + ix) * s.Cin + ci);
}

// Per-(row, tap) part of the gather, hoisted out of the K loop: padding / reflection / clamping and the pixel's
// element offset(s) are computed once per tap (every Cin/32 stages), a stage then only adds its channel.
// Offsets are 32-bit element indices (the host checks numel < 2^31).
template <int MODE>
__device__ __forceinline__ void tap_setup(const ConvShape& s, const InXform& t, int b, int iy, int ix, bool& ok,
                                          int& off, int& off2) {
    if (s.pad_mode == PAD_REFLECT) {
        iy = reflect_i(iy, s.H);
        ix = reflect_i(ix, s.W);
    } else {
        ok = ok && (unsigned)iy < (unsigned)s.H && (unsigned)ix < (unsigned)s.W;
    }
    iy = clampi(iy, s.H);
    ix = clampi(ix, s.W);
    if (MODE == IN_UPCAT) {
        int h2 = s.H >> 1, w2 = s.W >> 1;
        off = ((b * h2 + (iy >> 1)) * w2 + (ix >> 1)) * t.C1;
        off2 = ((b * s.H + iy) * s.W + ix) * (s.Cin - t.C1) - t.C1;     // + ci lands on channel ci - C1
    } else {
        off = ((b * s.H + iy) * s.W + ix) * s.Cin;
        off2 = 0;
    }
}

template <int MODE>
__device__ __forceinline__ f32x4 load_tap4(const float* __restrict__ x, const InXform& t, int off, int off2, int ci) {
    if (MODE == IN_UPCAT && ci >= t.C1) return *reinterpret_cast<const f32x4*>(t.x2 + (off2 + ci));
    return *reinterpret_cast<const f32x4*>(x + (off + ci));
}

// Same for the data-gradient gather: offset of dY pixel ((yp-ky)/stride, (xp-kx)/stride) and its validity.
__device__ __forceinline__ void dgrad_tap_setup(const ConvShape& s, int b, int yp, int xp, int ky, int kx, bool& ok,
                                                int& off) {
    int ty = yp - ky, tx = xp - kx;
    if (s.stride == 2) {
        ok = ok && ((ty | tx) & 1) == 0;
        ty >>= 1;
        tx >>= 1;
    }
    ok = ok && (unsigned)ty < (unsigned)s.H && (unsigned)tx < (unsigned)s.W;
    off = ((b * s.H + clampi(ty, s.H)) * s.W + clampi(tx, s.W)) * s.Cin;
}

// Reflection fold of the data gradient (see gather_dgrad_raw): extra terms of border pixels, already
// multiplied by the activation derivative.
__device__ __forceinline__ f32x4 load_dy4(const float* __restrict__ dy, const ConvShape& s, const InXform& t, int b,
                                          int ty, int tx, int co);
__device__ __forceinline__ f32x4 dgrad_reflect_extra(const float* __restrict__ dy, const ConvShape& s,
                                                     const InXform& t, int b, int yp, int xp, int ky, int kx, int co,
                                                     bool row_ok) {
    f32x4 extra = {0.f, 0.f, 0.f, 0.f};
    int y = yp - 1, x = xp - 1, ty = yp - ky, tx = xp - kx;
    int ey = (y == 1 && ky == 0) ? 0 : ((y == s.H - 2 && ky == s.kh - 1) ? s.H - 1 : -1);
    int ex = (x == 1 && kx == 0) ? 0 : ((x == s.W - 2 && kx == s.kw - 1) ? s.W - 1 : -1);
    if (row_ok && (ey >= 0 || ex >= 0)) {
        const bool in_y = (unsigned)ty < (unsigned)s.H, in_x = (unsigned)tx < (unsigned)s.W;
        if (ey >= 0 && in_x) extra += load_dy4(dy, s, t, b, ey, tx, co);
        if (ex >= 0 && in_y) extra += load_dy4(dy, s, t, b, ty, ex, co);
        if (ey >= 0 && ex >= 0) extra += load_dy4(dy, s, t, b, ey, ex, co);
    }
    return extra;
}

template <bool FOLD>
__device__ __forceinline__ f32x4 finalize(f32x4 v, bool ok, const f32x4& sc, const f32x4& sh, int relu) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float u = v[j];
        if (FOLD) {
            u = u * sc[j] + sh[j];
            u = relu ? fmaxf(u, 0.f) : u;
        }
        v[j] = ok ? u : 0.f;
    }
    return v;
}

// IN_DGRAD: 4 output channels [co, co+4) of dY (x act'(Y) if t.dact) feeding input pixel (y,x) through tap
// (ky,kx): source (ty,tx) = ((y + pad - ky)/stride, (x + pad - kx)/stride) when divisible and in range.
// yp = y + pad, xp = x + pad.  s.H/s.W are dY's spatial size here, s.Cin its channel count (= conv Cout).
// PAD_REFLECT (ReflectionPad2d(1) + 3x3 valid conv): the gradient of the mirrored border rows/columns folds
// back: input row 1 also receives what padded row 0 received (tap ky = 0 -> dY row 0), row H-2 what padded
// row H+1 received (tap ky = 2 -> dY row H-1); same for columns.  Those extra terms (border pixels only, a
// rare divergent branch) are returned already multiplied in `extra`.
__device__ __forceinline__ f32x4 load_dy4(const float* __restrict__ dy, const ConvShape& s, const InXform& t, int b,
                                          int ty, int tx, int co) {
    size_t o = (((size_t)b * s.H + ty) * s.W + tx) * s.Cin + co;
    f32x4 v = *reinterpret_cast<const f32x4*>(dy + o);
    if (t.dact) {
        f32x4 y = *reinterpret_cast<const f32x4*>(t.aux + o);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] *= act_grad_from_out(y[j], t.dact);
    }
    return v;
}

__device__ __forceinline__ f32x4 gather_dgrad_raw(const float* __restrict__ dy, const ConvShape& s, const InXform& t,
                                                  int b, int yp, int xp, int ky, int kx, int co, bool& ok, f32x4& yv,
                                                  f32x4& extra) {
    int ty = yp - ky, tx = xp - kx;
    if (s.stride == 2) {
        ok = ok && ((ty | tx) & 1) == 0;
        ty >>= 1;
        tx >>= 1;
    }
    const bool in_y = (unsigned)ty < (unsigned)s.H, in_x = (unsigned)tx < (unsigned)s.W;
    const bool row_ok = ok;
    ok = ok && in_y && in_x;
    size_t o = (((size_t)b * s.H + clampi(ty, s.H)) * s.W + clampi(tx, s.W)) * s.Cin + co;
    f32x4 v = *reinterpret_cast<const f32x4*>(dy + o);
    if (t.dact) yv = *reinterpret_cast<const f32x4*>(t.aux + o);
    extra = f32x4{0.f, 0.f, 0.f, 0.f};
    if (s.pad_mode == PAD_REFLECT) {
        // (yp, xp) = (y + 1, x + 1); rows/cols are the unpadded input's = dY's size
        int y = yp - 1, x = xp - 1;
        int ey = (y == 1 && ky == 0) ? 0 : ((y == s.H - 2 && ky == s.kh - 1) ? s.H - 1 : -1);
        int ex = (x == 1 && kx == 0) ? 0 : ((x == s.W - 2 && kx == s.kw - 1) ? s.W - 1 : -1);
        if (row_ok && (ey >= 0 || ex >= 0)) {
            if (ey >= 0 && in_x) extra += load_dy4(dy, s, t, b, ey, tx, co);
            if (ex >= 0 && in_y) extra += load_dy4(dy, s, t, b, ty, ex, co);
            if (ey >= 0 && ex >= 0) extra += load_dy4(dy, s, t, b, ey, ex, co);
        }
    }
    return v;
}

__device__ __forceinline__ f32x4 finalize_dgrad(f32x4 v, const f32x4& yv, const f32x4& extra, bool ok, int dact) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float u = dact ? v[j] * act_grad_from_out(yv[j], dact) : v[j];
        v[j] = (ok ? u : 0.f) + extra[j];
    }
    return v;
}

// IN_PLANAR: k = (ci*kh + ky)*8 + kx (kx padded to 8); a 16-byte k-slice is 4 horizontally adjacent
// pixels of one planar channel row.  mask4 bit e = element e is a real (in-image, kx < kw) sample.
__device__ __forceinline__ f32x4 gather_planar_raw(const float* __restrict__ x, const ConvShape& s, int b, int iy0,
                                                   int ix0, int k, bool ok, unsigned& mask4, int& ci_out) {
    int kx = k & 7, row = k >> 3;
    int ci = row / s.kh, ky = row - ci * s.kh;
    ci = min(ci, s.Cin - 1);
    ci_out = ci;
    int iy = iy0 + ky;
    bool oky = ok && (unsigned)iy < (unsigned)s.H;
    const float* rowp = x + (((size_t)b * s.Cin + ci) * s.H + clampi(iy, s.H)) * s.W;
    f32x4 v;
    mask4 = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int ix = ix0 + kx + e;
        v[e] = rowp[clampi(ix, s.W)];
        mask4 |= (oky && (unsigned)ix < (unsigned)s.W && kx + e < s.kw) ? (1u << e) : 0u;
    }
    return v;
}

template <bool FOLD>
__device__ __forceinline__ f32x4 finalize_planar(f32x4 v, unsigned mask4, float sc, float sh) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float u = FOLD ? v[e] * sc + sh : v[e];
        v[e] = ((mask4 >> e) & 1u) ? u : 0.f;
    }
    return v;
}

// One BK-deep MFMA pass over LDS tiles As[BM][LDK], Bs[BN][LDK] for a wave that owns TM x TN 32x32 tiles
// starting at rows a_row0 / b_row0.  lane = threadIdx.x & 63.
template <int TM, int TN>
__device__ __forceinline__ void mfma_stage(const float* __restrict__ As, const float* __restrict__ Bs, int a_row0,
                                           int b_row0, int lane, f32x16 (&acc)[TM][TN]) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m)
            a[m] = *reinterpret_cast<const f32x4*>(As + (a_row0 + m * 32 + r) * LDK + j * 8 + h * 4);
#pragma unroll
        for (int n = 0; n < TN; ++n)
            b[n] = *reinterpret_cast<const f32x4*>(Bs + (b_row0 + n * 32 + r) * LDK + j * 8 + h * 4);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][t], b[n][t], acc[m][n], 0, 0, 0);
    }
}

// ---- bf16 operands (dvs_set_precision(1)): LDS tiles [rows][BK] of bf16 with 16 bytes of row pad (80-byte rows: the eight lanes a
// ds_read_b128 serves together start 20 banks apart -- 0, 20, 8, 28, 16, 4, 24, 12 -- i.e. conflict-free), natural k order;
// v_mfma_f32_32x32x16_bf16 takes A[row r][k = 8 h + j] / B[k = 8 h + j][col r] as one 16-byte read per operand and 16-k step.
// The C/D map is the fp32 kernel's, so the epilogues are shared.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x4 = __attribute__((ext_vector_type(4))) __bf16;
constexpr int LDKH = BK + 8;    // padded LDS row (bf16 elements)
__device__ __forceinline__ bf16x4 to_bf16(f32x4 v) {
    return bf16x4{(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};      // v_cvt_pk_bf16_f32 x 2 (round to nearest even)
}
template <int TM, int TN>
__device__ __forceinline__ void mfma_stage_bf16(const __bf16* __restrict__ As, const __bf16* __restrict__ Bs, int a_row0, int b_row0,
                                                int lane, f32x16 (&acc)[TM][TN]) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < BK / 16; ++j) {
        bf16x8 a[TM], b[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) a[m] = *reinterpret_cast<const bf16x8*>(As + (a_row0 + m * 32 + r) * LDKH + j * 16 + h * 8);
#pragma unroll
        for (int n = 0; n < TN; ++n) b[n] = *reinterpret_cast<const bf16x8*>(Bs + (b_row0 + n * 32 + r) * LDKH + j * 16 + h * 8);
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m], b[n], acc[m][n], 0, 0, 0);
    }
}

// Buffer-addressed LDS-DMA: `buffer_load_dwordx4 v_off, s[rsrc], s_off offen lds`.  The per-lane byte offset (row, tap) is
// loop-invariant, the per-stage part (channel block / weight column) is a scalar, and a lane that must contribute zeros
// (padding, rows past the end) carries an offset beyond the descriptor's num_records: the hardware range check returns 0.
// No address arithmetic, no zero-page select: a DMA row costs NO vector-ALU instruction in the K loop.
constexpr int OOB_OFF = 0x7ffffff0;          // + any in-range scalar offset stays < 2^32 and >= num_records (< 2^31)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t dma_rsrc(const float* base, size_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ void dma16_buf(__amdgpu_buffer_rsrc_t r, int voff_bytes, int soff_bytes, float* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff_bytes,
                                             soff_bytes, 0, 0);
}

// conv_stem.hip: kernels specialised for the ResNet stems (7x7, stride 2, planar 3- / 6-channel image -> 64 channels)
bool stem_shape(const ConvShape& s);
void stem_wgrad(const float* x, const float* dy, float* dw, const ConvShape& s, const float* sc, const float* sh,
                hipStream_t st);
void stem_fwd(const float* x, const float* w, float* y, float* stats, int stat_groups, const ConvShape& s,
              const float* sc, const float* sh, hipStream_t st);

// conv_thin.hip: weight gradient of the decoder's thin full-resolution layers (Cout 16 / 32, reflection-padded 3x3);
// returns false (nothing launched) when the shape is not one of them
bool thin_wgrad_shape(const ConvShape& s, const InXform& t);
bool thin_wgrad(const float* x, const float* dy, float* dw, float* dbias, const ConvShape& s, const InXform& t,
                hipStream_t st);
bool thin_fwd(const float* x, const float* w, const float* bias, float* y, const ConvShape& s, const InXform& t, int act,
              hipStream_t st);
bool thin_dgrad(const float* dy, const float* wt, float* dx, const float* y_out, int dact, int B, int H, int W, int Cin,
                int Cout, int split_c1, float* dx_skip, hipStream_t st);

}  // namespace dvsconv
