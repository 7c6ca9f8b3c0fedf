// a14 / SURVEY.md section 8(f) rank 1: the Depth-Anything-V2 ViT-S encoder (BASELINE.json configs[4]) -- the kernels that the
// convolution engine does not already provide.  The token GEMMs (qkv, proj, fc1 + GELU, fc2, patch embedding, the DPT
// head's 1x1 / 3x3 convolutions) run on the implicit-GEMM kernels of conv_fwd.hip (a [M,K] token matrix is an NHWC tensor
// with M pixels); this file adds
//
//   dvs_attention_fwd     softmax(q k^T / sqrt(d)) v per head, flash style on the fp32 matrix cores   (attention.py:49-62)
//   dvs_layernorm_fwd     nn.LayerNorm(eps=1e-6) over the channel dimension                             (block.py:53,67; dinov2.py:166)
//   dvs_vit_patchify      [B,3,H,W] image -> [B*ph*pw, 3*14*14] patch rows, the A operand of PatchEmbed.proj   (patch_embed.py:69-82)
//   dvs_vit_assemble      cat(cls_token, patch tokens) + pos_embed                                      (dinov2.py:219-229)
//   dvs_resize_bilinear_ac  F.interpolate(mode="bilinear", align_corners=True) on NHWC maps             (blocks.py:143, dpt.py:145)
//   dvs_deconv_shuffle    the scatter half of a stride == kernel ConvTranspose2d                         (dpt.py:60-73)
//
// Attention kernel.  d = 64, N = 1370 tokens at 518x518, 6 heads.  One workgroup owns 32 queries of one head; its four
// waves split the KEYS (wave w takes key tiles w, w+4, ...) and merge their (max, sum, output) triples through LDS at the
// end, so a batch-1 forward still launches 43 x 6 = 258 workgroups for 256 CUs.  Per 32-key tile a wave computes the
// TRANSPOSED score tile S^T = K Q^T with v_mfma_f32_32x32x2_f32 (A = 32 keys x 64 dims straight from HBM as 16-byte
// vectors, B = the wave's 32 queries, resident in registers, pre-multiplied by 1/sqrt(d)): in the C/D map of that
// instruction a lane then holds 16 keys of ONE query, so the row maximum and the row sum of the online softmax are
// in-lane reductions plus one cross-half shuffle -- no LDS, no 32-lane butterflies -- and the probabilities
// p = exp(s - m) sit in exactly the registers the next product needs them in: O^T += V^T P^T takes P^T as its B
// operand register for register (the k index of the operand and the row index of the accumulator run through the keys
// in the same order), with A = V^T gathered with the head dimension along the lanes (coalesced 128-byte rows).
// Rescaling O by exp(m_old - m_new) is a per-lane scalar.  64 MFMAs and ~110 vector instructions per 32x32 tile.
#include "common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int NT = 256;
constexpr int HD = 64;             // head dimension
constexpr int QT = 32;             // queries per workgroup
constexpr int KT = 32;             // keys per tile

struct AttnParams {
    const float* qkv;              // [B][N][3][heads][HD]
    float* out;                    // [B][N][heads*HD]
    int B, N, heads;
    float scale;
    float* lse;                    // optional [B][heads][N]: log-sum-exp of each score row (kept for the backward)
};

__global__ __launch_bounds__(NT, 2) void attention_fwd_kernel(AttnParams p) {
    __shared__ float sO[4][HD][QT + 1];        // per wave: O^T (dims x queries); +1: conflict-free transposed read
    __shared__ float sM[4][QT], sL[4][QT];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // (scalar: the per-tile buffer resources derive from it)
    const int q0 = blockIdx.x * QT, head = blockIdx.y, b = blockIdx.z;
    const int N = p.N, C = p.heads * HD;
    const size_t row = (size_t)3 * C;                                   // floats between consecutive tokens
    const float* base = p.qkv + (size_t)b * N * row + (size_t)head * HD;
    const float* Q = base, * K = base + C, * V = base + 2 * C;

    // B operand: my query (column r), dims 4 (2 j + h) .. + 3 for j = 0..7, pre-scaled
    f32x4 qv[8];
    {
        const int qi = min(q0 + r, N - 1);
        const float* qp = Q + (size_t)qi * row;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            f32x4 t = *reinterpret_cast<const f32x4*>(qp + 4 * (2 * j + h));
            qv[j] = t * p.scale;
        }
    }
    f32x16 o0 = {0.f}, o1 = {0.f};             // O^T tiles: dims 0..31 and 32..63 x my query
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (N + KT - 1) / KT;
    // K / V through buffer resources that end with this image's last token: a key row past N reads zeros from the range check (its
    // score is masked below, its V row multiplies p = 0), and a lane's 8 + 32 offsets are loop invariant -- the tile moves the
    // resource's base.  (Round 2 clamped and multiplied every row index per load: ~160 vector instructions of address arithmetic per tile
    // beside 64 fp32 MFMAs that run on the same vector pipe.)
    const unsigned rowb = (unsigned)row * 4u, kvbytes = (unsigned)((size_t)N * row * 4 - (size_t)(C + head * HD) * 4);
    unsigned koff[8], voff[2][4][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) koff[j] = (unsigned)r * rowb + (unsigned)(4 * (2 * j + h)) * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            voff[0][j][u] = (unsigned)(8 * j + 4 * h + u) * rowb + (unsigned)r * 4u;
            voff[1][j][u] = voff[0][j][u] + 128u;
        }
    for (int t = wave; t < ntiles; t += 4) {
        const int k0 = t * KT;
        // the tile's resources start at its first key row (scalar arithmetic); the range check looks at the voffset alone
        const unsigned tb = (unsigned)k0 * rowb;
        const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(K)) + tb, 0, (int)(kvbytes - tb), 0x00020000);
        const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(V)) + tb, 0, (int)(kvbytes - (unsigned)C * 4u - tb), 0x00020000);
        // ---- S^T = K Q^T : A operand = key row k0 + r, dims 4 (2 j + h) .. + 3
        f32x4 kv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) kv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kr, koff[j], 0, 0));
        // V^T operand: A[m = dim][k = key]: lane (r, h) of step j needs V[k0 + 8 j + 4 h + u][32 dt + r], u = 0..3
        float vv[2][4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                vv[0][j][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vr, voff[0][j][u], 0, 0));
                vv[1][j][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vr, voff[1][j][u], 0, 0));
            }
        }
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int u = 0; u < 4; ++u) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kv[j][u], qv[j][u], s, 0, 0, 0);
        }
        // s[i]: key k0 + (i & 3) + 8 (i >> 2) + 4 h, query q0 + r.  Keys beyond N do not exist.
        if (k0 + KT > N) {                                     // (wave-uniform: only the last tile has such keys)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = k0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                s[i] = key < N ? s[i] : -INFINITY;
            }
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, s[i]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);                 // finite: every tile holds at least one real key
        const float alpha = __expf(m_run - m_new);              // exp(-inf) = 0 on the first tile
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[i] = __expf(s[i] - m_new);
            psum += s[i];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            o0[i] *= alpha;
            o1[i] *= alpha;
        }
        // ---- O^T += V^T P^T : B operand of step (j, u) = p for key 8 j + 4 h + u of my query = s[4 j + u]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                o0 = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[0][j][u], s[4 * j + u], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[1][j][u], s[4 * j + u], o1, 0, 0, 0);
            }
        }
    }

    // ---- merge the four waves' partial results: o[i] = O^T[dim = (i & 3) + 8 (i >> 2) + 4 h (+32)][query r]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int d = (i & 3) + 8 * (i >> 2) + 4 * h;
        sO[wave][d][r] = o0[i];
        sO[wave][32 + d][r] = o1[i];
    }
    if (h == 0) {
        sM[wave][r] = m_run;
        sL[wave][r] = l_run;
    }
    __syncthreads();
    // thread -> (query = tid >> 3, 8 consecutive dims): a token's 64 output floats are one 256-byte segment
    {
        const int q = tid >> 3, d0 = (tid & 7) * 8;
        const float m = fmaxf(fmaxf(sM[0][q], sM[1][q]), fmaxf(sM[2][q], sM[3][q]));
        float w[4], l = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            w[v] = __expf(sM[v][q] - m);                       // a wave that saw no tile has m = -inf, l = 0: weight 0
            l += w[v] * sL[v][q];
        }
        const float inv = 1.f / l;
        if (p.lse && d0 == 0 && q0 + q < N) p.lse[((size_t)b * p.heads + head) * N + q0 + q] = m + __logf(l);
        if (q0 + q < N) {
            float* op = p.out + ((size_t)b * N + q0 + q) * C + (size_t)head * HD + d0;
            float res[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float acc = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v) acc += w[v] * sO[v][d0 + e][q];
                res[e] = acc * inv;
            }
            *reinterpret_cast<f32x4*>(op) = f32x4{res[0], res[1], res[2], res[3]};
            *reinterpret_cast<f32x4*>(op + 4) = f32x4{res[4], res[5], res[6], res[7]};
        }
    }
}


// bf16 mode (dvs_set_precision(1)), inference forward (no log-sum-exp kept): the same kernel with both products on
// v_mfma_f32_32x32x16_bf16 -- 8 matrix instructions per 32 x 32 tile instead of 64.  The loads are the fp32 kernel's; the operands
// are rounded to bf16 in registers.  The second product uses the score tile as an operand in place (guide: an accumulator tile as the
// next MFMA's operand): registers 8 s .. 8 s + 7 are the 16-key step s, whose k order is 16 s + 8 (j >> 2) + 4 h + (j & 3) -- the
// V^T operand is assembled in that same order from the loads the fp32 kernel already makes.  Softmax stays fp32.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
    return bf16x8{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
}
__global__ __launch_bounds__(NT, 2) void attention_fwd_bf16_kernel(AttnParams p) {
    __shared__ float sO[4][HD][QT + 1];        // per wave: O^T (dims x queries); +1: conflict-free transposed read
    __shared__ float sM[4][QT], sL[4][QT];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);         // (scalar: the per-tile buffer resources derive from it)
    const int q0 = blockIdx.x * QT, head = blockIdx.y, b = blockIdx.z;
    const int N = p.N, C = p.heads * HD;
    const size_t row = (size_t)3 * C;                                   // floats between consecutive tokens
    const float* base = p.qkv + (size_t)b * N * row + (size_t)head * HD;
    const float* Q = base, * K = base + C, * V = base + 2 * C;

    // B operand of 16-dim step s: my query (column r), dims 16 s + 8 h .. + 7, pre-scaled, rounded to bf16
    bf16x8 qh[4];
    {
        const int qi = min(q0 + r, N - 1);
        const float* qp = Q + (size_t)qi * row;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(qp + 16 * s4 + 8 * h) * p.scale;
            const f32x4 t1 = *reinterpret_cast<const f32x4*>(qp + 16 * s4 + 8 * h + 4) * p.scale;
            qh[s4] = pack8(t0, t1);
        }
    }
    f32x16 o0 = {0.f}, o1 = {0.f};             // O^T tiles: dims 0..31 and 32..63 x my query
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    float m_run = -INFINITY, l_run = 0.f;

    const int ntiles = (N + KT - 1) / KT;
    // K / V through buffer resources that end with this image's last token: a key row past N reads zeros from the range check (its
    // score is masked below, its V row multiplies p = 0), and a lane's 8 + 32 offsets are loop invariant -- the tile moves the
    // resource's base.  (Round 2 clamped and multiplied every row index per load: ~160 vector instructions of address arithmetic per tile
    // beside 64 fp32 MFMAs that run on the same vector pipe.)
    const unsigned rowb = (unsigned)row * 4u, kvbytes = (unsigned)((size_t)N * row * 4 - (size_t)(C + head * HD) * 4);
    unsigned koff[8], voff[2][4][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) koff[j] = (unsigned)r * rowb + (unsigned)(16 * (j >> 1) + 8 * h + 4 * (j & 1)) * 4u;      // step j >> 1, half j & 1
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            voff[0][j][u] = (unsigned)(8 * j + 4 * h + u) * rowb + (unsigned)r * 4u;
            voff[1][j][u] = voff[0][j][u] + 128u;
        }
    for (int t = wave; t < ntiles; t += 4) {
        const int k0 = t * KT;
        // the tile's resources start at its first key row (scalar arithmetic); the range check looks at the voffset alone
        const unsigned tb = (unsigned)k0 * rowb;
        const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(K)) + tb, 0, (int)(kvbytes - tb), 0x00020000);
        const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(V)) + tb, 0, (int)(kvbytes - (unsigned)C * 4u - tb), 0x00020000);
        // ---- S^T = K Q^T : A operand of step s = key row k0 + r, dims 16 s + 8 h .. + 7 (loads 2 s and 2 s + 1)
        f32x4 kv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) kv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kr, koff[j], 0, 0));
        // V^T operand: A[m = dim][k = key]: lane (r, h) of step j needs V[k0 + 8 j + 4 h + u][32 dt + r], u = 0..3
        float vv[2][4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                vv[0][j][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vr, voff[0][j][u], 0, 0));
                vv[1][j][u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(vr, voff[1][j][u], 0, 0));
            }
        }
        f32x16 s;
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pack8(kv[2 * s4], kv[2 * s4 + 1]), qh[s4], s, 0, 0, 0);
        // s[i]: key k0 + (i & 3) + 8 (i >> 2) + 4 h, query q0 + r.  Keys beyond N do not exist.
        if (k0 + KT > N) {                                     // (wave-uniform: only the last tile has such keys)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = k0 + (i & 3) + 8 * (i >> 2) + 4 * h;
                s[i] = key < N ? s[i] : -INFINITY;
            }
        }
        float tmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) tmax = fmaxf(tmax, s[i]);
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);                 // finite: every tile holds at least one real key
        const float alpha = __expf(m_run - m_new);              // exp(-inf) = 0 on the first tile
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            s[i] = __expf(s[i] - m_new);
            psum += s[i];
        }
        psum += __shfl_xor(psum, 32, 64);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            o0[i] *= alpha;
            o1[i] *= alpha;
        }
        // ---- O^T += V^T P^T in two 16-key steps: element jj of lane half h is key 16 s2 + 8 (jj >> 2) + 4 h + (jj & 3) in BOTH operands --
        // P^T from registers 8 s2 .. 8 s2 + 7 of the score tile, V^T from the loads (j = 2 s2 + (jj >> 2), u = jj & 3) above
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16x8 ph = bf16x8{(__bf16)s[8 * s2], (__bf16)s[8 * s2 + 1], (__bf16)s[8 * s2 + 2], (__bf16)s[8 * s2 + 3],
                                     (__bf16)s[8 * s2 + 4], (__bf16)s[8 * s2 + 5], (__bf16)s[8 * s2 + 6], (__bf16)s[8 * s2 + 7]};
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const bf16x8 vh = bf16x8{(__bf16)vv[dt][2 * s2][0], (__bf16)vv[dt][2 * s2][1], (__bf16)vv[dt][2 * s2][2], (__bf16)vv[dt][2 * s2][3],
                                         (__bf16)vv[dt][2 * s2 + 1][0], (__bf16)vv[dt][2 * s2 + 1][1], (__bf16)vv[dt][2 * s2 + 1][2],
                                         (__bf16)vv[dt][2 * s2 + 1][3]};
                if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o0, 0, 0, 0);
                else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vh, ph, o1, 0, 0, 0);
            }
        }
    }

    // ---- merge the four waves' partial results: o[i] = O^T[dim = (i & 3) + 8 (i >> 2) + 4 h (+32)][query r]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int d = (i & 3) + 8 * (i >> 2) + 4 * h;
        sO[wave][d][r] = o0[i];
        sO[wave][32 + d][r] = o1[i];
    }
    if (h == 0) {
        sM[wave][r] = m_run;
        sL[wave][r] = l_run;
    }
    __syncthreads();
    // thread -> (query = tid >> 3, 8 consecutive dims): a token's 64 output floats are one 256-byte segment
    {
        const int q = tid >> 3, d0 = (tid & 7) * 8;
        const float m = fmaxf(fmaxf(sM[0][q], sM[1][q]), fmaxf(sM[2][q], sM[3][q]));
        float w[4], l = 0.f;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            w[v] = __expf(sM[v][q] - m);                       // a wave that saw no tile has m = -inf, l = 0: weight 0
            l += w[v] * sL[v][q];
        }
        const float inv = 1.f / l;
        if (p.lse && d0 == 0 && q0 + q < N) p.lse[((size_t)b * p.heads + head) * N + q0 + q] = m + __logf(l);
        if (q0 + q < N) {
            float* op = p.out + ((size_t)b * N + q0 + q) * C + (size_t)head * HD + d0;
            float res[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float acc = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v) acc += w[v] * sO[v][d0 + e][q];
                res[e] = acc * inv;
            }
            *reinterpret_cast<f32x4*>(op) = f32x4{res[0], res[1], res[2], res[3]};
            *reinterpret_cast<f32x4*>(op + 4) = f32x4{res[4], res[5], res[6], res[7]};
        }
    }
}



// ---- attention backward -------------------------------------------------------------------------------------------------
// With P = exp(S - lse), D_q = sum_d dO[q][d] O[q][d]:   dV = P^T dO,   dP = dO V^T,   dS = P o (dP - D),
// dQ = scale dS K,   dK = scale dS^T Q.   Two kernels in the forward's register layout, each recomputing S and dP for its
// tiles (7 GEMM units against the forward's 2; no atomics):
//   attention_bwd_dq_kernel   owns 32 queries (waves split the keys):  S^T = K Q^T and dP^T = V dO^T land key-major per
//                             lane exactly like the forward's scores, dS^T feeds dQ^T += K^T dS^T as a B operand from its
//                             registers;
//   attention_bwd_dkv_kernel  owns 32 keys (waves split the queries):  S = Q K^T and dP = dO V^T land query-major per lane
//                             (a lane holds 16 queries of one key), P and dS feed dV^T += dO^T P and dK^T += Q^T dS.
struct AttnBwdParams {
    const float* qkv;              // [B][N][3][heads][HD]
    const float* d_out;            // [B][N][heads*HD]
    const float* lse;              // [B][heads][N]
    const float* delta;            // [B][heads][N]   D_q
    float* d_qkv;                  // [B][N][3][heads][HD]
    int B, N, heads;
    float scale;
};

__global__ __launch_bounds__(NT) void attention_delta_kernel(const float* __restrict__ out, const float* __restrict__ d_out,
                                                             float* __restrict__ delta, int B, int N, int heads) {
    const size_t n = (size_t)B * N * heads;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const int head = (int)(i % heads);
        const size_t bq = i / heads;
        const int q = (int)(bq % N), b = (int)(bq / N);
        const f32x4* o = reinterpret_cast<const f32x4*>(out + (bq * heads + head) * HD);
        const f32x4* g = reinterpret_cast<const f32x4*>(d_out + (bq * heads + head) * HD);
        float acc = 0.f;
#pragma unroll
        for (int e = 0; e < HD / 4; ++e) {
            const f32x4 a = o[e], c = g[e];
            acc += (a[0] * c[0] + a[1] * c[1]) + (a[2] * c[2] + a[3] * c[3]);
        }
        delta[((size_t)b * heads + head) * N + q] = acc;
    }
}

__global__ __launch_bounds__(NT, 2) void attention_bwd_dq_kernel(AttnBwdParams p) {
    __shared__ float sO[4][HD][QT + 1];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q0 = blockIdx.x * QT, head = blockIdx.y, b = blockIdx.z;
    const int N = p.N, C = p.heads * HD;
    const size_t row = (size_t)3 * C;
    const float* base = p.qkv + (size_t)b * N * row + (size_t)head * HD;
    const float* Q = base, * K = base + C, * V = base + 2 * C;
    const int qi = min(q0 + r, N - 1);
    f32x4 qv[8], gv[8];
    {
        const float* qp = Q + (size_t)qi * row;
        const float* gp = p.d_out + ((size_t)b * N + qi) * C + (size_t)head * HD;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            qv[j] = *reinterpret_cast<const f32x4*>(qp + 4 * (2 * j + h)) * p.scale;
            gv[j] = *reinterpret_cast<const f32x4*>(gp + 4 * (2 * j + h));
        }
    }
    const float Lq = p.lse[((size_t)b * p.heads + head) * N + qi], Dq = p.delta[((size_t)b * p.heads + head) * N + qi];
    f32x16 a0, a1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a0[i] = 0.f; a1[i] = 0.f; }
    const int ntiles = (N + KT - 1) / KT;
    // K / V through per-tile buffer resources with loop-invariant lane offsets (see attention_fwd_kernel): key rows past N read zeros
    const unsigned rowb = (unsigned)row * 4u, kvbytes = (unsigned)((size_t)N * row * 4 - (size_t)(C + head * HD) * 4);
    unsigned koff[8], ktoff[2][4][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) koff[j] = (unsigned)r * rowb + (unsigned)(4 * (2 * j + h)) * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ktoff[0][j][u] = (unsigned)(8 * j + 4 * h + u) * rowb + (unsigned)r * 4u;
            ktoff[1][j][u] = ktoff[0][j][u] + 128u;
        }
    for (int t = wave; t < ntiles; t += 4) {
        const int k0 = t * KT;
        const unsigned tb = (unsigned)k0 * rowb;
        const __amdgpu_buffer_rsrc_t kr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(K)) + tb, 0, (int)(kvbytes - tb), 0x00020000);
        const __amdgpu_buffer_rsrc_t vr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(V)) + tb, 0, (int)(kvbytes - (unsigned)C * 4u - tb), 0x00020000);
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
        {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 kk = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(kr, koff[j], 0, 0));
                const f32x4 vv = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(vr, koff[j], 0, 0));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(kk[u], qv[j][u], s, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(vv[u], gv[j][u], dp, 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] = __expf(s[i] - Lq);
        if (k0 + KT > N) {                                     // (wave-uniform: only the last tile has keys past N)
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = k0 + (i & 3) + 8 * (i >> 2) + 4 * h < N ? s[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s[i] *= dp[i] - Dq;      // dS^T for (key, my query)
        // dQ^T[dim][query] += K^T[dim][key] dS^T[key][query]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float ka = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(kr, ktoff[0][j][u], 0, 0));
                const float kb = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(kr, ktoff[1][j][u], 0, 0));
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(ka, s[4 * j + u], a0, 0, 0, 0);       // (a key row past N: dS = 0 times K = 0)
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(kb, s[4 * j + u], a1, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int d = (i & 3) + 8 * (i >> 2) + 4 * h;
        sO[wave][d][r] = a0[i];
        sO[wave][32 + d][r] = a1[i];
    }
    __syncthreads();
    {
        const int q = tid >> 3, d0 = (tid & 7) * 8;
        if (q0 + q < N) {
            float* op = p.d_qkv + ((size_t)b * N + q0 + q) * row + (size_t)head * HD + d0;
            float res[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                res[e] = ((sO[0][d0 + e][q] + sO[1][d0 + e][q]) + (sO[2][d0 + e][q] + sO[3][d0 + e][q])) * p.scale;
            *reinterpret_cast<f32x4*>(op) = f32x4{res[0], res[1], res[2], res[3]};
            *reinterpret_cast<f32x4*>(op + 4) = f32x4{res[4], res[5], res[6], res[7]};
        }
    }
}

__global__ __launch_bounds__(NT, 1) void attention_bwd_dkv_kernel(AttnBwdParams p) {
    __shared__ float sK[4][HD][QT + 1];
    __shared__ float sV[4][HD][QT + 1];
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int k0 = blockIdx.x * KT, head = blockIdx.y, b = blockIdx.z;
    const int N = p.N, C = p.heads * HD;
    const size_t row = (size_t)3 * C;
    const float* base = p.qkv + (size_t)b * N * row + (size_t)head * HD;
    const float* Q = base, * K = base + C, * V = base + 2 * C;
    const float* G = p.d_out + (size_t)b * N * C + (size_t)head * HD;          // dO rows, stride C
    const float* lse = p.lse + ((size_t)b * p.heads + head) * N;
    const float* dlt = p.delta + ((size_t)b * p.heads + head) * N;
    // B operands: my key (column r), resident
    f32x4 kv[8], vv[8];
    {
        const int ki = min(k0 + r, N - 1);
        const float* kp = K + (size_t)ki * row;
        const float* vp = V + (size_t)ki * row;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            kv[j] = *reinterpret_cast<const f32x4*>(kp + 4 * (2 * j + h));
            vv[j] = *reinterpret_cast<const f32x4*>(vp + 4 * (2 * j + h));
        }
    }
    f32x16 dk0, dk1, dv0, dv1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dk0[i] = 0.f; dk1[i] = 0.f; dv0[i] = 0.f; dv1[i] = 0.f; }
    const int ntiles = (N + QT - 1) / QT;
    // Q / dO rows through per-tile buffer resources with loop-invariant lane offsets (see attention_fwd_kernel): query rows past N
    // read zeros (their P and dS are masked below)
    const unsigned rowb = (unsigned)row * 4u, gb = (unsigned)C * 4u;
    const unsigned qbytes = (unsigned)((size_t)N * row * 4 - (size_t)(head * HD) * 4), gbytes = (unsigned)((size_t)N * C * 4 - (size_t)(head * HD) * 4);
    unsigned qoff[8], goff[8], qtoff[2][4][4], gtoff[2][4][4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        qoff[j] = (unsigned)r * rowb + (unsigned)(4 * (2 * j + h)) * 4u;
        goff[j] = (unsigned)r * gb + (unsigned)(4 * (2 * j + h)) * 4u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            qtoff[0][j][u] = (unsigned)(8 * j + 4 * h + u) * rowb + (unsigned)r * 4u;
            qtoff[1][j][u] = qtoff[0][j][u] + 128u;
            gtoff[0][j][u] = (unsigned)(8 * j + 4 * h + u) * gb + (unsigned)r * 4u;
            gtoff[1][j][u] = gtoff[0][j][u] + 128u;
        }
    for (int t = wave; t < ntiles; t += 4) {
        const int q0 = t * QT;
        const unsigned tq = (unsigned)q0 * rowb, tg = (unsigned)q0 * gb;
        const __amdgpu_buffer_rsrc_t qr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(Q)) + tq, 0, (int)(qbytes - tq), 0x00020000);
        const __amdgpu_buffer_rsrc_t gr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<char*>(reinterpret_cast<const char*>(G)) + tg, 0, (int)(gbytes - tg), 0x00020000);
        f32x16 s, dp;
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[i] = 0.f; dp[i] = 0.f; }
        {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 qq = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qr, qoff[j], 0, 0)) * p.scale;
                const f32x4 gg = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(gr, goff[j], 0, 0));
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    s = __builtin_amdgcn_mfma_f32_32x32x2f32(qq[u], kv[j][u], s, 0, 0, 0);       // S[query][key]
                    dp = __builtin_amdgcn_mfma_f32_32x32x2f32(gg[u], vv[j][u], dp, 0, 0, 0);     // dP[query][key]
                }
            }
        }
        // s[i] / dp[i]: query q0 + (i & 3) + 8 (i >> 2) + 4 h, my key
        {
            // log-sum-exp and delta of my 16 queries: per-tile resources again (a query past N reads 0 and is masked in the last tile)
            const unsigned tl = (unsigned)q0 * 4u;
            const __amdgpu_buffer_rsrc_t lr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(lse)) + tl, 0, (int)((unsigned)N * 4u - tl), 0x00020000);
            const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<char*>(reinterpret_cast<const char*>(dlt)) + tl, 0, (int)((unsigned)N * 4u - tl), 0x00020000);
            float lv[16], dv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                lv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(lr, (unsigned)(16 * h) + (unsigned)(((i & 3) + 8 * (i >> 2)) * 4), 0, 0));
                dv[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dr, (unsigned)(16 * h) + (unsigned)(((i & 3) + 8 * (i >> 2)) * 4), 0, 0));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = __expf(s[i] - lv[i]);          // P
            if (q0 + QT > N) {
#pragma unroll
                for (int i = 0; i < 16; ++i) s[i] = q0 + (i & 3) + 8 * (i >> 2) + 4 * h < N ? s[i] : 0.f;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) dp[i] = s[i] * (dp[i] - dv[i]);       // dS
        }
        // dV^T[dim][key] += dO^T[dim][query] P[query][key];  dK^T[dim][key] += Q^T[dim][query] dS[query][key]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float g0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, gtoff[0][j][u], 0, 0));
                const float g1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(gr, gtoff[1][j][u], 0, 0));
                const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(qr, qtoff[0][j][u], 0, 0));
                const float x1 = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(qr, qtoff[1][j][u], 0, 0));
                dv0 = __builtin_amdgcn_mfma_f32_32x32x2f32(g0, s[4 * j + u], dv0, 0, 0, 0);
                dv1 = __builtin_amdgcn_mfma_f32_32x32x2f32(g1, s[4 * j + u], dv1, 0, 0, 0);
                dk0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, dp[4 * j + u], dk0, 0, 0, 0);
                dk1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, dp[4 * j + u], dk1, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int d = (i & 3) + 8 * (i >> 2) + 4 * h;
        sK[wave][d][r] = dk0[i];
        sK[wave][32 + d][r] = dk1[i];
        sV[wave][d][r] = dv0[i];
        sV[wave][32 + d][r] = dv1[i];
    }
    __syncthreads();
    {
        const int k = tid >> 3, d0 = (tid & 7) * 8;
        if (k0 + k < N) {
            float* okp = p.d_qkv + ((size_t)b * N + k0 + k) * row + C + (size_t)head * HD + d0;
            float* ovp = okp + C;
            float rk[8], rv[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                rk[e] = ((sK[0][d0 + e][k] + sK[1][d0 + e][k]) + (sK[2][d0 + e][k] + sK[3][d0 + e][k])) * p.scale;
                rv[e] = (sV[0][d0 + e][k] + sV[1][d0 + e][k]) + (sV[2][d0 + e][k] + sV[3][d0 + e][k]);
            }
            *reinterpret_cast<f32x4*>(okp) = f32x4{rk[0], rk[1], rk[2], rk[3]};
            *reinterpret_cast<f32x4*>(okp + 4) = f32x4{rk[4], rk[5], rk[6], rk[7]};
            *reinterpret_cast<f32x4*>(ovp) = f32x4{rv[0], rv[1], rv[2], rv[3]};
            *reinterpret_cast<f32x4*>(ovp + 4) = f32x4{rv[4], rv[5], rv[6], rv[7]};
        }
    }
}

// ---- LayerNorm over C channels of each of M rows: one wavefront per row, two passes over registers -----------------
template <int VPL>    // float4 vectors per lane: C = 256 * VPL (partially filled for smaller C)
__global__ __launch_bounds__(NT) void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                       const float* __restrict__ bta, float* __restrict__ y, int M, int C,
                                                       float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int cv = C >> 2;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * C);
    f32x4 v[VPL];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < cv ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = dvs::wave_sum(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < cv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float d = v[i][e] - mean;
                sq += d * d;
            }
        }
    }
    const float rstd = rsqrtf(dvs::wave_sum(sq) / (float)C + eps);
    f32x4* yr = reinterpret_cast<f32x4*>(y + (size_t)row * C);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < cv) {
            const f32x4 gg = reinterpret_cast<const f32x4*>(g)[c], bb = reinterpret_cast<const f32x4*>(bta)[c];
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gg[e] + bb[e];
            yr[c] = o;
        }
    }
}

// ---- patch rows: out[(b, py, px)][ci * P * P + ky * P + kx] = img[b][ci][py * P + ky][px * P + kx] -----------------
__global__ __launch_bounds__(NT) void patchify_kernel(const float* __restrict__ img, float* __restrict__ out, int B, int H,
                                                      int W, int P, int Kpad) {
    const int ph = H / P, pw = W / P, K = 3 * P * P;
    const size_t n = (size_t)B * ph * pw * Kpad;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const int k = (int)(i % Kpad);
        size_t t = i / Kpad;
        const int px = (int)(t % pw);
        t /= pw;
        const int py = (int)(t % ph), b = (int)(t / ph);
        float v = 0.f;
        if (k < K) {
            const int ci = k / (P * P), rem = k - ci * P * P, ky = rem / P, kx = rem - ky * P;
            v = img[(((size_t)b * 3 + ci) * H + py * P + ky) * W + px * P + kx];
        }
        out[i] = v;
    }
}

// ---- x[b][0] = cls + pos[0]; x[b][1 + i] = tok[b][i] + pos[1 + i] ------------------------------------------------
__global__ __launch_bounds__(NT) void vit_assemble_kernel(const float* __restrict__ tok, const float* __restrict__ cls,
                                                          const float* __restrict__ pos, float* __restrict__ x, int B, int Np,
                                                          int C) {
    const int cv = C >> 2;
    const size_t n = (size_t)B * (Np + 1) * cv;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % cv);
        size_t t = i / cv;
        const int tokn = (int)(t % (Np + 1)), b = (int)(t / (Np + 1));
        const f32x4 pe = reinterpret_cast<const f32x4*>(pos)[(size_t)tokn * cv + c];
        const f32x4 v = tokn == 0 ? reinterpret_cast<const f32x4*>(cls)[c]
                                  : reinterpret_cast<const f32x4*>(tok)[((size_t)b * Np + tokn - 1) * cv + c];
        reinterpret_cast<f32x4*>(x)[i] = v + pe;
    }
}

// ---- bilinear resize, align_corners=True, NHWC ---------------------------------------------------------------------
__global__ __launch_bounds__(NT) void resize_bilinear_ac_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int h,
                                                                int w, int H, int W, int C) {
    const int cv = C >> 2;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const size_t n = (size_t)B * H * W * cv;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % cv);
        size_t t = i / cv;
        const int X = (int)(t % W);
        t /= W;
        const int Y = (int)(t % H), b = (int)(t / H);
        const float fy = sy * Y, fx = sx * X;
        const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly = fy - y0, lx = fx - x0;
        const f32x4* xb = reinterpret_cast<const f32x4*>(x) + (size_t)b * h * w * cv + c;
        const f32x4 v00 = xb[((size_t)y0 * w + x0) * cv], v01 = xb[((size_t)y0 * w + x1) * cv];
        const f32x4 v10 = xb[((size_t)y1 * w + x0) * cv], v11 = xb[((size_t)y1 * w + x1) * cv];
        // torch's upsample_bilinear2d: (1 - ly) * ((1 - lx) v00 + lx v01) + ly * ((1 - lx) v10 + lx v11)
        reinterpret_cast<f32x4*>(y)[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
}

// ---- ConvTranspose2d with stride == kernel: y[b][i k + a][j k + c][co] = g[b][i][j][(a k + c) Co + co] --------------
__global__ __launch_bounds__(NT) void deconv_shuffle_kernel(const float* __restrict__ g, float* __restrict__ y, int B, int h,
                                                            int w, int k, int Co) {
    const int cv = Co >> 2, H = h * k, W = w * k;
    const size_t n = (size_t)B * H * W * cv;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % cv);
        size_t t = i / cv;
        const int X = (int)(t % W);
        t /= W;
        const int Y = (int)(t % H), b = (int)(t / H);
        const int ii = Y / k, a = Y - ii * k, jj = X / k, cc = X - jj * k;
        reinterpret_cast<f32x4*>(y)[i] =
            reinterpret_cast<const f32x4*>(g)[(((size_t)b * h + ii) * w + jj) * (size_t)(k * k * cv) + (size_t)(a * k + cc) * cv + c];
    }
}


// ---- LayerNorm backward: dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma; dgamma += sum_rows dy xhat, dbeta += sum_rows dy
template <int VPL>
__global__ __launch_bounds__(NT) void layernorm_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g,
                                                           const float* __restrict__ dy, float* __restrict__ dx,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta, int M, int C,
                                                           float eps) {
    const int lane = threadIdx.x & 63, wrow = blockIdx.x * (NT / 64) + (threadIdx.x >> 6), nw = gridDim.x * (NT / 64);
    const int cv = C >> 2;
    f32x4 ag[VPL], ab[VPL];
#pragma unroll
    for (int i = 0; i < VPL; ++i) { ag[i] = f32x4{0.f, 0.f, 0.f, 0.f}; ab[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    for (int row = wrow; row < M; row += nw) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * C);
        const f32x4* gr = reinterpret_cast<const f32x4*>(dy + (size_t)row * C);
        f32x4 v[VPL], d[VPL];
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < cv ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
            d[i] = c < cv ? gr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
            sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        }
        const float mean = dvs::wave_sum(sum) / (float)C;
        float sq = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < cv) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = v[i][e] - mean;
                    sq += t * t;
                }
            }
        }
        const float rstd = rsqrtf(dvs::wave_sum(sq) / (float)C + eps);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < cv) {
                const f32x4 gg = reinterpret_cast<const f32x4*>(g)[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = (v[i][e] - mean) * rstd;
                    ag[i][e] += d[i][e] * xh;
                    ab[i][e] += d[i][e];
                    const float gd = d[i][e] * gg[e];
                    v[i][e] = xh;
                    d[i][e] = gd;
                    s1 += gd;
                    s2 += gd * xh;
                }
            }
        }
        s1 = dvs::wave_sum(s1) / (float)C;
        s2 = dvs::wave_sum(s2) / (float)C;
        f32x4* dr = reinterpret_cast<f32x4*>(dx + (size_t)row * C);
#pragma unroll
        for (int i = 0; i < VPL; ++i) {
            const int c = lane + 64 * i;
            if (c < cv) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rstd * (d[i][e] - s1 - v[i][e] * s2);
                dr[c] = o;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const int c = lane + 64 * i;
        if (c < cv) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                atomicAdd(dgamma + 4 * c + e, ag[i][e]);
                atomicAdd(dbeta + 4 * c + e, ab[i][e]);
            }
        }
    }
}

// ---- elementwise activations as their own passes (training keeps the pre-activation): 1 ReLU, 4 GELU (erf form) -----------
__global__ __launch_bounds__(NT) void act_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, size_t nvec, int act) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < nvec; i += (size_t)gridDim.x * NT) {
        f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            v[e] = act == 4 ? 0.5f * v[e] * (1.f + erff(v[e] * 0.70710678118654752f)) : fmaxf(v[e], 0.f);
        reinterpret_cast<f32x4*>(y)[i] = v;
    }
}

// dx = dy * act'(x) from the INPUT x of the activation
__global__ __launch_bounds__(NT) void act_bwd_in_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx,
                                                        size_t nvec, int act) {
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < nvec; i += (size_t)gridDim.x * NT) {
        const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (act == 4) {     // d/dx [x Phi(x)] = Phi(x) + x phi(x)
                const float cdf = 0.5f * (1.f + erff(v[e] * 0.70710678118654752f));
                const float pdf = 0.3989422804014327f * __expf(-0.5f * v[e] * v[e]);
                g[e] *= cdf + v[e] * pdf;
            } else {
                g[e] = v[e] > 0.f ? g[e] : 0.f;
            }
        }
        reinterpret_cast<f32x4*>(dx)[i] = g;
    }
}

// ---- bilinear resize (align_corners=True) backward: scatter of dy [B,H,W,C] into dx [B,h,w,C] (zero-filled by the caller)
__global__ __launch_bounds__(NT) void resize_bilinear_ac_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int h,
                                                                    int w, int H, int W, int C) {
    const int cv = C >> 2;
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const size_t n = (size_t)B * H * W * cv;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % cv);
        size_t t = i / cv;
        const int X = (int)(t % W);
        t /= W;
        const int Y = (int)(t % H), b = (int)(t / H);
        const float fy = sy * Y, fx = sx * X;
        const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly = fy - y0, lx = fx - x0;
        const f32x4 g = reinterpret_cast<const f32x4*>(dy)[i];
        float* xb = dx + ((size_t)b * h * w) * C + c * 4;
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            atomicAdd(xb + ((size_t)y0 * w + x0) * C + e, g[e] * w00);
            atomicAdd(xb + ((size_t)y0 * w + x1) * C + e, g[e] * w01);
            atomicAdd(xb + ((size_t)y1 * w + x0) * C + e, g[e] * w10);
            atomicAdd(xb + ((size_t)y1 * w + x1) * C + e, g[e] * w11);
        }
    }
}


// scalar-channel forms of the two resize kernels (C not a multiple of 4: the 1-channel disparity maps and 3-channel images
// of the encoder-swap adapter)
__global__ __launch_bounds__(NT) void resize_bilinear_ac_scalar_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                                       float* __restrict__ dx, int B, int h, int w, int H, int W,
                                                                       int C) {
    const float sy = H > 1 ? (float)(h - 1) / (float)(H - 1) : 0.f, sx = W > 1 ? (float)(w - 1) / (float)(W - 1) : 0.f;
    const size_t n = (size_t)B * H * W * C;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % C);
        size_t t = i / C;
        const int X = (int)(t % W);
        t /= W;
        const int Y = (int)(t % H), b = (int)(t / H);
        const float fy = sy * Y, fx = sx * X;
        const int y0 = min((int)fy, h - 1), x0 = min((int)fx, w - 1);
        const int y1 = min(y0 + 1, h - 1), x1 = min(x0 + 1, w - 1);
        const float ly = fy - y0, lx = fx - x0;
        const size_t base = (size_t)b * h * w * C + c;
        const size_t o00 = base + ((size_t)y0 * w + x0) * C, o01 = base + ((size_t)y0 * w + x1) * C;
        const size_t o10 = base + ((size_t)y1 * w + x0) * C, o11 = base + ((size_t)y1 * w + x1) * C;
        if (dx == nullptr) {
            y[i] = (1.f - ly) * ((1.f - lx) * x[o00] + lx * x[o01]) + ly * ((1.f - lx) * x[o10] + lx * x[o11]);
        } else {                      // backward: x = dy (read at i), scatter into dx
            const float g = x[i];
            atomicAdd(dx + o00, g * (1.f - ly) * (1.f - lx));
            atomicAdd(dx + o01, g * (1.f - ly) * lx);
            atomicAdd(dx + o10, g * ly * (1.f - lx));
            atomicAdd(dx + o11, g * ly * lx);
        }
    }
}

// ---- inverse of deconv_shuffle: dg[b][i][j][(a k + c) Co + co] = dy[b][i k + a][j k + c][co] ------------------------------------
__global__ __launch_bounds__(NT) void deconv_unshuffle_kernel(const float* __restrict__ dy, float* __restrict__ dg, int B, int h, int w,
                                                              int k, int Co) {
    const int cv = Co >> 2, H = h * k, W = w * k;
    const size_t n = (size_t)B * H * W * cv;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const int c = (int)(i % cv);
        size_t t = i / cv;
        const int X = (int)(t % W);
        t /= W;
        const int Y = (int)(t % H), b = (int)(t / H);
        const int ii = Y / k, a = Y - ii * k, jj = X / k, cc = X - jj * k;
        reinterpret_cast<f32x4*>(dg)[(((size_t)b * h + ii) * w + jj) * (size_t)(k * k * cv) + (size_t)(a * k + cc) * cv + c] =
            reinterpret_cast<const f32x4*>(dy)[i];
    }
}

inline unsigned sgrid(size_t n) {
    size_t b = (n + NT - 1) / NT;
    return (unsigned)(b > 4096 ? 4096 : (b == 0 ? 1 : b));
}

}  // namespace

extern "C" {

int dvs_attention_fwd(const float* qkv, float* out, float* lse, int B, int N, int heads, int head_dim, float scale, void* stream) {
    DVS_REQUIRE(qkv && out && B > 0 && N > 0 && heads > 0, "dvs_attention_fwd: bad argument");
    DVS_REQUIRE(head_dim == HD, "dvs_attention_fwd: head dimension 64 only (got %d)", head_dim);
    DVS_REQUIRE((double)B * N * 3 * heads * HD < 2147483648.0, "dvs_attention_fwd: qkv must have fewer than 2^31 elements");
    AttnParams p{qkv, out, B, N, heads, scale, lse};
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        dvs::ProfScope prof(dvs::SLOT_ATTN, st);
        prof.work(4.0 * B * heads * (double)N * N * HD);      // q k^T and p v: 2 N^2 d flops each
        // bf16 mode: only where no log-sum-exp is kept (inference) -- the backward kernels recompute the scores in fp32
        if (dvs::precision_bf16() && lse == nullptr)
            hipLaunchKernelGGL(attention_fwd_bf16_kernel, dim3((N + QT - 1) / QT, heads, B), dim3(NT), 0, st, p);
        else hipLaunchKernelGGL(attention_fwd_kernel, dim3((N + QT - 1) / QT, heads, B), dim3(NT), 0, st, p);
    }
    return dvs::check_launch("dvs_attention_fwd");
}

int dvs_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, int M, int C, float eps, void* stream) {
    DVS_REQUIRE(x && gamma && beta && y && M > 0 && C > 0 && (C & 3) == 0 && C <= 2048, "dvs_layernorm_fwd: C %% 4 == 0, C <= 2048 (got %d)", C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((M + NT / 64 - 1) / (NT / 64));
    if (C <= 256) hipLaunchKernelGGL(layernorm_kernel<1>, grid, dim3(NT), 0, st, x, gamma, beta, y, M, C, eps);
    else if (C <= 512) hipLaunchKernelGGL(layernorm_kernel<2>, grid, dim3(NT), 0, st, x, gamma, beta, y, M, C, eps);
    else if (C <= 1024) hipLaunchKernelGGL(layernorm_kernel<4>, grid, dim3(NT), 0, st, x, gamma, beta, y, M, C, eps);
    else hipLaunchKernelGGL(layernorm_kernel<8>, grid, dim3(NT), 0, st, x, gamma, beta, y, M, C, eps);
    return dvs::check_launch("dvs_layernorm_fwd");
}

int dvs_vit_patchify(const float* image, float* rows, int B, int H, int W, int patch, int k_padded, void* stream) {
    DVS_REQUIRE(image && rows && B > 0 && patch > 0 && H % patch == 0 && W % patch == 0 && k_padded >= 3 * patch * patch,
                "dvs_vit_patchify: bad argument");
    const size_t n = (size_t)B * (H / patch) * (W / patch) * k_padded;
    hipLaunchKernelGGL(patchify_kernel, dim3(sgrid(n)), dim3(NT), 0, static_cast<hipStream_t>(stream), image, rows, B, H, W, patch,
                       k_padded);
    return dvs::check_launch("dvs_vit_patchify");
}

int dvs_vit_assemble(const float* patch_tokens, const float* cls_token, const float* pos_embed, float* x, int B, int num_patches,
                     int C, void* stream) {
    DVS_REQUIRE(patch_tokens && cls_token && pos_embed && x && B > 0 && num_patches > 0 && C > 0 && (C & 3) == 0,
                "dvs_vit_assemble: bad argument");
    const size_t n = (size_t)B * (num_patches + 1) * (C / 4);
    hipLaunchKernelGGL(vit_assemble_kernel, dim3(sgrid(n)), dim3(NT), 0, static_cast<hipStream_t>(stream), patch_tokens, cls_token,
                       pos_embed, x, B, num_patches, C);
    return dvs::check_launch("dvs_vit_assemble");
}

int dvs_resize_bilinear_ac(const float* x, float* y, int B, int h, int w, int H, int W, int C, void* stream) {
    DVS_REQUIRE(x && y && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0, "dvs_resize_bilinear_ac: bad argument");
    if (C & 3) {
        hipLaunchKernelGGL(resize_bilinear_ac_scalar_kernel, dim3(sgrid((size_t)B * H * W * C)), dim3(NT), 0, static_cast<hipStream_t>(stream), x,
                           y, static_cast<float*>(nullptr), B, h, w, H, W, C);
        return dvs::check_launch("dvs_resize_bilinear_ac");
    }
    const size_t n = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(resize_bilinear_ac_kernel, dim3(sgrid(n)), dim3(NT), 0, static_cast<hipStream_t>(stream), x, y, B, h, w, H, W, C);
    return dvs::check_launch("dvs_resize_bilinear_ac");
}

int dvs_deconv_shuffle(const float* g, float* y, int B, int h, int w, int k, int Cout, void* stream) {
    DVS_REQUIRE(g && y && B > 0 && h > 0 && w > 0 && k > 0 && Cout > 0 && (Cout & 3) == 0, "dvs_deconv_shuffle: bad argument");
    const size_t n = (size_t)B * h * k * w * k * (Cout / 4);
    hipLaunchKernelGGL(deconv_shuffle_kernel, dim3(sgrid(n)), dim3(NT), 0, static_cast<hipStream_t>(stream), g, y, B, h, w, k, Cout);
    return dvs::check_launch("dvs_deconv_shuffle");
}

int dvs_attention_bwd(const float* qkv, const float* out, const float* d_out, const float* lse, float* delta, float* d_qkv, int B,
                      int N, int heads, int head_dim, float scale, void* stream) {
    DVS_REQUIRE(qkv && out && d_out && lse && delta && d_qkv && B > 0 && N > 0 && heads > 0, "dvs_attention_bwd: bad argument");
    DVS_REQUIRE(head_dim == HD, "dvs_attention_bwd: head dimension 64 only (got %d)", head_dim);
    DVS_REQUIRE((double)B * N * 3 * heads * HD < 2147483648.0, "dvs_attention_bwd: qkv must have fewer than 2^31 elements");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(attention_delta_kernel, dim3(sgrid((size_t)B * N * heads)), dim3(NT), 0, st, out, d_out, delta, B, N, heads);
    AttnBwdParams p{qkv, d_out, lse, delta, d_qkv, B, N, heads, scale};
    {
        dvs::ProfScope prof(dvs::SLOT_ATTN_BWD, st);
        prof.work(8.0 * B * heads * (double)N * N * HD);      // algorithmic: dV, dP, dQ, dK (the two kernels also recompute S and dP)
        hipLaunchKernelGGL(attention_bwd_dq_kernel, dim3((N + QT - 1) / QT, heads, B), dim3(NT), 0, st, p);
        hipLaunchKernelGGL(attention_bwd_dkv_kernel, dim3((N + KT - 1) / KT, heads, B), dim3(NT), 0, st, p);
    }
    return dvs::check_launch("dvs_attention_bwd");
}

int dvs_layernorm_bwd(const float* x, const float* gamma, const float* dy, float* dx, float* dgamma_acc, float* dbeta_acc, int M, int C,
                      float eps, void* stream) {
    DVS_REQUIRE(x && gamma && dy && dx && dgamma_acc && dbeta_acc && M > 0 && C > 0 && (C & 3) == 0 && C <= 2048,
                "dvs_layernorm_bwd: C %% 4 == 0, C <= 2048 (got %d)", C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    int blocks = (M + NT / 64 - 1) / (NT / 64);
    blocks = blocks > 512 ? 512 : blocks;            // <= 2048 row-walking wavefronts: few atomics per channel
    const dim3 grid(blocks);
    if (C <= 256) hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, dim3(NT), 0, st, x, gamma, dy, dx, dgamma_acc, dbeta_acc, M, C, eps);
    else if (C <= 512) hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, dim3(NT), 0, st, x, gamma, dy, dx, dgamma_acc, dbeta_acc, M, C, eps);
    else if (C <= 1024) hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, dim3(NT), 0, st, x, gamma, dy, dx, dgamma_acc, dbeta_acc, M, C, eps);
    else hipLaunchKernelGGL(layernorm_bwd_kernel<8>, grid, dim3(NT), 0, st, x, gamma, dy, dx, dgamma_acc, dbeta_acc, M, C, eps);
    return dvs::check_launch("dvs_layernorm_bwd");
}

int dvs_act_fwd(const float* x, float* y, size_t n, int act, void* stream) {
    DVS_REQUIRE(x && y && n > 0 && (n & 3) == 0 && (act == 1 || act == 4), "dvs_act_fwd: n %% 4 == 0, act 1 (ReLU) or 4 (GELU)");
    hipLaunchKernelGGL(act_fwd_kernel, dim3(sgrid(n / 4)), dim3(NT), 0, static_cast<hipStream_t>(stream), x, y, n / 4, act);
    return dvs::check_launch("dvs_act_fwd");
}

int dvs_act_bwd_in(const float* x, const float* dy, float* dx, size_t n, int act, void* stream) {
    DVS_REQUIRE(x && dy && dx && n > 0 && (n & 3) == 0 && (act == 1 || act == 4), "dvs_act_bwd_in: n %% 4 == 0, act 1 (ReLU) or 4 (GELU)");
    hipLaunchKernelGGL(act_bwd_in_kernel, dim3(sgrid(n / 4)), dim3(NT), 0, static_cast<hipStream_t>(stream), x, dy, dx, n / 4, act);
    return dvs::check_launch("dvs_act_bwd_in");
}

int dvs_resize_bilinear_ac_bwd(const float* dy, float* dx, int B, int h, int w, int H, int W, int C, void* stream) {
    DVS_REQUIRE(dy && dx && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0, "dvs_resize_bilinear_ac_bwd: bad argument");
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(dx, 0, (size_t)B * h * w * C * sizeof(float), st);
    if (e != hipSuccess) return dvs::fail(DVS_ERR_LAUNCH, "dvs_resize_bilinear_ac_bwd: memset: %s", hipGetErrorString(e));
    if (C & 3) {
        hipLaunchKernelGGL(resize_bilinear_ac_scalar_kernel, dim3(sgrid((size_t)B * H * W * C)), dim3(NT), 0, st, dy, static_cast<float*>(nullptr),
                           dx, B, h, w, H, W, C);
        return dvs::check_launch("dvs_resize_bilinear_ac_bwd");
    }
    const size_t n = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(resize_bilinear_ac_bwd_kernel, dim3(sgrid(n)), dim3(NT), 0, st, dy, dx, B, h, w, H, W, C);
    return dvs::check_launch("dvs_resize_bilinear_ac_bwd");
}

int dvs_deconv_unshuffle(const float* dy, float* dg, int B, int h, int w, int k, int Cout, void* stream) {
    DVS_REQUIRE(dy && dg && B > 0 && h > 0 && w > 0 && k > 0 && Cout > 0 && (Cout & 3) == 0, "dvs_deconv_unshuffle: bad argument");
    const size_t n = (size_t)B * h * k * w * k * (Cout / 4);
    hipLaunchKernelGGL(deconv_unshuffle_kernel, dim3(sgrid(n)), dim3(NT), 0, static_cast<hipStream_t>(stream), dy, dg, B, h, w, k, Cout);
    return dvs::check_launch("dvs_deconv_unshuffle");
}

}  // extern "C"
