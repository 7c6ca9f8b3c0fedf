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
};

__global__ __launch_bounds__(NT, 2) void attention_fwd_kernel(AttnParams p) {
    __shared__ float sO[4][HD][QT + 1];        // per wave: O^T (dims x queries); +1: conflict-free transposed read
    __shared__ float sM[4][QT], sL[4][QT];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, h = lane >> 5;
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
    for (int t = wave; t < ntiles; t += 4) {
        const int k0 = t * KT;
        // ---- S^T = K Q^T : A operand = key row k0 + r, dims 4 (2 j + h) .. + 3
        f32x4 kv[8];
        {
            const int ki = min(k0 + r, N - 1);
            const float* kp = K + (size_t)ki * row;
#pragma unroll
            for (int j = 0; j < 8; ++j) kv[j] = *reinterpret_cast<const f32x4*>(kp + 4 * (2 * j + h));
        }
        // V^T operand: A[m = dim][k = key]: lane (r, h) of step j needs V[k0 + 8 j + 4 h + u][32 dt + r], u = 0..3
        float vv[2][4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ki = min(k0 + 8 * j + 4 * h + u, N - 1);
                const float* vp = V + (size_t)ki * row;
                vv[0][j][u] = vp[r];
                vv[1][j][u] = vp[32 + r];
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
        float tmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int key = k0 + (i & 3) + 8 * (i >> 2) + 4 * h;
            s[i] = key < N ? s[i] : -INFINITY;
            tmax = fmaxf(tmax, s[i]);
        }
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

inline unsigned sgrid(size_t n) {
    size_t b = (n + NT - 1) / NT;
    return (unsigned)(b > 4096 ? 4096 : (b == 0 ? 1 : b));
}

}  // namespace

extern "C" {

int dvs_attention_fwd(const float* qkv, float* out, int B, int N, int heads, int head_dim, float scale, void* stream) {
    DVS_REQUIRE(qkv && out && B > 0 && N > 0 && heads > 0, "dvs_attention_fwd: bad argument");
    DVS_REQUIRE(head_dim == HD, "dvs_attention_fwd: head dimension 64 only (got %d)", head_dim);
    DVS_REQUIRE((double)B * N * 3 * heads * HD < 2147483648.0, "dvs_attention_fwd: qkv must have fewer than 2^31 elements");
    AttnParams p{qkv, out, B, N, heads, scale};
    hipStream_t st = static_cast<hipStream_t>(stream);
    {
        dvs::ProfScope prof(dvs::SLOT_ATTN, st);
        prof.work(4.0 * B * heads * (double)N * N * HD);      // q k^T and p v: 2 N^2 d flops each
        hipLaunchKernelGGL(attention_fwd_kernel, dim3((N + QT - 1) / QT, heads, B), dim3(NT), 0, st, p);
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
    DVS_REQUIRE(x && y && B > 0 && h > 0 && w > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "dvs_resize_bilinear_ac: bad argument");
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

}  // extern "C"
