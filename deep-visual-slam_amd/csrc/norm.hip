// a1: training-mode BatchNorm2d (+ residual add, + ReLU) of the ResNet encoder on NHWC activations.
//
// The batch statistics come for free from the convolution epilogue (per-channel sum / sum of squares,
// conv_fwd.hip); what is left of nn.BatchNorm2d is
//   dvs_bn_finalize      C lanes: mean / inv-std, folded scale & shift, running-stat update
//   dvs_bn_apply_fwd     one HBM pass: z = relu(y * scale + shift [+ residual or + its own folded BN])
//   dvs_bn_bwd_reduce    one pass: du = dz * [z > 0], per-channel sum(du), sum(du * xhat)
//   dvs_bn_bwd_apply     one pass: dy = gamma * invstd * (du - mean(du) - xhat * mean(du * xhat))
// replacing the BasicBlock tail bn -> (+identity) -> relu of torchvision's ResNet as used by
// model/resnet_encoder.py:100-111 (about 10 eager kernels per block, forward + backward).
// All kernels are HBM-bound: 16-byte accesses, a lane owns 4 consecutive channels of a pixel.
#include "common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int NT = 256;

// groups > 1: stats is [G][2][C], the outputs are rows of a [G][4][C] table (scale, shift, mean, invstd); the running
// statistics receive the G momentum updates in order -- what G successive forward calls of the module would have done.
__global__ void bn_finalize_kernel(const float* __restrict__ stats, float count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ running_mean,
                                   float* __restrict__ running_var, float momentum, float eps,
                                   float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ mean_out,
                                   float* __restrict__ invstd_out, int C, long long* __restrict__ num_batches_tracked,
                                   int groups, int slots) {
    int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const size_t slot_stride = (size_t)groups * 2 * C;     // stats = [slots][G][2][C]: the copies are added up here
    if (c == 0 && num_batches_tracked) *num_batches_tracked += groups;
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    float rm = running_mean ? running_mean[c] : 0.f, rv = running_mean ? running_var[c] : 0.f;
    for (int grp = 0; grp < groups; ++grp) {
        const float* st = stats + (size_t)grp * 2 * C;
        const size_t o = (size_t)grp * 4 * C + c;
        float a0[16], a1[16];                               // every copy's load in flight before the first add
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const size_t o = (k < slots ? k : 0) * slot_stride;
            a0[k] = st[o + c];
            a1[k] = st[o + C + c];
        }
        float s0 = a0[0], s1 = a1[0];
#pragma unroll
        for (int k = 1; k < 16; ++k)
            if (k < slots) {
                s0 += a0[k];
                s1 += a1[k];
            }
        for (int k = 16; k < slots; ++k) {
            s0 += st[k * slot_stride + c];
            s1 += st[k * slot_stride + C + c];
        }
        float mean = s0 / count;
        float var = fmaxf(s1 / count - mean * mean, 0.f);     // biased, as used for normalisation
        float invstd = rsqrtf(var + eps);
        scale[o] = g * invstd;
        shift[o] = b - mean * g * invstd;
        mean_out[o] = mean;
        invstd_out[o] = invstd;
        float unbiased = count > 1.f ? var * count / (count - 1.f) : var;
        rm = (1.f - momentum) * rm + momentum * mean;
        rv = (1.f - momentum) * rv + momentum * unbiased;
    }
    if (running_mean) {
        running_mean[c] = rm;
        running_var[c] = rv;
    }
}

// z = act(y * sc + sh [+ r * rsc + rsh]); n4 = number of float4 (M * C / 4); C % 4 == 0
__global__ __launch_bounds__(NT) void bn_apply_fwd_kernel(const float* __restrict__ y, const float* __restrict__ sc,
                                                          const float* __restrict__ sh, const float* __restrict__ r,
                                                          const float* __restrict__ rsc, const float* __restrict__ rsh,
                                                          float* __restrict__ z, size_t n4, int C, int relu) {
    {   // blockIdx.y = group: n4 vectors of rows per group, parameter rows 4C floats apart
        const size_t go = (size_t)blockIdx.y * n4 * 4, po = (size_t)blockIdx.y * 4 * C;
        y += go; z += go; sc += po; sh += po;
        if (r) r += go;
        if (rsc) { rsc += po; rsh += po; }
    }
    size_t stride = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride) {
        int c = (int)((i * 4) % C);
        f32x4 v = reinterpret_cast<const f32x4*>(y)[i];
        f32x4 s = *reinterpret_cast<const f32x4*>(sc + c), t = *reinterpret_cast<const f32x4*>(sh + c);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] * s[j] + t[j];
        if (r) {
            f32x4 u = reinterpret_cast<const f32x4*>(r)[i];
            if (rsc) {
                f32x4 s2 = *reinterpret_cast<const f32x4*>(rsc + c), t2 = *reinterpret_cast<const f32x4*>(rsh + c);
#pragma unroll
                for (int j = 0; j < 4; ++j) u[j] = u[j] * s2[j] + t2[j];
            }
            v += u;
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        reinterpret_cast<f32x4*>(z)[i] = v;
    }
}

// finalize + apply in one launch: every thread derives scale / shift of ITS 4 channels from the raw statistics (its
// channel vector is the same in every grid-stride iteration because C/4 divides the 256 threads), and the first
// workgroup also writes the [G][4][C] table the backward needs and updates the running statistics (G updates in order).
struct BnFusedArgs {
    const float* stats;      // [G][2][C]
    const float* gamma;
    const float* beta;
    float* running_mean;     // or NULL
    float* running_var;
    long long* nbt;          // or NULL
    float* fin;              // [G][4][C] out: scale, shift, mean, invstd
    float count, momentum, eps;
    int slots;               // stats = [slots][G][2][C] (copies the producing kernel spread its atomics over), added up on load
};
// Sums of the copies of one table entry pair (sum, sum of squares; C floats apart).  One copy: two plain loads.  Several: all loads
// are issued before the first add (a loop of load-then-add waits for every load in turn: 16 dependent L2 round trips, ~11 us, at
// the head of every workgroup of the BatchNorm kernel).
__device__ __forceinline__ void load_stat2(const float* __restrict__ st, int C, int slots, size_t slot_stride, f32x4& s0, f32x4& s1) {
    if (slots <= 1) {
        s0 = *reinterpret_cast<const f32x4*>(st);
        s1 = *reinterpret_cast<const f32x4*>(st + C);
        return;
    }
    f32x4 a[16], b[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const float* q = st + (k < slots ? k : 0) * slot_stride;
        a[k] = *reinterpret_cast<const f32x4*>(q);
        b[k] = *reinterpret_cast<const f32x4*>(q + C);
    }
#pragma unroll
    for (int k = 1; k < 16; ++k)
        if (k < slots) {
            a[0] += a[k];
            b[0] += b[k];
        }
    for (int k = 16; k < slots; ++k) {
        a[0] += *reinterpret_cast<const f32x4*>(st + k * slot_stride);
        b[0] += *reinterpret_cast<const f32x4*>(st + k * slot_stride + C);
    }
    s0 = a[0];
    s1 = b[0];
}
__global__ __launch_bounds__(NT) void bn_fwd_fused_kernel(BnFusedArgs a, const float* __restrict__ y, const float* __restrict__ r,
                                                          const float* __restrict__ rsc, const float* __restrict__ rsh,
                                                          float* __restrict__ z, size_t n4, int C, int relu, int groups) {
    const int grp = blockIdx.y;
    const int c = (int)(((size_t)threadIdx.x * 4) % C);
    if (blockIdx.x == 0 && grp == 0 && threadIdx.x < C / 4) {
        if (threadIdx.x == 0 && a.nbt) *a.nbt += groups;
        f32x4 rm = {0.f, 0.f, 0.f, 0.f}, rv = {0.f, 0.f, 0.f, 0.f};
        if (a.running_mean) {
            rm = *reinterpret_cast<const f32x4*>(a.running_mean + c);
            rv = *reinterpret_cast<const f32x4*>(a.running_var + c);
        }
        const f32x4 g = a.gamma ? *reinterpret_cast<const f32x4*>(a.gamma + c) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 b = a.beta ? *reinterpret_cast<const f32x4*>(a.beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        for (int q = 0; q < groups; ++q) {
            const float* st = a.stats + (size_t)q * 2 * C;
            float* out = a.fin + (size_t)q * 4 * C;
            f32x4 s0, s1;
            load_stat2(st + c, C, a.slots, (size_t)groups * 2 * C, s0, s1);
            f32x4 sc, sh, mu, is;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                mu[j] = s0[j] / a.count;
                const float var = fmaxf(s1[j] / a.count - mu[j] * mu[j], 0.f);
                is[j] = rsqrtf(var + a.eps);
                sc[j] = g[j] * is[j];
                sh[j] = b[j] - mu[j] * g[j] * is[j];
                const float unbiased = a.count > 1.f ? var * a.count / (a.count - 1.f) : var;
                rm[j] = (1.f - a.momentum) * rm[j] + a.momentum * mu[j];
                rv[j] = (1.f - a.momentum) * rv[j] + a.momentum * unbiased;
            }
            *reinterpret_cast<f32x4*>(out + c) = sc;
            *reinterpret_cast<f32x4*>(out + C + c) = sh;
            *reinterpret_cast<f32x4*>(out + 2 * C + c) = mu;
            *reinterpret_cast<f32x4*>(out + 3 * C + c) = is;
        }
        if (a.running_mean) {
            *reinterpret_cast<f32x4*>(a.running_mean + c) = rm;
            *reinterpret_cast<f32x4*>(a.running_var + c) = rv;
        }
    }
    // my channels' folded scale / shift for this group: derived once per workgroup by the first C/4 threads (with a slotted table
    // every thread adding up the copies itself cost +0.5 ms per step), read back from LDS by everyone
    __shared__ f32x4 s_sc[NT], s_sh[NT];
    f32x4 s, t;
    if (threadIdx.x < C / 4) {
        const float* st = a.stats + (size_t)grp * 2 * C;
        f32x4 s0, s1;
        load_stat2(st + c, C, a.slots, (size_t)groups * 2 * C, s0, s1);
        const f32x4 g = a.gamma ? *reinterpret_cast<const f32x4*>(a.gamma + c) : f32x4{1.f, 1.f, 1.f, 1.f};
        const f32x4 b = a.beta ? *reinterpret_cast<const f32x4*>(a.beta + c) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float mu = s0[j] / a.count;
            const float var = fmaxf(s1[j] / a.count - mu * mu, 0.f);
            const float is = rsqrtf(var + a.eps);
            s[j] = g[j] * is;
            t[j] = b[j] - mu * g[j] * is;
        }
        s_sc[threadIdx.x] = s;                 // thread i < C/4 holds channels 4i .. 4i+3 (c = 4i)
        s_sh[threadIdx.x] = t;
    }
    __syncthreads();
    s = s_sc[c / 4];
    t = s_sh[c / 4];
    f32x4 s2 = {1.f, 1.f, 1.f, 1.f}, t2 = {0.f, 0.f, 0.f, 0.f};
    if (rsc) {
        s2 = *reinterpret_cast<const f32x4*>(rsc + (size_t)grp * 4 * C + c);
        t2 = *reinterpret_cast<const f32x4*>(rsh + (size_t)grp * 4 * C + c);
    }
    const size_t go = (size_t)grp * n4;
    const f32x4* yv = reinterpret_cast<const f32x4*>(y) + go;
    const f32x4* rv4 = r ? reinterpret_cast<const f32x4*>(r) + go : nullptr;
    f32x4* zv = reinterpret_cast<f32x4*>(z) + go;
    const size_t stride = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride) {
        f32x4 v = yv[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = v[j] * s[j] + t[j];
        if (rv4) {
            f32x4 u = rv4[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) u[j] = u[j] * s2[j] + t2[j];
            v += u;
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        zv[i] = v;
    }
}

// One-launch-less BatchNorm backward (dvs_bn_bwd): instead of one partial row per workgroup and a second kernel that sums the rows,
// workgroup w adds its partial sums into copy w % BWD_SLOTS of a zero-filled [G][BWD_SLOTS][2][C] table (2048 workgroups -> 64
// same-address atomics per copy, issued while seven other workgroups of the CU stream), and every workgroup of the apply kernel adds
// the copies up at its head (C/4 threads, all loads of a round in flight together) -- the 40 sum kernels of a step (6 us + a launch
// gap each, inside the dependency chains) are gone.  The deterministic mode keeps the ordered two-kernel sum.
constexpr int BWD_SLOTS = 32;

__device__ __forceinline__ void slot_add(float* __restrict__ slots, int grp, int wg, int C, int c, const f32x4& a, const f32x4& b) {
    float* row = slots + ((size_t)grp * BWD_SLOTS + (wg % BWD_SLOTS)) * 2 * C;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        atomicAdd(row + c + j, a[j]);
        atomicAdd(row + C + c + j, b[j]);
    }
}

// Totals of group `grp` for the channels 4 * tid .. of the first C/4 threads -> LDS tot[2][C]; the caller synchronises.
__device__ __forceinline__ void slot_totals(const float* __restrict__ slots, int grp, int C, float* __restrict__ tot) {
    if ((int)threadIdx.x < C / 4) {
        const int c = threadIdx.x * 4;
        const float* base = slots + (size_t)grp * BWD_SLOTS * 2 * C;
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < BWD_SLOTS / 16; ++r) {
            f32x4 a[16], b[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                a[k] = *reinterpret_cast<const f32x4*>(base + (size_t)(r * 16 + k) * 2 * C + c);
                b[k] = *reinterpret_cast<const f32x4*>(base + (size_t)(r * 16 + k) * 2 * C + C + c);
            }
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                s0 += a[k];
                s1 += b[k];
            }
        }
        *reinterpret_cast<f32x4*>(tot + c) = s0;
        *reinterpret_cast<f32x4*>(tot + C + c) = s1;
    }
}

// Column reduction over pixels.  A workgroup owns rows [blockIdx.x * rows_per_block, ...): lane -> (row lane,
// 4-channel vector); sums[0][c] += sum du, sums[1][c] += sum du * xhat; optionally writes du.
__global__ __launch_bounds__(NT) void bn_bwd_reduce_kernel(const float* __restrict__ dz, const float* __restrict__ z,
                                                           const float* __restrict__ y, const float* __restrict__ mean,
                                                           const float* __restrict__ invstd, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, float* __restrict__ du_out,
                                                           float* __restrict__ sums, int M, int C, int rows_per_block,
                                                           float* __restrict__ slots) {
    extern __shared__ __attribute__((aligned(16))) float red[];        // [NT][8]
    {   // blockIdx.y = group: M rows each; the partial rows of group g follow those of group g - 1
        const size_t go = (size_t)blockIdx.y * M * C, po = (size_t)blockIdx.y * 4 * C;
        dz += go; y += go; mean += po; invstd += po;
        if (scale) { scale += po; shift += po; }
        if (z) z += go;
        if (du_out) du_out += go;
        sums += (size_t)blockIdx.y * gridDim.x * 2 * C;
    }
    const int cv = C / 4, tid = threadIdx.x;
    const int rl = tid / cv, c = (tid % cv) * 4, rlanes = NT / cv;     // cv = C/4 divides 256 (host-checked)
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
    {
        const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
        const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
        // ReLU mask recomputed from y (no residual): z = max(y * scale + shift, 0) in the forward kernel, the same expression here
        f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
        if (scale) {
            sc = *reinterpret_cast<const f32x4*>(scale + c);
            sh = *reinterpret_cast<const f32x4*>(shift + c);
        }
        constexpr int U = 4;                                            // rows in flight per lane
        for (int r = r0 + rl; r < r1; r += rlanes * U) {
            f32x4 g[U], zz[U], yy[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                               // issue every load first
                int ru = r + u * rlanes;
                ok[u] = ru < r1;
                size_t o = ((size_t)min(ru, r1 - 1) * cv) + (tid % cv);
                g[u] = reinterpret_cast<const f32x4*>(dz)[o];
                yy[u] = reinterpret_cast<const f32x4*>(y)[o];
                if (z) zz[u] = reinterpret_cast<const f32x4*>(z)[o];
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (!ok[u]) continue;
                size_t o = ((size_t)(r + u * rlanes) * cv) + (tid % cv);
                if (z) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[u][j] = zz[u][j] > 0.f ? g[u][j] : 0.f;
                } else if (scale) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) g[u][j] = yy[u][j] * sc[j] + sh[j] > 0.f ? g[u][j] : 0.f;
                }
                if (du_out) reinterpret_cast<f32x4*>(du_out)[o] = g[u];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    a[j] += g[u][j];
                    b[j] += g[u][j] * (yy[u][j] - mu[j]) * is[j];
                }
            }
        }
    }
    // reduce over the row lanes that share my channel vector
    float* mine = red + tid * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mine[j] = a[j];
        mine[4 + j] = b[j];
    }
    __syncthreads();
    if (rl == 0) {
        for (int k = 1; k < rlanes; ++k) {
            const float* o = red + (tid + k * cv) * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] += o[j];
                b[j] += o[4 + j];
            }
        }
        // one partial row per workgroup (same-address float atomics from ~2000 workgroups serialise at
        // ~85 ns each -- measured 6x the streaming time of this kernel -- so a second small kernel sums them)
        if (slots) {       // ... or the workgroups spread over BWD_SLOTS copies of the table, which the apply kernel adds up (dvs_bn_bwd)
            slot_add(slots, blockIdx.y, blockIdx.x, C, c, a, b);
            return;
        }
        float* row = sums + (size_t)blockIdx.x * 2 * C;
        *reinterpret_cast<f32x4*>(row + c) = a;
        *reinterpret_cast<f32x4*>(row + C + c) = b;
    }
}

// sums[j] += sum over the partial rows of a slice (gridDim.y slices -> gridDim.y atomics per address)
__global__ __launch_bounds__(NT) void bn_bwd_sum_partials_kernel(const float* __restrict__ partials, float* __restrict__ sums,
                                                                 int rows, int C2, int rows_per_slice) {
    int j = blockIdx.x * NT + threadIdx.x;
    if (j >= C2) return;
    partials += (size_t)blockIdx.z * rows * C2;         // blockIdx.z = group
    sums += (size_t)blockIdx.z * C2;
    int r0 = blockIdx.y * rows_per_slice, r1 = min(rows, r0 + rows_per_slice);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
        acc0 += partials[(size_t)r * C2 + j];
        acc1 += partials[(size_t)(r + 1) * C2 + j];
        acc2 += partials[(size_t)(r + 2) * C2 + j];
        acc3 += partials[(size_t)(r + 3) * C2 + j];
    }
    for (; r < r1; ++r) acc0 += partials[(size_t)r * C2 + j];
    if (r0 < r1) atomicAdd(sums + j, (acc0 + acc1) + (acc2 + acc3));
}

// dy = gamma * invstd * (du - sum_du / N - xhat * sum_du_xhat / N)
__global__ __launch_bounds__(NT) void bn_bwd_apply_kernel(const float* __restrict__ du, const float* __restrict__ y,
                                                          const float* __restrict__ mean, const float* __restrict__ invstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ sums,
                                                          float* __restrict__ dy, size_t n4, int C, float inv_count,
                                                          float* __restrict__ dgamma_acc, float* __restrict__ dbeta_acc,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ slots, float* __restrict__ sums_out) {
    __shared__ __attribute__((aligned(16))) float s_tot[2 * 1024];      // C <= 1024 (C/4 divides 256)
    {   // blockIdx.y = group
        const size_t go = (size_t)blockIdx.y * n4 * 4, po = (size_t)blockIdx.y * 4 * C;
        du += go; y += go; dy += go; mean += po; invstd += po;
        if (scale) { scale += po; shift += po; }
        if (sums) sums += (size_t)blockIdx.y * 2 * C;
    }
    if (slots) {                                        // dvs_bn_bwd: the totals come from the copies the reduce kernel added into
        slot_totals(slots, blockIdx.y, C, s_tot);
        __syncthreads();
        sums = s_tot;
        if (blockIdx.x == 0 && sums_out)
            for (int c = threadIdx.x; c < 2 * C; c += NT) sums_out[(size_t)blockIdx.y * 2 * C + c] = s_tot[c];
    }
    if (blockIdx.x == 0 && dgamma_acc) {               // gradient sink: d gamma / d beta added straight into .grad
        for (int c = threadIdx.x; c < C; c += NT) {     // (atomics: the groups of one launch add to the same parameters)
            atomicAdd(dbeta_acc + c, sums[c]);
            atomicAdd(dgamma_acc + c, sums[C + c]);
        }
    }
    size_t stride = (size_t)gridDim.x * NT;
    // a lane's four channels are the same in every grid-stride iteration (C/4 divides the 256 threads): per-channel values once
    const int c = (int)(((size_t)threadIdx.x * 4) % C);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(mean + c), is = *reinterpret_cast<const f32x4*>(invstd + c);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + c), s1 = *reinterpret_cast<const f32x4*>(sums + C + c);
    f32x4 ga = {1.f, 1.f, 1.f, 1.f};
    if (gamma) ga = *reinterpret_cast<const f32x4*>(gamma + c);
    f32x4 sc = {0.f, 0.f, 0.f, 0.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (scale) {
        sc = *reinterpret_cast<const f32x4*>(scale + c);
        sh = *reinterpret_cast<const f32x4*>(shift + c);
    }
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n4; i += stride) {
        f32x4 g = reinterpret_cast<const f32x4*>(du)[i], yy = reinterpret_cast<const f32x4*>(y)[i];
        if (scale) {                                    // `du` is dz: the ReLU mask is recomputed from y as in the reduction
#pragma unroll
            for (int j = 0; j < 4; ++j) g[j] = yy[j] * sc[j] + sh[j] > 0.f ? g[j] : 0.f;
        }
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float xh = (yy[j] - mu[j]) * is[j];
            o[j] = ga[j] * is[j] * (g[j] - s0[j] * inv_count - xh * s1[j] * inv_count);
        }
        reinterpret_cast<f32x4*>(dy)[i] = o;
    }
}

// ---- stem tail: relu(bn1(conv1(x))) -> MaxPool2d(3, 2, 1) without the tensor in between (model/resnet_encoder.py:102-104) ----
// Forward: one lane per (pooled pixel, 4 channels) reads the nine y values of its window, applies scale / shift / ReLU and keeps
// the first maximum (torch's scan order, as maxpool_fwd_kernel in pool.hip); z is never written unless a caller needs it (DepthNet's
// finest skip connection) -- then the lane also stores the 2x2 block of z its window owns (taps ky, kx in {1, 2}: every pixel
// has exactly one owner).  fin = [G][4][C] (scale, shift, mean, invstd); group = image / (B / G).
__global__ __launch_bounds__(NT) void bn_relu_maxpool_fwd_kernel(const float* __restrict__ y, const float* __restrict__ fin,
                                                                 float* __restrict__ z, float* __restrict__ pooled,
                                                                 unsigned* __restrict__ idx, int B, int H, int W, int C, int Ho, int Wo,
                                                                 int per_group) {
    const int cv = C >> 2;
    const size_t n = (size_t)B * Ho * Wo * cv;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += (size_t)gridDim.x * NT) {
        const int c4 = (int)(i % cv);
        size_t pix = i / cv;
        const int ox = (int)(pix % Wo);
        pix /= Wo;
        const int oy = (int)(pix % Ho), b = (int)(pix / Ho);
        const float* tab = fin + (size_t)(b / per_group) * 4 * C + c4 * 4;
        const f32x4 sc = *reinterpret_cast<const f32x4*>(tab), sh = *reinterpret_cast<const f32x4*>(tab + C);
        f32x4 best = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        unsigned bi = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const int ky = t / 3, kx = t % 3;
            const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
            if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) {
                const size_t o = (((size_t)b * H + iy) * W + ix) * C + c4 * 4;
                f32x4 v = *reinterpret_cast<const f32x4*>(y + o);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = v[j] * sc[j] + sh[j];
                    v[j] = a != a ? a : fmaxf(a, 0.f);              // NaN propagates, as in torch's relu / max_pool2d
                }
                if (z && ky >= 1 && kx >= 1) *reinterpret_cast<f32x4*>(z + o) = v;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (v[j] > best[j] || v[j] != v[j]) {
                        best[j] = v[j];
                        bi = (bi & ~(0xffu << (8 * j))) | ((unsigned)t << (8 * j));
                    }
                }
            }
        }
        *reinterpret_cast<f32x4*>(pooled + i * 4) = best;
        idx[i] = bi;
    }
}

// Backward of the stem tail.  A lane owns one 2x2 block of input pixels (the block the pooled pixel (oy, ox) "owns") and four
// channels: the four windows that can reach those pixels -- (oy..oy+1, ox..ox+1) -- are loaded once (4 tap words, 4 gradient
// vectors) for all four pixels, every load is unconditional from a clamped address (all in flight together), and pixel (a, c) of
// the block receives window (a', c') iff a' <= a, c' <= c and its winning tap is ((a + 1 - 2a') * 3 + (c + 1 - 2c')).
struct PoolBlock {
    f32x4 g[4];       // masked dz of the pixels (a, c) = (0,0) (0,1) (1,0) (1,1); zero where the pixel does not exist
    f32x4 y[4];
    size_t o[4];      // float4 index of the pixel in y / dy
    bool ok[4];
};

__device__ __forceinline__ void pool_block_load(PoolBlock& k, const float* __restrict__ dpool, const unsigned* __restrict__ idx,
                                                const float* __restrict__ extra, const float* __restrict__ y, f32x4 sc, f32x4 sh,
                                                int b, int oy, int ox, int cq, int cv, int H, int W, int Ho, int Wo) {
    unsigned w[4];
    f32x4 d[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int wy = oy + (q >> 1), wx = ox + (q & 1);
        const bool wok = wy < Ho && wx < Wo;
        const size_t o = (((size_t)b * Ho + min(wy, Ho - 1)) * Wo + min(wx, Wo - 1)) * cv + cq;
        w[q] = wok ? idx[o] : 0xffffffffu;                  // tap 255 never matches
        d[q] = reinterpret_cast<const f32x4*>(dpool)[o];
    }
    f32x4 e[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int iy = 2 * oy + (q >> 1), ix = 2 * ox + (q & 1);
        k.ok[q] = iy < H && ix < W;
        k.o[q] = (((size_t)b * H + min(iy, H - 1)) * W + min(ix, W - 1)) * cv + cq;
        k.y[q] = reinterpret_cast<const f32x4*>(y)[k.o[q]];
        if (extra) e[q] = reinterpret_cast<const f32x4*>(extra)[k.o[q]];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int a = q >> 1, c = q & 1;
        f32x4 g = extra ? e[q] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int a2 = 0; a2 <= a; ++a2)
#pragma unroll
            for (int c2 = 0; c2 <= c; ++c2) {
                const unsigned t = (unsigned)((a + 1 - 2 * a2) * 3 + (c + 1 - 2 * c2));
                const unsigned ww = w[a2 * 2 + c2];
                const f32x4 dd = d[a2 * 2 + c2];
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (((ww >> (8 * j)) & 0xffu) == t) g[j] += dd[j];
            }
#pragma unroll
        for (int j = 0; j < 4; ++j) g[j] = (k.ok[q] && k.y[q][j] * sc[j] + sh[j] > 0.f) ? g[j] : 0.f;
        k.g[q] = g;
    }
}

// Pass 1 (the shape of bn_bwd_reduce_kernel, rows = 2x2 blocks): partial sums of dz and dz * xhat per workgroup.
__global__ __launch_bounds__(NT) void bn_pool_bwd_reduce_kernel(const float* __restrict__ dpool, const unsigned* __restrict__ idx,
                                                                const float* __restrict__ extra, const float* __restrict__ y,
                                                                const float* __restrict__ fin, float* __restrict__ partials, int Mb,
                                                                int C, int H, int W, int Ho, int Wo, int per_group,
                                                                int rows_per_block) {
    extern __shared__ __attribute__((aligned(16))) float red[];        // [NT][8]
    const int grp = blockIdx.y;
    fin += (size_t)grp * 4 * C;
    const int cv = C / 4, tid = threadIdx.x;
    const int rl = tid / cv, cq = tid % cv, c = cq * 4, rlanes = NT / cv;
    const int HWo = Ho * Wo, b0 = grp * per_group;
    const f32x4 sc = *reinterpret_cast<const f32x4*>(fin + c), sh = *reinterpret_cast<const f32x4*>(fin + C + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(fin + 2 * C + c), is = *reinterpret_cast<const f32x4*>(fin + 3 * C + c);
    f32x4 a = {0.f, 0.f, 0.f, 0.f}, bsum = {0.f, 0.f, 0.f, 0.f};
    const int r0 = blockIdx.x * rows_per_block, r1 = min(Mb, r0 + rows_per_block);
    for (int r = r0 + rl; r < r1; r += rlanes) {
        const int bl = r / HWo, rem = r - bl * HWo, oy = rem / Wo, ox = rem - oy * Wo;
        PoolBlock k;
        pool_block_load(k, dpool, idx, extra, y, sc, sh, b0 + bl, oy, ox, cq, cv, H, W, Ho, Wo);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] += k.g[q][j];
                bsum[j] += k.g[q][j] * (k.y[q][j] - mu[j]) * is[j];
            }
    }
    float* mine = red + tid * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mine[j] = a[j];
        mine[4 + j] = bsum[j];
    }
    __syncthreads();
    if (rl == 0) {
        for (int k2 = 1; k2 < rlanes; ++k2) {
            const float* o = red + (tid + k2 * cv) * 8;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a[j] += o[j];
                bsum[j] += o[4 + j];
            }
        }
        slot_add(partials, grp, blockIdx.x, C, c, a, bsum);      // `partials` = the zero-filled [G][BWD_SLOTS][2][C] table
    }
}

// Pass 2: dy = gamma * invstd * (dz - sum_dz / N - xhat * sum_dz_xhat / N) for the four pixels of the block.
__global__ __launch_bounds__(NT) void bn_pool_bwd_apply_kernel(const float* __restrict__ dpool, const unsigned* __restrict__ idx,
                                                               const float* __restrict__ extra, const float* __restrict__ y,
                                                               const float* __restrict__ fin, const float* __restrict__ gamma,
                                                               const float* __restrict__ sums, float* __restrict__ dy, int Mb, int C,
                                                               int H, int W, int Ho, int Wo, int per_group, float inv_count,
                                                               float* __restrict__ dgamma_acc, float* __restrict__ dbeta_acc,
                                                               float* __restrict__ sums_out) {
    __shared__ __attribute__((aligned(16))) float s_tot[2 * 1024];
    const int grp = blockIdx.y;
    fin += (size_t)grp * 4 * C;
    slot_totals(sums, grp, C, s_tot);                   // `sums` = the slot table the reduce kernel added into
    __syncthreads();
    if (blockIdx.x == 0 && sums_out)
        for (int c = threadIdx.x; c < 2 * C; c += NT) sums_out[(size_t)grp * 2 * C + c] = s_tot[c];
    sums = s_tot;
    if (blockIdx.x == 0 && dgamma_acc) {
        for (int c = threadIdx.x; c < C; c += NT) {
            atomicAdd(dbeta_acc + c, sums[c]);
            atomicAdd(dgamma_acc + c, sums[C + c]);
        }
    }
    const int cv = C / 4, HWo = Ho * Wo, b0 = grp * per_group;
    const int cq = threadIdx.x % cv, c = cq * 4;            // cv divides NT: a lane's channels are the same in every iteration
    const f32x4 sc = *reinterpret_cast<const f32x4*>(fin + c), sh = *reinterpret_cast<const f32x4*>(fin + C + c);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(fin + 2 * C + c), is = *reinterpret_cast<const f32x4*>(fin + 3 * C + c);
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sums + c), s1 = *reinterpret_cast<const f32x4*>(sums + C + c);
    f32x4 ga = {1.f, 1.f, 1.f, 1.f};
    if (gamma) ga = *reinterpret_cast<const f32x4*>(gamma + c);
    const size_t n = (size_t)Mb * cv, stride = (size_t)gridDim.x * NT;
    for (size_t i = (size_t)blockIdx.x * NT + threadIdx.x; i < n; i += stride) {
        const int r = (int)(i / cv);
        const int bl = r / HWo, rem = r - bl * HWo, oy = rem / Wo, ox = rem - oy * Wo;
        PoolBlock k;
        pool_block_load(k, dpool, idx, extra, y, sc, sh, b0 + bl, oy, ox, cq, cv, H, W, Ho, Wo);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float xh = (k.y[q][j] - mu[j]) * is[j];
                o[j] = ga[j] * is[j] * (k.g[q][j] - s0[j] * inv_count - xh * s1[j] * inv_count);
            }
            if (k.ok[q]) reinterpret_cast<f32x4*>(dy)[k.o[q]] = o;
        }
    }
}

inline unsigned stream_grid(size_t n4) {
    size_t b = (n4 + NT - 1) / NT;
    return (unsigned)(b > 2048 ? 2048 : (b == 0 ? 1 : b));     // 256 CUs x 8 blocks, grid-stride the rest
}

}  // namespace

extern "C" {

int dvs_bn_finalize(const float* stats, double count, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                    float* invstd, int C, long long* num_batches_tracked, int groups, void* stream) {
    return dvs_bn_finalize_slots(stats, 1, count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, invstd, C,
                                 num_batches_tracked, groups, stream);
}

int dvs_bn_finalize_slots(const float* stats, int stat_slots, double count, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, float momentum, float eps, float* scale, float* shift, float* mean,
                          float* invstd, int C, long long* num_batches_tracked, int groups, void* stream) {
    DVS_REQUIRE(stats && scale && shift && mean && invstd && C > 0 && count >= 1 && groups >= 1 && stat_slots >= 1,
                "dvs_bn_finalize: bad argument");
    DVS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "dvs_bn_finalize: running stats come together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 63) / 64), dim3(64), 0, static_cast<hipStream_t>(stream), stats,
                       (float)count, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, mean, invstd, C,
                       num_batches_tracked, groups, stat_slots);
    return dvs::check_launch("dvs_bn_finalize");
}

int dvs_bn_fwd(const float* y, const float* stats, double count, const float* gamma, const float* beta,
               float* running_mean, float* running_var, float momentum, float eps, long long* num_batches_tracked,
               float* fin, const float* residual, const float* res_scale, const float* res_shift, float* z, size_t M,
               int C, int relu, int groups, void* stream) {
    return dvs_bn_fwd_slots(y, stats, 1, count, gamma, beta, running_mean, running_var, momentum, eps, num_batches_tracked, fin,
                            residual, res_scale, res_shift, z, M, C, relu, groups, stream);
}

int dvs_bn_fwd_slots(const float* y, const float* stats, int stat_slots, double count, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, long long* num_batches_tracked,
                     float* fin, const float* residual, const float* res_scale, const float* res_shift, float* z, size_t M,
                     int C, int relu, int groups, void* stream) {
    DVS_REQUIRE(y && stats && fin && z && M > 0 && C > 0 && (C & 3) == 0 && count >= 1 && groups >= 1 && stat_slots >= 1,
                "dvs_bn_fwd: bad argument");
    DVS_REQUIRE(C / 4 <= NT && (NT % (C / 4)) == 0, "dvs_bn_fwd: C/4 must divide 256 (C=%d)", C);
    DVS_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "dvs_bn_fwd: running stats come together");
    DVS_REQUIRE((res_scale == nullptr) == (res_shift == nullptr) && (!res_scale || residual),
                "dvs_bn_fwd: residual affine needs residual, scale and shift");
    BnFusedArgs a{stats, gamma, beta, running_mean, running_var, num_batches_tracked, fin, (float)count, momentum, eps, stat_slots};
    size_t n4 = M * C / 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dvs::ProfScope prof(dvs::SLOT_BN_FWD, st);
    hipLaunchKernelGGL(bn_fwd_fused_kernel, dim3(stream_grid(n4), groups), dim3(NT), 0, st, a, y, residual, res_scale, res_shift,
                       z, n4, C, relu, groups);
    return dvs::check_launch("dvs_bn_fwd");
}

int dvs_bn_apply_fwd(const float* y, const float* scale, const float* shift, const float* residual,
                     const float* res_scale, const float* res_shift, float* z, size_t M, int C, int relu, int groups,
                     void* stream) {
    DVS_REQUIRE(y && scale && shift && z && M > 0 && C > 0 && (C & 3) == 0 && groups >= 1, "dvs_bn_apply_fwd: bad argument");
    DVS_REQUIRE((res_scale == nullptr) == (res_shift == nullptr) && (!res_scale || residual),
                "dvs_bn_apply_fwd: residual affine needs residual, scale and shift");
    size_t n4 = M * C / 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dvs::ProfScope prof(dvs::SLOT_BN_FWD, st);
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3(stream_grid(n4), groups), dim3(NT), 0, st, y, scale, shift, residual, res_scale,
                       res_shift, z, n4, C, relu);
    return dvs::check_launch("dvs_bn_apply_fwd");
}

namespace {
inline void bn_reduce_geometry(size_t M, int C, int* blocks, int* rpb) {
    const int rlanes = NT / (C / 4);
    int r = (int)((M + 2047) / 2048);                 // at most 2048 workgroups ...
    if (r < rlanes * 4) r = rlanes * 4;               // ... each with at least one unrolled pass per lane
    *rpb = r;
    *blocks = (int)((M + r - 1) / r);
}
}  // namespace

size_t dvs_bn_bwd_workspace(size_t M, int C, int groups) {
    if (M == 0 || C <= 0 || (C & 3) || C / 4 > NT || NT % (C / 4) || groups < 1) return 0;
    int blocks, rpb;
    bn_reduce_geometry(M, C, &blocks, &rpb);
    return (size_t)groups * blocks * 2 * C * sizeof(float);
}

static int dvs_bn_bwd_reduce_impl(const float* dz, const float* z, const float* y, const float* mean, const float* invstd,
                                  const float* scale, const float* shift, float* du, float* sums, float* workspace, size_t M, int C,
                                  int groups, void* stream);

int dvs_bn_bwd_reduce(const float* dz, const float* z, const float* y, const float* mean, const float* invstd,
                      float* du, float* sums, float* workspace, size_t M, int C, int groups, void* stream) {
    return dvs_bn_bwd_reduce_impl(dz, z, y, mean, invstd, nullptr, nullptr, du, sums, workspace, M, C, groups, stream);
}

int dvs_bn_bwd_reduce_ymask(const float* dz, const float* y, const float* mean, const float* invstd, const float* scale,
                            const float* shift, float* sums, float* workspace, size_t M, int C, int groups, void* stream) {
    DVS_REQUIRE(scale && shift, "dvs_bn_bwd_reduce_ymask: scale / shift of the forward pass are required");
    return dvs_bn_bwd_reduce_impl(dz, nullptr, y, mean, invstd, scale, shift, nullptr, sums, workspace, M, C, groups, stream);
}

static int dvs_bn_bwd_reduce_impl(const float* dz, const float* z, const float* y, const float* mean, const float* invstd,
                                  const float* scale, const float* shift, float* du, float* sums, float* workspace, size_t M, int C,
                                  int groups, void* stream) {
    DVS_REQUIRE(dz && y && mean && invstd && sums && workspace && M > 0 && C > 0 && (C & 3) == 0 && groups >= 1,
                "dvs_bn_bwd_reduce: bad argument");
    const int cv = C / 4;
    DVS_REQUIRE(cv <= NT && (NT % cv) == 0, "dvs_bn_bwd_reduce: C/4 must divide 256 (C=%d)", C);
    DVS_REQUIRE(M < 2147483648ull, "dvs_bn_bwd_reduce: too many rows");
    hipStream_t st = static_cast<hipStream_t>(stream);
    int blocks, rpb;
    bn_reduce_geometry(M, C, &blocks, &rpb);
    {
        dvs::ProfScope prof(dvs::SLOT_BN_BWD, st);
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)blocks, groups), dim3(NT), NT * 8 * sizeof(float), st, dz, z, y, mean,
                           invstd, scale, shift, du, workspace, (int)M, C, rpb, nullptr);
    }
    // one slice = one ordered sum per address (deterministic mode); 32 slices = 32 float atomics per address
    const int slices = (blocks >= 64 && !dvs::deterministic()) ? 32 : 1, rps = (blocks + slices - 1) / slices;
    hipLaunchKernelGGL(bn_bwd_sum_partials_kernel, dim3((2 * C + NT - 1) / NT, slices, groups), dim3(NT), 0, st, workspace, sums, blocks,
                       2 * C, rps);
    return dvs::check_launch("dvs_bn_bwd_reduce");
}

static int dvs_bn_bwd_apply_impl(const float* du, const float* y, const float* mean, const float* invstd, const float* gamma,
                                 const float* sums, float* dy, size_t M, int C, float* dgamma_acc, float* dbeta_acc, int groups,
                                 const float* scale, const float* shift, void* stream);

int dvs_bn_bwd_apply(const float* du, const float* y, const float* mean, const float* invstd, const float* gamma,
                     const float* sums, float* dy, size_t M, int C, float* dgamma_acc, float* dbeta_acc, int groups,
                     void* stream) {
    return dvs_bn_bwd_apply_impl(du, y, mean, invstd, gamma, sums, dy, M, C, dgamma_acc, dbeta_acc, groups, nullptr, nullptr, stream);
}

int dvs_bn_bwd_apply_ymask(const float* dz, const float* y, const float* mean, const float* invstd, const float* scale,
                           const float* shift, const float* gamma, const float* sums, float* dy, size_t M, int C, float* dgamma_acc,
                           float* dbeta_acc, int groups, void* stream) {
    DVS_REQUIRE(scale && shift, "dvs_bn_bwd_apply_ymask: scale / shift of the forward pass are required");
    return dvs_bn_bwd_apply_impl(dz, y, mean, invstd, gamma, sums, dy, M, C, dgamma_acc, dbeta_acc, groups, scale, shift, stream);
}

static int dvs_bn_bwd_apply_impl(const float* du, const float* y, const float* mean, const float* invstd, const float* gamma,
                                 const float* sums, float* dy, size_t M, int C, float* dgamma_acc, float* dbeta_acc, int groups,
                                 const float* scale, const float* shift, void* stream) {
    DVS_REQUIRE(du && y && mean && invstd && sums && dy && M > 0 && C > 0 && (C & 3) == 0 && groups >= 1, "dvs_bn_bwd_apply: bad argument");
    DVS_REQUIRE((dgamma_acc == nullptr) == (dbeta_acc == nullptr), "dvs_bn_bwd_apply: gradient sinks come together");
    DVS_REQUIRE(C / 4 <= NT && (NT % (C / 4)) == 0, "dvs_bn_bwd_apply: C/4 must divide 256 (C=%d)", C);
    size_t n4 = M * C / 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dvs::ProfScope prof(dvs::SLOT_BN_BWD, st);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n4), groups), dim3(NT), 0, st, du, y, mean, invstd, gamma, sums, dy, n4,
                       C, (float)(1.0 / (double)M), dgamma_acc, dbeta_acc, scale, shift, nullptr, nullptr);
    return dvs::check_launch("dvs_bn_bwd_apply");
}

int dvs_bn_bwd_slot_floats(int C, int groups) { return (C > 0 && groups > 0) ? groups * BWD_SLOTS * 2 * C : 0; }

int dvs_bn_bwd(const float* dz, const float* z, const float* y, const float* fin, int ymask, const float* gamma, float* du, float* dy,
               float* slot_table, float* sums_out, size_t M, int C, float* dgamma_acc, float* dbeta_acc, int groups, void* stream) {
    DVS_REQUIRE(dz && y && fin && dy && slot_table && M > 0 && C > 0 && (C & 3) == 0 && groups >= 1, "dvs_bn_bwd: bad argument");
    const int cv = C / 4;
    DVS_REQUIRE(cv <= NT && (NT % cv) == 0, "dvs_bn_bwd: C/4 must divide 256 (C=%d)", C);
    DVS_REQUIRE(M < 2147483648ull, "dvs_bn_bwd: too many rows");
    DVS_REQUIRE(!(ymask && (z || du)), "dvs_bn_bwd: the y-mask form neither reads z nor writes du");
    DVS_REQUIRE((dgamma_acc == nullptr) == (dbeta_acc == nullptr), "dvs_bn_bwd: gradient sinks come together");
    const float *scale = fin, *shift = fin + C, *mean = fin + 2 * C, *invstd = fin + 3 * C;     // rows of group 0; 4C per group
    hipStream_t st = static_cast<hipStream_t>(stream);
    int blocks, rpb;
    bn_reduce_geometry(M, C, &blocks, &rpb);
    {
        dvs::ProfScope prof(dvs::SLOT_BN_BWD, st);
        hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3((unsigned)blocks, groups), dim3(NT), NT * 8 * sizeof(float), st, dz, z, y, mean,
                           invstd, ymask ? scale : nullptr, ymask ? shift : nullptr, du, nullptr, (int)M, C, rpb, slot_table);
    }
    {
        const size_t n4 = M * C / 4;
        dvs::ProfScope prof(dvs::SLOT_BN_BWD, st);
        // the masked gradient: du when the reduce pass wrote it, else dz (masked again from y in the y-mask form, or unmasked: no ReLU)
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(n4), groups), dim3(NT), 0, st, du ? du : dz, y, mean, invstd, gamma,
                           nullptr, dy, n4, C, (float)(1.0 / (double)M), dgamma_acc, dbeta_acc, ymask ? scale : nullptr,
                           ymask ? shift : nullptr, slot_table, sums_out);
    }
    return dvs::check_launch("dvs_bn_bwd");
}

int dvs_bn_relu_maxpool_fwd(const float* y, const float* fin, float* z, float* pooled, unsigned char* idx, int B, int H, int W, int C,
                            int groups, void* stream) {
    DVS_REQUIRE(y && fin && pooled && idx && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && groups >= 1 && B % groups == 0,
                "dvs_bn_relu_maxpool_fwd: bad argument");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t n4 = (size_t)B * Ho * Wo * (C / 4);
    size_t blocks = (n4 + NT - 1) / NT;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dvs::ProfScope prof(dvs::SLOT_BN_FWD, st);
    hipLaunchKernelGGL(bn_relu_maxpool_fwd_kernel, dim3((unsigned)blocks), dim3(NT), 0, st, y, fin, z, pooled,
                       reinterpret_cast<unsigned*>(idx), B, H, W, C, Ho, Wo, B / groups);
    return dvs::check_launch("dvs_bn_relu_maxpool_fwd");
}

int dvs_bn_relu_maxpool_bwd(const float* dpool, const unsigned char* idx, const float* dz_extra, const float* y, const float* fin,
                            const float* gamma, float* sums, float* workspace, float* dy, int B, int H, int W, int C,
                            float* dgamma_acc, float* dbeta_acc, int groups, void* stream) {
    DVS_REQUIRE(dpool && idx && y && fin && sums && workspace && dy && B > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0 && groups >= 1 &&
                    B % groups == 0, "dvs_bn_relu_maxpool_bwd: bad argument");
    const int cv = C / 4;
    DVS_REQUIRE(cv <= NT && (NT % cv) == 0, "dvs_bn_relu_maxpool_bwd: C/4 must divide 256 (C=%d)", C);
    DVS_REQUIRE((dgamma_acc == nullptr) == (dbeta_acc == nullptr), "dvs_bn_relu_maxpool_bwd: gradient sinks come together");
    const size_t M = (size_t)(B / groups) * H * W;
    DVS_REQUIRE((size_t)B * H * W * cv < 2147483648ull, "dvs_bn_relu_maxpool_bwd: too many elements");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t Mb = (size_t)(B / groups) * Ho * Wo;     // 2x2 blocks per group (one per pooled pixel)
    hipStream_t st = static_cast<hipStream_t>(stream);
    int blocks, rpb;
    bn_reduce_geometry(Mb, C, &blocks, &rpb);
    const unsigned* ix = reinterpret_cast<const unsigned*>(idx);
    {
        dvs::ProfScope prof(dvs::SLOT_BN_BWD, st);
        hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3((unsigned)blocks, groups), dim3(NT), NT * 8 * sizeof(float), st, dpool, ix,
                           dz_extra, y, fin, workspace, (int)Mb, C, H, W, Ho, Wo, B / groups, rpb);
    }
    {
        dvs::ProfScope prof(dvs::SLOT_BN_BWD, st);
        hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(stream_grid(Mb * cv), groups), dim3(NT), 0, st, dpool, ix, dz_extra, y, fin,
                           gamma, workspace, dy, (int)Mb, C, H, W, Ho, Wo, B / groups, (float)(1.0 / (double)M), dgamma_acc, dbeta_acc,
                           sums);
    }
    return dvs::check_launch("dvs_bn_relu_maxpool_bwd");
}

}  // extern "C"
