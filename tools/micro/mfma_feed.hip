// Micro-benchmark: how fast can one CU's matrix pipes be fed?  hipcc --offload-arch=gfx950 -O3 -o mfma_feed mfma_feed.hip
//   A: MFMAs only (operands in registers)            B: ds_read_b32 operands, read -> use in the same step
//   C: ds_read_b32 operands prefetched one step ahead D: as C with ds_read_b128 (4 k-steps per read)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int STEPS = 256, ITERS = 64;

template <int VAR, int TN>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = (float)(i & 7) * 0.125f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    f32x16 acc[TN];
    for (int t = 0; t < TN; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    float a = lane * 0.001f, b[TN];
    for (int t = 0; t < TN; ++t) b[t] = (lane + t) * 0.002f;
    int iv[4] = {lane, lane + 1, lane + 2, lane + 3};
    for (int it = 0; it < iters; ++it) {
        const float* smi = sm + ((it * 36) & 1020);      // iteration-dependent base: the reads cannot be hoisted
        if (VAR == 0) {
#pragma unroll
            for (int s = 0; s < STEPS; ++s)
#pragma unroll
                for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
        } else if (VAR == 1) {
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                float av = smi[(s * 64 + lane) & 4095];
                float bv[TN];
#pragma unroll
                for (int t = 0; t < TN; ++t) bv[t] = smi[(s * 8 + t * 136 + lane + 2048) & 4095];
#pragma unroll
                for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
            }
        } else if (VAR == 2) {
            float av = smi[lane], bv[TN];
#pragma unroll
            for (int t = 0; t < TN; ++t) bv[t] = smi[(t * 136 + lane + 2048) & 4095];
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                float an = smi[((s + 1) * 64 + lane) & 4095], bn[TN];
#pragma unroll
                for (int t = 0; t < TN; ++t) bn[t] = smi[((s + 1) * 8 + t * 136 + lane + 2048) & 4095];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[t], acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                av = an;
#pragma unroll
                for (int t = 0; t < TN; ++t) bv[t] = bn[t];
            }
        } else if (VAR == 4 || VAR == 5) {
            // E: NV independent integer VALU instructions after every MFMA, same wave (does the wave's own VALU work
            // issue in the shadow of its MFMA?)
            constexpr int NV = VAR == 4 ? 4 : 12;
#pragma unroll
            for (int s = 0; s < STEPS; ++s)
#pragma unroll
                for (int t = 0; t < TN; ++t) {
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b[t], acc[t], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int q = 0; q < NV; ++q) {
                        asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(iv[q % 4]) : "v"(lane), "v"(it));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        } else {
#pragma unroll
            for (int s = 0; s < STEPS; s += 4) {
                f32x4 av = *reinterpret_cast<const f32x4*>(smi + ((s * 64 + lane * 4) & 4092));
                f32x4 bv[TN];
#pragma unroll
                for (int t = 0; t < TN; ++t) bv[t] = *reinterpret_cast<const f32x4*>(smi + ((s * 8 + t * 144 + lane * 4 + 2048) & 4092));
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int t = 0; t < TN; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[t][q], acc[t], 0, 0, 0);
            }
        }
    }
    float r = (float)(iv[0] + iv[1] + iv[2] + iv[3]);
    for (int t = 0; t < TN; ++t) for (int i = 0; i < 16; ++i) r += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int VAR, int TN>
void run(const char* name, int blocks_per_cu) {
    float* out;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<VAR, TN>), dim3(blocks), dim3(256), 0, 0, out, 4);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<VAR, TN>), dim3(blocks), dim3(256), 0, 0, out, ITERS);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * 4 * ITERS * STEPS * TN * 2.0 * 32 * 32 * 2;
    printf("%-28s TN=%d blocks/CU=%d  %8.3f ms  %7.1f TF\n", name, TN, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int bpc = 1; bpc <= 3; ++bpc) {
        run<0, 3>("A mfma only", bpc);
        run<1, 3>("B lds b32, read->use", bpc);
        run<2, 3>("C lds b32, prefetch 1 step", bpc);
        run<3, 3>("D lds b128, read->use", bpc);
        run<4, 3>("E mfma + 4 VALU each", bpc);
        run<5, 3>("E mfma + 12 VALU each", bpc);
        run<0, 4>("A mfma only", bpc);
        run<1, 4>("B lds b32, read->use", bpc);
        run<2, 4>("C lds b32, prefetch 1 step", bpc);
    }
    return 0;
}
