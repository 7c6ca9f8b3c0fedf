// LDS-DMA variant of the implicit-GEMM kernel (included by conv_fwd.hip after FwdParams / conv_epilogue).
//
// Staging through registers costs every stage ~100 VALU instructions (address math, padding selects) plus
// the ds_write pass; counters showed the matrix pipe only 58 % busy while a stage-free loop (LDS reads +
// MFMAs only) reached 100-115 TF.  Here the global -> LDS copy is done by the memory pipeline itself:
// `global_load_lds_dwordx4` writes 64 lanes x 16 B = 1 KB of LDS per instruction (wave-uniform base + lane*16,
// i.e. 8 tile rows of 32 floats), no VGPR destination, no ds_write, no padding select -- lanes whose pixel is
// padding or whose k is past the end fetch from a 16-byte zero page instead.
// Because the DMA destination is lane-linear the tile rows cannot carry the +16 B pad; bank conflicts of the
// ds_read_b128 fragment fetch are avoided by an XOR swizzle applied on the SOURCE side (which k-chunk a lane
// fetches) and again on the read: LDS slot s of row r holds k-chunk s ^ ((r >> 1) & 7), so the 16 rows a
// ds_read_b128 lane group touches map to 16 distinct 16-byte slots of the 256-byte bank row.
// Two LDS buffers, one barrier per stage: DMA(k+1) -> MFMA(k) -> s_waitcnt vmcnt(0) -> barrier.
// Used for NHWC / upsample+concat inputs and un-fused data gradients with Cin % 32 == 0 (the encoder's convs).
#pragma once

__device__ __attribute__((aligned(16))) float g_dvs_zero_page[4] = {0.f, 0.f, 0.f, 0.f};

__device__ __forceinline__ void dma16(const float* gp, float* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gp,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// MFMA pass over swizzled, un-padded tiles As[BM][32], Bs[BN][32]
template <int TM, int TN>
__device__ __forceinline__ void mfma_stage_swz(const float* __restrict__ As, const float* __restrict__ Bs, int a_row0,
                                               int b_row0, int lane, f32x16 (&acc)[TM][TN]) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            int row = a_row0 + m * 32 + r;
            a[m] = *reinterpret_cast<const f32x4*>(As + row * BK + (((2 * j + h) ^ ((row >> 1) & 7)) << 2));
        }
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            int row = b_row0 + n * 32 + r;
            b[n] = *reinterpret_cast<const f32x4*>(Bs + row * BK + (((2 * j + h) ^ ((row >> 1) & 7)) << 2));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][t], b[n][t], acc[m][n], 0, 0, 0);
    }
}

template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(NT) void conv_dma_kernel(FwdParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_INS = BM / 32, B_INS = BN / 32;          // DMA instructions per wave per stage (8 rows each)
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                // [2][BM][32]
    float* Bs = smem + 2 * BM * BK;                  // [2][BN][32]

    ConvShape s = p.s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    int lg = xcd_logical(blockIdx.x, p.g.x * p.g.y * p.g.z, p.g.remap);      // see conv_fwd_kernel
    const int bid_y = lg % p.g.y;
    lg /= p.g.y;
    const int bid_z = lg % p.g.z, bid_x = lg / p.g.z;
    const int m0 = bid_x * BM, n0 = bid_y * BN;

    int Hr = s.Ho, Wr = s.Wo, rstep = 1, oy0 = 0, ox0 = 0, ky0 = 0, kx0 = 0, kw_full = s.kw;
    if (MODE == IN_DGRAD && s.stride == 2) {         // parity classes, see conv_fwd_kernel
        const int py = bid_z >> 1, px = bid_z & 1;
        oy0 = (py - s.pad) & 1;
        ox0 = (px - s.pad) & 1;
        Hr = (s.Ho - oy0 + 1) >> 1;
        Wr = (s.Wo - ox0 + 1) >> 1;
        rstep = 2;
        ky0 = py;
        kx0 = px;
        s.kh = py < s.kh ? (s.kh - py + 1) >> 1 : 0;
        s.kw = px < s.kw ? (s.kw - px + 1) >> 1 : 0;
        s.Ktot = s.kh * s.kw * s.Cin;
    }
    const int M = s.B * Hr * Wr;
    if (m0 >= M) return;

    // my (row, slot) inside each 8-row DMA instruction, and the k-chunk that slot must receive
    const int rsub = lane >> 3, slot = lane & 7;
    int a_b[A_INS], a_iy[A_INS], a_ix[A_INS], a_q[A_INS];
    bool a_ok[A_INS];
#pragma unroll
    for (int j = 0; j < A_INS; ++j) {
        const int row = (wave * A_INS + j) * 8 + rsub;
        a_q[j] = (slot ^ ((row >> 1) & 7)) << 2;                   // float offset of my k-chunk inside the stage
        int m = m0 + row;
        a_ok[j] = m < M;
        m = min(m, M - 1);
        int b = m / (Hr * Wr), rem = m - b * (Hr * Wr);
        int oy = rem / Wr, ox = rem - oy * Wr;
        a_b[j] = b;
        if (MODE == IN_DGRAD) {
            a_iy[j] = oy * rstep + oy0 + s.pad;
            a_ix[j] = ox * rstep + ox0 + s.pad;
        } else {
            a_iy[j] = oy * s.stride - s.pad;
            a_ix[j] = ox * s.stride - s.pad;
        }
    }
    const float* b_ptr[B_INS];
    int b_q[B_INS];
    bool b_ok[B_INS];
#pragma unroll
    for (int j = 0; j < B_INS; ++j) {
        const int row = (wave * B_INS + j) * 8 + rsub;
        b_q[j] = (slot ^ ((row >> 1) & 7)) << 2;
        int n = n0 + row;
        b_ok[j] = n < s.Cout;
        b_ptr[j] = p.w + (size_t)min(n, s.Cout - 1) * p.s.Ktot;
    }

    // the stage's (tap, channel base) is workgroup-uniform because Cin % 32 == 0
    int kt_k = 0, ci0 = 0, tky = 0, tkx = 0, cur_tap = -1;
    int t_off[A_INS], t_off2[A_INS];
    bool t_ok[A_INS];
    auto issue_stage = [&](int buf) {
        const bool k_ok = kt_k < s.Ktot;
        const int ky = ky0 + rstep * tky, kx = kx0 + rstep * tkx;
        const int tap = ky * kw_full + kx;
        if (tap != cur_tap) {
            cur_tap = tap;
#pragma unroll
            for (int j = 0; j < A_INS; ++j) {
                bool ok = a_ok[j];
                if (MODE == IN_DGRAD) dgrad_tap_setup(p.s, a_b[j], a_iy[j], a_ix[j], ky, kx, ok, t_off[j]);
                else tap_setup<MODE>(s, p.t, a_b[j], a_iy[j] + tky, a_ix[j] + tkx, ok, t_off[j], t_off2[j]);
                t_ok[j] = ok;
            }
        }
#pragma unroll
        for (int j = 0; j < A_INS; ++j) {
            const int ci = ci0 + a_q[j];
            const float* gp;
            if (MODE == IN_UPCAT && ci >= p.t.C1) gp = p.t.x2 + (t_off2[j] + ci);
            else gp = p.x + (t_off[j] + ci);
            if (!(t_ok[j] && k_ok)) gp = g_dvs_zero_page;
            dma16(gp, As + (buf * BM + (wave * A_INS + j) * 8) * BK);
        }
        const int kc = (MODE == IN_DGRAD) ? tap * s.Cin + ci0 : kt_k;        // column in the full weight row
#pragma unroll
        for (int j = 0; j < B_INS; ++j) {
            const float* gp = b_ptr[j] + kc + b_q[j];
            if (!(b_ok[j] && k_ok)) gp = g_dvs_zero_page;
            dma16(gp, Bs + (buf * BN + (wave * B_INS + j) * 8) * BK);
        }
        // advance to the next stage
        kt_k += BK;
        ci0 += BK;
        if (ci0 >= s.Cin) {
            ci0 = 0;
            if (++tkx == s.kw) {
                tkx = 0;
                ++tky;
            }
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
    issue_stage(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my LDS-DMA writes have landed ...
    __syncthreads();                                       // ... and so have everyone else's
#pragma unroll 1
    for (int kt = 0; kt < KT; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < KT) issue_stage(buf ^ 1);          // lands in the other buffer while this one is multiplied
        mfma_stage_swz<TM, TN>(As + buf * BM * BK, Bs + buf * BN * BK, wm * TM * 32, wn * TN * 32, lane, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    conv_epilogue<TM, TN, MODE>(p, s, acc, m0, n0, wm, wn, lane, M, Hr, Wr, rstep, oy0, ox0);
}
