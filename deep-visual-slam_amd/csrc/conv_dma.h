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
//
// VALU budget.  On this chip the fp32 MFMA shares the SIMD's vector ALU: every VALU instruction a wave -- any wave
// of the SIMD -- issues costs ~4.5 cycles of matrix throughput (tools/micro/mfma_feed.hip: 135 -> 105 -> 75 TF with
// 0 / 4 / 12 VALU instructions per MFMA, the same at 1, 2 and 3 waves per SIMD), so address arithmetic is not hidden
// behind the MFMAs, it is ADDED to them.  The gather offsets of a lane's rows depend on the tap only, so all of them
// (rows x up to 9 taps) are computed once in the prologue and kept in registers; the K loop is unrolled over the taps
// with a runtime loop over the 32-channel blocks inside, and a stage's issue is one add, one compare, one 64-bit
// shift-add and two selects per DMA row: ~40 VALU per stage instead of ~175.
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

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

template <int BM, int BN, int WM, int WN, int MODE>
__global__ __launch_bounds__(NT) void conv_dma_kernel(FwdParams p) {
    constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
    constexpr int A_INS = BM / 32, B_INS = BN / 32;          // DMA instructions per wave per stage (8 rows each)
    constexpr int MAXTAP = 9;                                // 3x3 (or fewer: 1x1, the parity classes of a stride-2 dgrad)
    static_assert(WM * WN == 4 && TM >= 1 && TN >= 1, "4 waves");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                                // [2][BM][32]
    float* Bs = smem + 2 * BM * BK;                  // [2][BN][32]

    ConvShape s = p.s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform: LDS destinations stay scalar
    const int wm = wave / WN, wn = wave % WN;
    // N tile fastest, then M tile: the tiles that share an im2col slice are neighbours on one XCD (xcd_logical).
    // The four parity classes of a stride-2 data gradient carry 1, 2, 2 and 4 taps: dealt round-robin in launch order
    // (M tile fastest, class slowest, no remap) they balance across the XCDs; contiguous ranges measured 1.6x slower.
    int bid_x, bid_y, bid_z, ks = 0;
    if (p.g.z > 1) {
        int lg = blockIdx.x;
        bid_x = lg % p.g.x;
        lg /= p.g.x;
        bid_y = lg % p.g.y;
        bid_z = lg / p.g.y;
    } else {
        // split-K launches (p.ksplit > 1, small-M inference layers): the K slices of one tile are launch neighbours
        const int tiles = p.g.x * p.g.y;
        int lg = blockIdx.x;
        if (p.ksplit > 1) {
            ks = lg % p.ksplit;
            lg /= p.ksplit;
        }
        lg = xcd_logical(lg, tiles, p.ksplit > 1 ? 0 : p.g.remap);
        bid_y = lg % p.g.y;
        bid_x = lg / p.g.y;
        bid_z = 0;
    }
    const int m0 = bid_x * BM, n0 = bid_y * BN;

    int Hr = s.Ho, Wr = s.Wo, rstep = 1, oy0 = 0, ox0 = 0, ky0 = 0, kx0 = 0;
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
    const int ntap = s.kh * s.kw, nC = s.Cin / BK;   // taps of this launch / class, 32-channel blocks per tap
    // my K slice: channel blocks [c_lo, c_hi) of every tap (the whole range unless this is a split-K launch)
    const int c_per = (nC + p.ksplit - 1) / max(p.ksplit, 1);
    const int c_lo = p.ksplit > 1 ? ks * c_per : 0, c_hi = p.ksplit > 1 ? min(nC, c_lo + c_per) : nC;

    // ---- prologue: per (DMA row, tap) element offsets (NO_TAP: padding, row past the end) ----------------------
    constexpr int NO_TAP = OOB_OFF;                   // byte offset beyond every tensor: the buffer load returns zeros
    const int rsub = lane >> 3, slot = lane & 7;
    int toff[A_INS][MAXTAP];                         // into p.x   (IN_UPCAT: the half-resolution source)
    int toff2[(MODE == IN_UPCAT) ? A_INS : 1][MAXTAP];   // into p.t.x2 (IN_UPCAT only)
    int r_b[A_INS], r_y[A_INS], r_x[A_INS], r_q[A_INS];
    bool r_ok[A_INS];
#pragma unroll
    for (int j = 0; j < A_INS; ++j) {
        const int row = (wave * A_INS + j) * 8 + rsub;
        r_q[j] = (slot ^ ((row >> 1) & 7)) << 2;                     // float offset of my k-chunk inside the stage
        int m = m0 + row;
        r_ok[j] = m < M;
        m = min(m, M - 1);
        const int b = m / (Hr * Wr), rem = m - b * (Hr * Wr);
        const int oy = rem / Wr, ox = rem - oy * Wr;
        r_b[j] = b;
        if (MODE == IN_DGRAD) {
            r_y[j] = oy * rstep + oy0 + s.pad;
            r_x[j] = ox * rstep + ox0 + s.pad;
        } else {
            r_y[j] = oy * s.stride - s.pad;
            r_x[j] = ox * s.stride - s.pad;
        }
    }
    int kc_tap[MAXTAP];                              // first weight column of each tap (uniform)
#pragma unroll
    for (int t = 0; t < MAXTAP; ++t) {
        kc_tap[t] = 0;
#pragma unroll
        for (int j = 0; j < A_INS; ++j) {
            toff[j][t] = NO_TAP;
            if (MODE == IN_UPCAT) toff2[j][t] = NO_TAP;
        }
        if (t < ntap) {                                              // workgroup-uniform: unused taps cost nothing
            const int tky = (t >= s.kw ? 1 : 0) + (t >= 2 * s.kw ? 1 : 0), tkx = t - tky * s.kw;   // kw <= 3, t < 9
            kc_tap[t] = (MODE == IN_DGRAD) ? ((ky0 + rstep * tky) * p.s.kw + (kx0 + rstep * tkx)) * s.Cin : t * s.Cin;
#pragma unroll
            for (int j = 0; j < A_INS; ++j) {
                bool ok = r_ok[j];
                int off = 0, off2 = 0;
                if (MODE == IN_DGRAD) dgrad_tap_setup(p.s, r_b[j], r_y[j], r_x[j], ky0 + rstep * tky, kx0 + rstep * tkx, ok, off);
                else tap_setup<MODE>(s, p.t, r_b[j], r_y[j] + tky, r_x[j] + tkx, ok, off, off2);
                toff[j][t] = ok ? (off + r_q[j]) * 4 : NO_TAP;                              // bytes
                if (MODE == IN_UPCAT) toff2[j][t] = ok ? (off2 + p.t.C1 + r_q[j]) * 4 : NO_TAP;   // undo tap_setup's -C1 rebasing
            }
        }
    }
    int b_off[B_INS];                                // byte offset of my weight row + k-chunk (rows past Cout: out of range)
#pragma unroll
    for (int j = 0; j < B_INS; ++j) {
        const int row = (wave * B_INS + j) * 8 + rsub;
        const int n = n0 + row;
        b_off[j] = (n < s.Cout) ? (n * p.s.Ktot + ((slot ^ ((row >> 1) & 7)) << 2)) * 4 : NO_TAP;
    }
    // buffer descriptors (sizes in bytes; the host only selects this kernel for tensors < 2 GiB)
    const size_t x_elems = (MODE == IN_UPCAT) ? (size_t)s.B * (s.H >> 1) * (s.W >> 1) * p.t.C1 : (size_t)s.B * s.H * s.W * s.Cin;
    const __amdgpu_buffer_rsrc_t rx = dma_rsrc(p.x, x_elems * 4);
    const __amdgpu_buffer_rsrc_t rx2 = (MODE == IN_UPCAT && p.t.x2 != p.x)
                                           ? dma_rsrc(p.t.x2, (size_t)s.B * s.H * s.W * (s.Cin - p.t.C1) * 4) : rx;
    const __amdgpu_buffer_rsrc_t rw = dma_rsrc(p.w, (size_t)p.s.Cout * p.s.Ktot * 4);

    // stage (tap T, channel block c) -> LDS buffer `buf`.  T is a compile-time constant: toff[.][T] is a register.
    auto issue = [&](auto tc, int c, int buf) {
        constexpr int T = decltype(tc)::value;
        const int ci0 = c * BK;
        if (MODE == IN_UPCAT && ci0 >= p.t.C1) {                     // workgroup-uniform: the skip source
            const int soff = (ci0 - p.t.C1) * 4;
#pragma unroll
            for (int j = 0; j < A_INS; ++j) dma16_buf(rx2, toff2[j][T], soff, As + (buf * BM + (wave * A_INS + j) * 8) * BK);
        } else {
            const int soff = ci0 * 4;
#pragma unroll
            for (int j = 0; j < A_INS; ++j) dma16_buf(rx, toff[j][T], soff, As + (buf * BM + (wave * A_INS + j) * 8) * BK);
        }
        const int kc = (kc_tap[T] + ci0) * 4;
#pragma unroll
        for (int j = 0; j < B_INS; ++j) dma16_buf(rw, b_off[j], kc, Bs + (buf * BN + (wave * B_INS + j) * 8) * BK);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;

    int buf = 0;
    auto compute_and_sync = [&]() {
        mfma_stage_swz<TM, TN>(As + buf * BM * BK, Bs + buf * BN * BK, wm * TM * 32, wn * TN * 32, lane, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my LDS-DMA writes for the next stage have landed ...
        __syncthreads();                                       // ... and so have everyone else's
        buf ^= 1;
    };
    if (ntap > 0) {
        issue(std::integral_constant<int, 0>{}, c_lo, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    static_for<0, MAXTAP>([&](auto tc) {
        constexpr int T = decltype(tc)::value;
        if (T < ntap) {                                         // workgroup-uniform
#pragma unroll 1
            for (int c = c_lo; c + 1 < c_hi; ++c) {
                issue(tc, c + 1, buf ^ 1);                      // lands in the other buffer while this one is multiplied
                compute_and_sync();
            }
            if constexpr (T + 1 < MAXTAP) {
                if (T + 1 < ntap) issue(std::integral_constant<int, T + 1>{}, c_lo, buf ^ 1);
            }
            compute_and_sync();
        }
    });
    int* rowtab = nullptr;
    if (MODE == IN_DGRAD && rstep == 2) {
        rowtab = reinterpret_cast<int*>(smem);
        fill_row_table<BM>(p, s, rowtab, m0, M, Hr, Wr, rstep, oy0, ox0);
    }
    conv_epilogue<TM, TN, MODE>(p, s, acc, m0, n0, wm, wn, lane, M, Hr, Wr, rstep, oy0, ox0, rowtab, BM);
}
