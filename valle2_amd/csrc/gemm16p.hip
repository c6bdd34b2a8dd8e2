// Perf-mode tile GEMM, fourth form (round 6): ONE persistent workgroup per CU, 256 x 256 output tiles, 8 waves as two groups
// that take turns on the matrix pipe.
//
//   out = act(A W^T + bias) + residual, A (M,K) bf16 row-major, W (N,K) bf16 row-major, fp32 accumulators
//   (v_mfma_f32_32x32x16_bf16), outputs as bf16.hip: fp32 (+ fp32 residual) | bf16 (+ GELU) | the QKV scatter.
//
// Why (profiles/r5_probe_tile16.log): a 128^2 tile moves 32 KB through the CU's vector-memory path per 512
// MFMA cycles — no slack — and a K step of the two-slab form is the latency of its one slab in flight (0.8 us against
// 0.21 us of products).  A 256^2 tile halves the bytes per product; what it needs is (a) operand traffic that stays in
// flight ACROSS barriers and tiles, and (b) something on the matrix pipe while a wave reads LDS.
//
// Structure (after cdna_hip_programming.md section 5, "The 256^2 8-phase template", re-cut by measurement —
// tools/probe_p256.hip, profiles/r6_probe_p256.log):
//   * wave w = (wr = w >> 2, wc = w & 3) owns rows wr*128 .. +127, columns wc*64 .. +63 of the tile: 4 x 2 accumulator
//     blocks of 32 x 32 = 128 registers.  A K step of 64 (a "K-tile") is TWO phases of 16 MFMAs (512 cycles): phase 0 the
//     column half n0 (quadrants (m0,n0), (m1,n0)), phase 1 the half n1 ((m1,n1), (m0,n1)); m0/m1 = the wave's first / last
//     64 rows, n0/n1 its first / last 32 columns.  Operand registers: A(m0), A(m1) (32 each, read in phase 0), W(n1) and
//     the NEXT K-tile's W(n0) (16 each, read in phase 1): 16 / 8 ds_read_b128 per phase, 128 + 96 registers.
//     (First cut: four phases of 8 MFMAs.  Its load sections — up to 12 reads, two requests, the counted wait, two
//     barriers — took 330-400 cycles against 256 of MFMA: the pipe idled a third of the time.)
//   * A phase = [LDS reads | request two half-tiles by LDS-DMA | lgkmcnt(0), counted vmcnt] barrier [16 MFMAs] barrier.
//     Through a tile's K loop waves 4-7 run ONE BARRIER behind waves 0-3 (an extra s_barrier for them before it, one for
//     waves 0-3 after it): while one group of four (one wave per SIMD) multiplies, the other reads and requests.
//   * HALF-TILES: the 256 rows of a K-tile of A are staged as Aa = the m0 rows of both wave rows and Ab = the m1 rows; the
//     256 rows of W as Wa = the n0 columns of the four wave columns and Wb = the n1 columns (16 KB each: 128 rows x 128 B,
//     chunk c of row l at c ^ ((l >> 1) & 7) by swizzling the SOURCE address: the image of bf16.hip, conflict-free for
//     the 32x32x16 operand reads).  Phase 0 of K-tile T reads Aa(T), Ab(T) and requests Wb(T+1), Wa(T+2); phase 1 reads
//     Wb(T), Wa(T+1) and requests Aa(T+2), Ab(T+2): whatever a phase requests is READ THREE PHASES LATER and waited for in
//     the phase before that with vmcnt(8) — four half-tiles (64 KB) stay in flight across every barrier — and lands in a
//     buffer whose last reads were retired (lgkmcnt(0)) before the previous phase's first barrier (two buffers per kind:
//     128 KB).  The stream runs on across tile boundaries: the next tile's first slabs are requested while this tile's last
//     K-tile multiplies and its epilogue stores.
//   * EPILOGUE without a workgroup barrier, through a private 4 KB per wave of the remaining 32 KB of LDS.  bf16 results:
//     bias / GELU in the accumulator layout (a lane = a column: one bias value per lane), rows packed in pairs and written
//     COLUMN-major (8 bytes per lane), read back by ds_read_b64_tr_b16 — the hardware transpose hands a lane 4 consecutive
//     columns of one row — and stored 16 bytes per lane.  fp32 results: 32 x 32 blocks through a row-major fp32 image, the
//     residual's loads one block ahead.  (Stores straight from the accumulator layout — dwords, 128-byte row segments —
//     measured SLOWER: 10.8 k against 6.8 k cycles per bf16 tile; the cost is per store instruction.)
//     The stores sit in the same in-order vmcnt queue as the DMA: the phase after an epilogue waits with vmcnt(8 + stores
//     per wave); a ragged tile (rows beyond M) drains instead.
//
// Operand maps (cdna_hip_programming.md section 3), lane l: r = l & 31, h = l >> 5: A[i = r][k = 8h + j], B[k = 8h + j][col = r];
// D register x: row (x & 3) + 8 (x >> 2) + 4 h, column r.
#include <type_traits>

#include "bf16_common.h"

#define PK_HALF 16384                  // bytes of a half-tile (128 rows x 128 B)
#define PK_SCRATCH 131072              // byte offset of the epilogue scratch (8 waves x 4 KB)
#define PK_LDS 163840
// LDS map of the staging buffers, [half-tile kind][K-tile parity][16 KB]: Aa 0, Ab 32 K | Wa 64 K, Wb 96 K — the A kinds lie
// within the 16-bit offset field of a ds_read from one base register, the W kinds from a second
#define PK_AA 0
#define PK_AB 32768
#define PK_WA 65536
#define PK_WB 98304

#define PK_BAR()                                          \
    do {                                                  \
        __builtin_amdgcn_sched_barrier(0);                \
        asm volatile("s_barrier" ::: "memory");           \
        __builtin_amdgcn_sched_barrier(0);                \
    } while (0)

template <int N>
__device__ __forceinline__ void pk_vmwait() {
    static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one LDS-DMA piece (1 KiB): M0 = wave base + DST, source = scalar base + per-lane offset
template <int DST>
__device__ __forceinline__ void pk_dma(uint32_t wave_lds, uint32_t voff, const char* base) {
#ifdef PK_ABLATE_DMA                                        // tools/probe_p256.hip: timing-only build without the requests (results are wrong)
    asm volatile("" ::"s"(wave_lds), "v"(voff), "s"(base));
    return;
#endif
    asm volatile("s_add_u32 m0, %0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(wave_lds), "v"(voff), "s"(base), "n"(DST)
                 : "memory", "scc");
}

#ifdef VH_P256_PROBE
// tools/probe_p256.hip only: per wave, shader-clock sums of the segments of a phase (LDS reads + requests, reads waited for |
// counted wait | wait at the first barrier | MFMA section | wait at the second barrier) and of the epilogues, + wall-clock
// spans of the prologue and the kernel
__device__ unsigned long long vh_p256_probe[256 * 8 * 8];
#define PK_SEG(i)                                                                                       \
    do {                                                                                                \
        unsigned long long t_;                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                      \
        __builtin_amdgcn_sched_barrier(0);                                                              \
        pk_sum[i] += t_ - pk_last;                                                                      \
        pk_last = t_;                                                                                   \
    } while (0)
#else
#define PK_SEG(i) do { } while (0)
#endif

template <int OUT>
__global__ __launch_bounds__(512, 2) void gemm16_p256_kernel(Gemm16Args a, int tiles_m, int tiles_n) {
    __shared__ __attribute__((aligned(16))) char lds[PK_LDS];
    constexpr int NST = OUT == G16_F32 ? 32 : 16;           // epilogue stores per wave and tile
    const int tid = threadIdx.x;
    const int ws = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = ws >> 2, wc = ws & 3;
#ifdef VH_P256_PROBE
    unsigned long long pk_sum[6] = {0, 0, 0, 0, 0, 0}, pk_last = 0;
    const unsigned long long pk_wall0 = wall_clock64();
#endif

    // ---- this workgroup's tiles: each XCD (blocks b, b + 8, ... share one) gets a contiguous run, numbered n-fastest, so
    // the 32 CUs of an XCD work on neighbouring tiles of the same row panels at the same time
    const int ntiles = tiles_m * tiles_n, G = gridDim.x, bid = blockIdx.x;
    int first, stride, n_my;
    if ((G & 7) == 0) {
        const int xcd = bid & 7, idx = bid >> 3, gx = G >> 3;
        const int q8 = ntiles >> 3, r8 = ntiles & 7;
        const int start = xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8;
        const int cnt = q8 + (xcd < r8 ? 1 : 0);
        first = start + idx;
        stride = gx;
        n_my = idx < cnt ? (cnt - idx + gx - 1) / gx : 0;
    } else {
        first = bid;
        stride = G;
        n_my = bid < ntiles ? (ntiles - bid + G - 1) / G : 0;
    }
    if (n_my == 0) return;
    const int nk = a.K >> 6;                                 // K-tiles per tile: even, >= 4 (host check)
    const int NT = n_my * nk;                                // K-tiles of this workgroup's stream

    auto tile_mn = [&](int i, int& m0, int& n0) __attribute__((always_inline)) {
        const int t = first + i * stride;
        m0 = (t / tiles_n) * 256;
        n0 = (t - (t / tiles_n) * tiles_n) * 256;
    };

    // ---- LDS-DMA staging.  A half-tile = 16 pieces of 8 rows x 128 B (1 KiB per wave-instruction); wave w requests pieces
    // 2w, 2w + 1: local rows l = 16 w + 8 i + (lane >> 3).  Lane L fills slot (row L >> 3 of the piece, chunk L & 7) and so
    // fetches chunk (L & 7) ^ ((l >> 1) & 7) of the source row: Aa / Ab = panel row (l >> 6) * 128 + (l & 63) (+ 64),
    // Wa / Wb = weight row (l >> 5) * 64 + (l & 31) (+ 32).
    // Two cursors run ahead of the K-tile T being multiplied: LEAD = K-tile T + 2 (its Wa, Aa, Ab are requested during T) and
    // TRAIL = K-tile T + 1 (its Wb).  The A offsets depend on the tile (rows beyond M fetch the last row again), the W ones do not.
    uint32_t voA[2][2], voW[2][2];                           // [a | b][piece]
    int lead_i = 0, lead_kt = 0;
    const char* leadA;                                       // LEAD's tile: A / W panel bases
    const char* leadW;
    const char* trailW;                                      // TRAIL's W panel base + its k offset
    auto lead_tile = [&](int i) __attribute__((always_inline)) {
        int m0, n0;
        tile_mn(i, m0, n0);
        leadA = (const char*)(a.A + (int64_t)m0 * a.lda);
        leadW = (const char*)(a.W + (int64_t)n0 * a.K);
        const int rmax = a.M - 1 - m0;
        const int lane = tid & 63;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int l = 16 * ws + 8 * i2 + (lane >> 3);
            const int c = (lane & 7) ^ ((l >> 1) & 7);
            const int pr = (l >> 6) * 128 + (l & 63);
            voA[0][i2] = (uint32_t)(min(pr, rmax) * a.lda + 8 * c) * 2u;
            voA[1][i2] = (uint32_t)(min(pr + 64, rmax) * a.lda + 8 * c) * 2u;
        }
    };
    {
        const int lane = tid & 63;
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) {
            const int l = 16 * ws + 8 * i2 + (lane >> 3);
            const int c = (lane & 7) ^ ((l >> 1) & 7);
            const int wrow = (l >> 5) * 64 + (l & 31);
            voW[0][i2] = (uint32_t)(wrow * a.K + 8 * c) * 2u;
            voW[1][i2] = (uint32_t)((wrow + 32) * a.K + 8 * c) * 2u;
        }
    }
    lead_tile(0);
    const uint32_t wave_lds = (uint32_t)(uintptr_t)lds + ws * 2048;
    // request half-tile KIND (source panel base + k offset given) into the buffer of parity PAR
    auto stage = [&](auto kind_c, auto par_c, const char* base) __attribute__((always_inline)) {
        constexpr int KIND = decltype(kind_c)::value, PAR = decltype(par_c)::value;
        constexpr bool isA = KIND == PK_AA || KIND == PK_AB;
        constexpr int ab = (KIND == PK_AB || KIND == PK_WB) ? 1 : 0;
        pk_dma<KIND + PAR * PK_HALF>(wave_lds, isA ? voA[ab][0] : voW[ab][0], base);
        pk_dma<KIND + PAR * PK_HALF + 1024>(wave_lds, isA ? voA[ab][1] : voW[ab][1], base);
    };
    auto lead_advance = [&]() __attribute__((always_inline)) {      // TRAIL <- LEAD, LEAD <- the K-tile after it
        trailW = leadW + (int64_t)lead_kt * 128;
        if (++lead_kt == nk) {
            lead_kt = 0;
            if (++lead_i < n_my) lead_tile(lead_i);
        }
    };
    using K_AA = std::integral_constant<int, PK_AA>;
    using K_AB = std::integral_constant<int, PK_AB>;
    using K_WA = std::integral_constant<int, PK_WA>;
    using K_WB = std::integral_constant<int, PK_WB>;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;

    // ---- operand fragments: row (wr * 64 | wc * 32) + r (+ 32 for the second block) of a half-tile, chunk (2 s + h) ^ swz
    uint32_t offA[4], offW[4];
    {
        const int lane = tid & 63, r = lane & 31, h = lane >> 5, swz = (r >> 1) & 7;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            offA[s] = (uint32_t)((wr * 64 + r) * 128 + (((2 * s + h) ^ swz) << 4));
            offW[s] = (uint32_t)(PK_WA + (wc * 32 + r) * 128 + (((2 * s + h) ^ swz) << 4));
        }
    }
    bf16x8 fa0[2][4], fa1[2][4], fw0[4], fw1[4];
    auto readA = [&](bf16x8 (&f)[2][4], int off) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int s = 0; s < 4; ++s) f[mt][s] = __builtin_bit_cast(bf16x8, ldq(lds + offA[s] + (off + mt * 4096)));
    };
    auto readW = [&](bf16x8 (&f)[4], int off) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) f[s] = __builtin_bit_cast(bf16x8, ldq(lds + offW[s] + off));
    };

    // accumulators start from the tile's bias (a lane = a column of every block: one value per lane and block column), so the
    // epilogues add nothing
    f32x16 acc[4][2];
    // The NEXT tile's two bias values (one per lane and block column) are requested by hand-counted loads at the start of a tile's
    // LAST PAIR of K-tiles and waited for at the epilogue's start with vmcnt(8) (the two phases in between request eight pieces):
    // a compiler-counted load here ends in vmcnt(0) — its counter is the in-order one and it cannot see the DMA — and drained every
    // store and every request in flight at each tile boundary (r6 ISA audit).
    float nb0 = 0.f, nb1 = 0.f;
    auto bias_request = [&](int i) __attribute__((always_inline)) {
        if (OUT != G16_QKV && a.bias && i < n_my) {
            int m0, n0;
            tile_mn(i, m0, n0);
            const float* base = a.bias + n0 + wc * 64;
            const uint32_t vo = (uint32_t)(tid & 31) * 4u;
            // ("+v": the destinations stay the registers nb0 / nb1 already live in — a fresh "=v" value merged with the old one
            // after this branch made hipcc COPY the registers before the wait below, i.e. before the data had landed)
            asm volatile("global_load_dword %0, %2, %3\n\tglobal_load_dword %1, %2, %3 offset:128"
                         : "+v"(nb0), "+v"(nb1) : "v"(vo), "s"(base) : "memory");
        }
    };
    auto bias_wait = [&]() __attribute__((always_inline)) {           // (after the tile's last tile nothing was requested: harmless)
        if (OUT != G16_QKV && a.bias) asm volatile("s_waitcnt vmcnt(8)" : "+v"(nb0), "+v"(nb1)::"memory");
    };
    auto acc_fill = [&](float b0, float b1) __attribute__((always_inline)) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                acc[mt][0][e] = b0;
                acc[mt][1][e] = b1;
            }
    };
    if (OUT != G16_QKV && a.bias) {                           // the first tile's bias: nothing else is in flight yet
        bias_request(0);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(nb0), "+v"(nb1)::"memory");
    }
    acc_fill(nb0, nb1);
    auto quadrant = [&](int mb, int nt, bf16x8 (&fa)[2][4], bf16x8 (&fw_)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc[mb][nt] = VH_MFMA16(fa[0][s], fw_[s], acc[mb][nt]);
            acc[mb + 1][nt] = VH_MFMA16(fa[1][s], fw_[s], acc[mb + 1][nt]);
        }
    };

    // ---- the counted wait that closes a load section: everything the NEXT phase reads has landed (what was requested two phases
    // ago and earlier); what this phase and the previous one requested stays in flight — four half-tiles in the steady state
    // (vmcnt(8), the fast path: ONE scalar branch per phase), fewer at the end of the stream; in the phase after an epilogue
    // also that tile's NST stores, which sit in the same in-order queue
    auto wait_slow = [&](int inflight, bool post) __attribute__((always_inline)) {
        if (post) pk_vmwait<8 + NST>();                      // (a tile has >= 4 K-tiles: every request around an epilogue exists)
        else if (inflight >= 4) pk_vmwait<8>();
        else if (inflight == 3) pk_vmwait<6>();
        else if (inflight == 2) pk_vmwait<4>();
        else if (inflight == 1) pk_vmwait<2>();
        else pk_vmwait<0>();
    };

    // ---- one K-tile T = two phases
    auto ktile = [&](auto par_c, int T, bool post) __attribute__((always_inline)) {
        constexpr int par = decltype(par_c)::value;
        using PAR = std::integral_constant<int, par>;
        using PARN = std::integral_constant<int, par ^ 1>;
        const bool fast = !post && T + 2 < NT;
        const int e1 = T + 1 < NT ? 1 : 0, e2 = T + 2 < NT ? 1 : 0;
        // phase 0: A(m0), A(m1) of this K-tile | request Wb of K-tile T + 1, Wa of T + 2 | quadrants (m0, n0), (m1, n0)
        readA(fa0, PK_AA + par * PK_HALF);
        readA(fa1, PK_AB + par * PK_HALF);
        if (fast) {
            stage(K_WB{}, PARN{}, trailW);
            stage(K_WA{}, PAR{}, leadW + (int64_t)lead_kt * 128);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PK_SEG(5);
            pk_vmwait<8>();
        } else {
            if (e1) stage(K_WB{}, PARN{}, trailW);
            if (e2) stage(K_WA{}, PAR{}, leadW + (int64_t)lead_kt * 128);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PK_SEG(5);
            wait_slow(2 * e1 + e1 + e2, post);               // the previous phase asked for Aa, Ab of T + 1
        }
        PK_SEG(0);
        PK_BAR();
        PK_SEG(1);
        __builtin_amdgcn_s_setprio(1);
        quadrant(0, 0, fa0, fw0);
        quadrant(2, 0, fa1, fw0);
        __builtin_amdgcn_s_setprio(0);
        PK_SEG(2);
        PK_BAR();
        PK_SEG(3);
        // phase 1: W(n1) of this K-tile, W(n0) of the next | request Aa, Ab of K-tile T + 2 | quadrants (m1, n1), (m0, n1)
        readW(fw1, (PK_WB - PK_WA) + par * PK_HALF);
        readW(fw0, (PK_WA - PK_WA) + (par ^ 1) * PK_HALF);
        if (fast) {
            const char* la = leadA + (int64_t)lead_kt * 128;
            stage(K_AA{}, PAR{}, la);
            stage(K_AB{}, PAR{}, la);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PK_SEG(5);
            pk_vmwait<8>();
        } else {
            if (e2) {
                const char* la = leadA + (int64_t)lead_kt * 128;
                stage(K_AA{}, PAR{}, la);
                stage(K_AB{}, PAR{}, la);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PK_SEG(5);
            wait_slow(e1 + e2 + 2 * e2, false);
        }
        PK_SEG(0);
        if (e2) lead_advance();
        PK_BAR();
        PK_SEG(1);
        __builtin_amdgcn_s_setprio(1);
        quadrant(2, 1, fa1, fw1);
        quadrant(0, 1, fa0, fw1);
        __builtin_amdgcn_s_setprio(0);
        PK_SEG(2);
        PK_BAR();
        PK_SEG(3);
    };

    // ---- epilogue of one tile (no workgroup barrier: each wave owns 4 KB of scratch)
    char* scb = lds + PK_SCRATCH + ws * 4096;
    auto epilogue = [&](int ci) __attribute__((always_inline)) {
        int m0, n0;
        tile_mn(ci, m0, n0);
        int lane = tid & 63;                                 // opaque copy: the epilogue's address arithmetic must not be hoisted over
        asm volatile("" : "+v"(lane));                       // the main loop, whose registers are all spoken for
        const int r = lane & 31, h = lane >> 5;
        const int mw = m0 + wr * 128, nw = n0 + wc * 64;    // this wave's first row / column
        const bool full = m0 + 256 <= a.M;
        bias_wait();
        if constexpr (OUT == G16_F32) {
            // 32 x 32 blocks through a row-major fp32 image (float index row * 32 + (col ^ (((row >> 1) & 1) << 2))); store side:
            // lane = (row (lane >> 3) + 8 i, columns 4 (lane & 7) .. + 3): 8 rows x 128 B per instruction; the residual's four
            // loads of the NEXT block are requested before this one is transposed
            float* sc = (float*)scb;
            const int c4 = lane & 7, r8 = lane >> 3, g = (lane >> 4) & 1;
            float* out = (float*)a.out;
            f32x4 rv[2][4];
            auto rload = [&](int blk, f32x4 (&dst)[4]) __attribute__((always_inline)) {
                const int mt = blk >> 1, nt = blk & 1;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = mw + mt * 32 + r8 + 8 * i;
                    dst[i] = ld4(a.res + (int64_t)min(m, a.M - 1) * a.ldr + nw + nt * 32 + 4 * c4);
                }
            };
            if (a.res) rload(0, rv[0]);
#pragma unroll
            for (int blk = 0; blk < 8; ++blk) {
                const int mt = blk >> 1, nt = blk & 1;
                if (a.res && blk + 1 < 8) rload(blk + 1, rv[(blk + 1) & 1]);
                const int col = nw + nt * 32 + 4 * c4;
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const int row = (x & 3) + 8 * (x >> 2) + 4 * h;
                    sc[row * 32 + (r ^ (((x >> 1) & 1) << 2))] = acc[mt][nt][x];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = r8 + 8 * i, m = mw + mt * 32 + row;
                    f32x4 v = ld4(sc + row * 32 + ((c4 ^ g) << 2));
                    if (a.act == VH_ACT_GELU_ERF) {
                        const vh_f32x2 g0 = gelu_erf2(vh_f32x2{v.x, v.y}), g1 = gelu_erf2(vh_f32x2{v.z, v.w});
                        v = f32x4{g0.x, g0.y, g1.x, g1.y};
                    }
                    if (a.res) v += rv[blk & 1][i];
                    if (full || m < a.M) st4(out + (int64_t)m * a.ldo + col, v);
                }
            }
        } else {
            // 32 rows x 64 columns (both blocks of a row group) as a COLUMN-major bf16 image: column c = 64 bytes = eight 8-byte
            // slots of 4 rows, slot s of column c stored at s ^ ((c >> 1) & 7).  Write side: lane (r, h), register group q holds
            // rows 8 q + 4 h .. + 3 of column nt * 32 + r: slot 2 q + h.  Read side (ds_read_b64_tr_b16): in a group of 16 lanes,
            // lane 4 q' + p supplies (column c0 + q', slot rows / 4 + p) and lane i receives columns c0 .. c0 + 3 of row
            // rows + i: two reads = 8 consecutive columns of one row = the 16 bytes the lane stores.
            const int g16 = lane >> 4, li = lane & 15, qq = (lane >> 2) & 3, pp = lane & 3;
            // QKV scatter: a wave's 64 columns are one head of q, K or V (d_model % 256 == 0: a tile lies in one of the three)
            int which = 0, hc = nw;
            uint16_t* cbase = nullptr;
            float invT = 0.f;
            if constexpr (OUT == G16_QKV) {
                which = nw / a.d_model;
                hc = nw - which * a.d_model;
                if (which != 0) cbase = (which == 1 ? a.kc : a.vc) + (int64_t)(hc / VH_HEAD_DIM) * a.S_max * VH_HEAD_DIM;
                invT = 1.0f / (float)a.T;
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const int c = nt * 32 + r;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float v0 = acc[mt][nt][4 * q], v1 = acc[mt][nt][4 * q + 1];
                        float v2 = acc[mt][nt][4 * q + 2], v3 = acc[mt][nt][4 * q + 3];
                        if (OUT == G16_BF16 && a.act == VH_ACT_GELU_ERF) {
                            const vh_f32x2 g0 = gelu16_2(vh_f32x2{v0, v1}), g1 = gelu16_2(vh_f32x2{v2, v3});
                            v0 = g0.x; v1 = g0.y; v2 = g1.x; v3 = g1.y;
                        }
                        if (OUT == G16_QKV && which == 0) {      // (wave-uniform) q leaves pre-scaled: vh_common.h
                            v0 *= VH_Q16_PRESCALE; v1 *= VH_Q16_PRESCALE; v2 *= VH_Q16_PRESCALE; v3 *= VH_Q16_PRESCALE;
                        }
                        const u32x2 pk = {pack_bf16(v0, v1), pack_bf16(v2, v3)};
                        *reinterpret_cast<u32x2*>(scb + c * 64 + (((2 * q + h) ^ ((c >> 1) & 7)) << 3)) = pk;
                    }
                }
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int j = st >> 1, k = st & 1;       // rows 16 j .. + 15, columns 8 (g16 + 4 k) .. + 7
                    const int c0 = 8 * (g16 + 4 * k);
                    const int ca = c0 + qq, cb = c0 + 4 + qq;
                    const char* pa = scb + ca * 64 + (((4 * j + pp) ^ ((ca >> 1) & 7)) << 3);
                    const char* pb = scb + cb * 64 + (((4 * j + pp) ^ ((cb >> 1) & 7)) << 3);
                    const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                                   (s16x4 __attribute__((address_space(3)))*)pa));
                    const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                                                   (s16x4 __attribute__((address_space(3)))*)pb));
                    const u32x4 pk = {lo.x, lo.y, hi.x, hi.y};
                    const int row = mt * 32 + 16 * j + li, m = mw + row;
                    uint16_t* dst;
                    if (OUT == G16_QKV && which != 0) {
                        // row m = b T + t: b by a float reciprocal, fixed up (m < 2^24)
                        const int mm = min(m, a.M - 1);
                        int b = (int)((float)mm * invT);
                        int t = mm - b * a.T;
                        if (t < 0) { --b; t += a.T; } else if (t >= a.T) { ++b; t -= a.T; }
                        const int cl = a.cache_len ? a.cache_len[b] : 0;
                        dst = cbase + ((int64_t)b * a.n_heads * a.S_max + cl + t) * VH_HEAD_DIM + c0;
                    } else if (OUT == G16_QKV) {
                        dst = (uint16_t*)a.out + (int64_t)m * a.ldo + hc + c0;
                    } else {
                        dst = (uint16_t*)a.out + (int64_t)m * a.ldo + nw + c0;
                    }
                    if (full || m < a.M) stq(dst, pk);
                }
            }
        }
        // a ragged tile issued fewer than NST stores: empty the queue, so that the counted waits that follow stay true
        if (!full) pk_vmwait<0>();
        acc_fill(nb0, nb1);
    };

    // ---- prologue: K-tile 0 complete (Wa, Aa, Ab, Wb), Wa, Aa, Ab of K-tile 1; then LEAD = K-tile 2, TRAIL = K-tile 1
    stage(K_WA{}, P0{}, leadW);
    stage(K_AA{}, P0{}, leadA);
    stage(K_AB{}, P0{}, leadA);
    stage(K_WB{}, P0{}, leadW);
    lead_advance();                                          // (nk >= 4: K-tiles 1 and 2 are still tile 0's)
    stage(K_WA{}, P1{}, leadW + 128);
    stage(K_AA{}, P1{}, leadA + 128);
    stage(K_AB{}, P1{}, leadA + 128);
    lead_advance();
    pk_vmwait<8>();                                          // Wa, Aa, Ab of K-tile 0 have landed; Wb(0) and K-tile 1's three in flight
    PK_BAR();
    readW(fw0, 0);

    int T = 0;
#ifdef VH_P256_PROBE
    const unsigned long long pk_wall1 = wall_clock64();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pk_last)::"memory");
#endif
#pragma unroll 1
    for (int ci = 0; ci < n_my; ++ci) {
        // waves 4-7 run one barrier behind through the tile's K loop and are waited for at its end, so that BOTH groups
        // store their results at the same time (staggered epilogues run one after the other: each group sits at a barrier
        // through the other's — 2 x 5.4 k cycles per bf16 tile, profiles/r6_probe_p256.log)
        if (wr == 1) PK_BAR();
#pragma unroll 1
        for (int kt = 0; kt < nk; kt += 2) {
            if (kt + 2 == nk) bias_request(ci + 1);
            ktile(P0{}, T, ci > 0 && kt == 0);
            ktile(P1{}, T + 1, false);
            T += 2;
        }
        if (wr == 0) PK_BAR();
        PK_SEG(3);
        epilogue(ci);
        PK_SEG(4);
    }
#ifdef VH_P256_PROBE
    if ((tid & 63) == 0 && bid < 256) {
        unsigned long long* q = vh_p256_probe + (bid * 8 + ws) * 8;
        for (int i = 0; i < 6; ++i) q[i] = pk_sum[i];
        q[6] = pk_wall1 - pk_wall0;
        q[7] = wall_clock64() - pk_wall0;
    }
#endif
}

bool vh_gemm16_p256_ok(const Gemm16Args& a, int out_kind) {
    if (a.N % 256 != 0 || a.K % 128 != 0 || a.K < 256 || a.M < 1) return false;
    if ((int64_t)256 * a.lda * 2 >= (1ll << 31) || (int64_t)256 * a.K * 2 >= (1ll << 31)) return false;
    if (out_kind == G16_QKV && (a.d_model % 256 != 0 || a.M >= (1 << 24))) return false;
    return true;
}

int vh_gemm16_p256_launch(const Gemm16Args& a, int out_kind, hipStream_t stream) {
    VH_REQUIRE(vh_gemm16_p256_ok(a, out_kind), VH_EUNSUPPORTED, "gemm16_p256: M=%d N=%d K=%d not a shape of the 256^2 form", a.M, a.N,
               a.K);
    const int tm = (a.M + 255) / 256, tn = a.N / 256;
    const int ntiles = tm * tn;
    int grid = ntiles < 256 ? ntiles : 256;                  // one persistent workgroup per CU
    if (grid >= 8) grid &= ~7;                               // whole XCD groups (the kernel's tile split wants G % 8 == 0)
    if (out_kind == G16_F32) hipLaunchKernelGGL(gemm16_p256_kernel<G16_F32>, dim3(grid), dim3(512), 0, stream, a, tm, tn);
    else if (out_kind == G16_BF16) hipLaunchKernelGGL(gemm16_p256_kernel<G16_BF16>, dim3(grid), dim3(512), 0, stream, a, tm, tn);
    else hipLaunchKernelGGL(gemm16_p256_kernel<G16_QKV>, dim3(grid), dim3(512), 0, stream, a, tm, tn);
    VH_CHECK_LAUNCH("gemm16_p256");
    return VH_OK;
}
