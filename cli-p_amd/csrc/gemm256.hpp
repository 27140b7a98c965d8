// gemm256.hpp — the large-shape bf16 NT GEMM: 256x256x64 block tile, 8 waves, LDS-DMA prefetch kept
// in flight ACROSS barriers (counted s_waitcnt vmcnt, raw s_barrier), four phases per K-tile and two
// wave groups running half a phase apart so that one group's MFMA cluster covers the other group's
// LDS reads and DMA issue (the "8-phase" structure of cdna_hip_programming.md §5, re-derived here).
//
// Same contract and epilogues as gemm.hpp (C[m][n] = sum_k A[m][k] W[n][k]); requires N % 256 == 0,
// K % 64 == 0, K >= 128.
//
// LDS (128 KiB, one array): 2 K-tile buffers x [A-lo | A-hi | B-lo | B-hi], each half-tile = 128 rows x
// 64 bf16 (128-B rows, 16-B chunks XOR-swizzled with row & 7: on the DMA SOURCE address and on the
// read address; the DMA destination is lane-linear).
//
// Wave (wm, wn), wm in {0,1}, wn in {0..3}, owns output rows {wm*64..+63} of BOTH A halves and columns
// {wn*32..+31} of BOTH B halves, so the four phases of a K-tile touch the half-tiles in the order
//   P0: A-lo x B-lo   P1: A-lo x B-hi   P2: A-hi x B-hi   P3: A-hi x B-lo
// and each phase needs at most ONE half-tile that was not needed before. Phase p of K-tile t issues
// the DMA of half-tile [A-lo, B-lo, B-hi, A-hi][p] of K-tile t+1 into the other buffer: every half-tile
// has >= 3 phases between its issue and its first read (3 half-tiles = 6 DMA instructions per wave in
// flight).
//
// Ordering argument (slots = intervals between consecutive workgroup barriers; group 1 = waves with
// wm == 1 runs ONE slot behind group 0):
//   RAW  a half-tile is read only after (i) every wave has executed the counted vmcnt that retires ITS
//        two DMA instructions of that half-tile, placed at the end of the load slot of the phase
//        BEFORE the first read, and (ii) a barrier that follows the later group's wait. The reader's
//        load slot begins after exactly that barrier.
//   WAR  a buffer is re-filled for K-tile t+1 from slot 8t on; its last reads (K-tile t-1) complete by
//        slot 8t-2 (lgkmcnt(0) at the head of every MFMA slot).
#pragma once
#include "gemm.hpp"

namespace clipmi {

constexpr int G256_HALF = 128 * 128;            // bytes per half-tile (128 rows x 128 B)
constexpr int G256_BUF = 4 * G256_HALF;         // 64 KiB per K-tile buffer
constexpr int G256_LDS = 2 * G256_BUF;          // 128 KiB

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// WD = true (development build, DESIGN 4.4h "W fragments straight from L2"): the W operand never enters LDS. Every wave
// loads its own B fragments (16 rows x 64 B per instruction, the fragment shape of v_mfma_f32_16x16x32_bf16) with
// global_load_dwordx4 straight into the registers the MFMAs read, two phases ahead of their first use and INTO the
// registers of the fragments they replace - no second register set: the phase order becomes
//   P0: A-lo x B-lo   P1: A-hi x B-lo   P2: A-hi x B-hi   P3: A-lo x B-hi   (A-lo is read from LDS twice)
// so that B-lo is dead after P1 (reloaded in P2 for the next K-tile's P0) and B-hi after P3 (reloaded in P0 for P2).
// Only the A half-tiles go through LDS-DMA: half the LDS-DMA bytes per K-tile. Every accumulator still receives its
// K-steps in the same order: bit-identical results. vmcnt is counted by hand over BOTH kinds (the loads are inline asm:
// beside LDS-DMA in flight the compiler would wait vmcnt(0) for an ordinary load); per K-tile and wave, in issue order:
//   P0: LD B-hi(t) x4 | P1: DMA A-lo(t+1) x2 | P2: LD B-lo(t+1) x4, DMA A-hi(t+1) x2
//   waits: P0 vmcnt(4) retires A-hi(t) [read in P1]; P2 vmcnt(8) retires B-hi(t) [used in P2]; P3 vmcnt(2) retires
//   B-lo(t+1), A-lo(t+1) [used / read in P0 of t+1].
// WAR on LDS: A-lo(t+1) lands in the buffer whose A-lo region group 1 finishes reading (P3 of t-1) one slot into P0(t);
// its DMA is issued in P1(t), two slots later - the margin of the original schedule.
// (A second register set for the fragments - the whole next K-tile requested four to six phases ahead - was measured too:
//  256 VGPRs with 13 spills, slower still: 538-572 TF against 682-743 for this form and 1000-1260 for the LDS-DMA form.)
template <int EPI, int WD = 0>
__global__ void __launch_bounds__(512, 2) gemm256_bf16_nt_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;

    const int ntn = g.N >> 8;
    int bm, bn;
    gemm_tile_coords(blockIdx.x, gridDim.x, (g.M + 255) >> 8, ntn, 8, 4, bm, bn);
    const int m0 = bm << 8, n0 = bn << 8;
    const int K = g.K;

    // ---- DMA source pointers: half-tile rows [16*wave, 16*wave+16), two 8-row pieces per wave
    const int srow = lane >> 3, spos = lane & 7;
    const unsigned short* src[4][2];            // [A-lo, A-hi, B-lo, B-hi][piece]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 16 + i * 8 + srow;
        const int chunk = (spos ^ (row & 7)) * 8;
        int ma = m0 + row, mb = m0 + 128 + row;
        ma = ma < g.M ? ma : g.M - 1;
        mb = mb < g.M ? mb : g.M - 1;
        src[0][i] = g.A + (size_t)ma * gemm_lda(g) + chunk;
        src[1][i] = g.A + (size_t)mb * gemm_lda(g) + chunk;
        src[2][i] = g.W + (size_t)(n0 + row) * K + chunk;
        src[3][i] = g.W + (size_t)(n0 + 128 + row) * K + chunk;
    }
    const int dma_off = wave * 16 * 128;
    // issue half-tile H (0..3 as in `src`) of K-tile kt into buffer `buf`
#define G256_ISSUE(H, kt, buf)                                                                                        \
    do {                                                                                                              \
        char* d_ = smem + (buf) * G256_BUF + (H) * G256_HALF + dma_off;                                               \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[H][0] + (kt) * 64),      \
                                         (__attribute__((address_space(3))) void*)(d_), 16, 0, 0);                    \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[H][1] + (kt) * 64),      \
                                         (__attribute__((address_space(3))) void*)(d_ + 1024), 16, 0, 0);             \
    } while (0)

    // ---- fragment read offsets
    const int sw = fr & 7;
    const int c0 = ((0 + fg) ^ sw) * 16, c1 = ((4 + fg) ^ sw) * 16;
    const int offA = (wm * 64 + fr) * 128;                       // + half*16384 + mt*2048
    const int offB = 2 * G256_HALF + (wn * 32 + fr) * 128;       // + half*16384 + nt*2048

    f32x4 acc[2][4][2][2];     // [A half][mt][B half][nt]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[a][i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    typedef int b128v __attribute__((ext_vector_type(4)));      // 16 bytes of fragment as a plain register quad (asm "=v")
    bf16x8 af[4][2];           // current A half: [mt][ks]
    b128v bl[2][2], bh[2][2];  // B-lo / B-hi: [nt][ks]

#define G256_READ_A(base, half)                                                                      \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                               \
        af[t_][0] = *reinterpret_cast<const bf16x8*>((base) + (half) * G256_HALF + offA + t_ * 2048 + c0); \
        af[t_][1] = *reinterpret_cast<const bf16x8*>((base) + (half) * G256_HALF + offA + t_ * 2048 + c1); \
    }
#define G256_READ_B(dst, base, half)                                                                 \
    _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) {                                               \
        dst[t_][0] = *reinterpret_cast<const b128v*>((base) + (half) * G256_HALF + offB + t_ * 2048 + c0); \
        dst[t_][1] = *reinterpret_cast<const b128v*>((base) + (half) * G256_HALF + offB + t_ * 2048 + c1); \
    }
    // MFMA slot: 16 MFMAs of quadrant (A half a, B half b); D = Wfrag x Afrag (C^T tile, see gemm.hpp)
#define G256_MFMA(a, bfr, b)                                                                         \
    do {                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        __builtin_amdgcn_s_setprio(1);                                                               \
        _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)                                          \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                         \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                     \
                    acc[a][i_][b][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[j_][ks_]), af[i_][ks_], acc[a][i_][b][j_], 0, 0, 0); \
        __builtin_amdgcn_s_setprio(0);                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        __builtin_amdgcn_s_barrier();                                                                \
    } while (0)

    const int nk = K >> 6;
    if constexpr (WD) {
        // per-lane byte offsets of the wave's four B fragment rows [half][nt] (+ fg * 16 B: k-chunk fg; ks = 1 is +64 B)
        unsigned wvo[2][2];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) wvo[h][nt] = (unsigned)((h * 128 + wn * 32 + nt * 16 + fr) * K + fg * 8) * 2u;
        const unsigned short* wb = g.W + (size_t)n0 * K;          // wave-uniform base, + 64 elements per K-tile
#define WD_LD(dst, vo, base, OFF) asm volatile("global_load_dwordx4 %0, %1, %2 offset:" #OFF : "=v"(dst) : "v"(vo), "s"(base) : "memory")
#define WD_LOAD_B(dst, half, kt)                                                                     \
    do {                                                                                             \
        const unsigned short* b_ = wb + (size_t)(kt) * 64;                                           \
        WD_LD(dst[0][0], wvo[half][0], b_, 0);                                                       \
        WD_LD(dst[0][1], wvo[half][0], b_, 64);                                                      \
        WD_LD(dst[1][0], wvo[half][1], b_, 0);                                                       \
        WD_LD(dst[1][1], wvo[half][1], b_, 64);                                                      \
    } while (0)
        // prologue: A-lo(0), B-lo(0), A-hi(0) in the K-loop's own order
        G256_ISSUE(0, 0, 0);
        WD_LOAD_B(bl, 0, 0);
        G256_ISSUE(1, 0, 0);
        wait_vmcnt<2>();
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();
        for (int t = 0; t < nk; ++t) {
            const char* cur = smem + (t & 1) * G256_BUF;
            const int nb = (t + 1) & 1;
            const bool more = t + 1 < nk;             // wave-uniform
            // P0: A-lo x B-lo
            G256_READ_A(cur, 0);
            WD_LOAD_B(bh, 1, t);
            wait_vmcnt<4>();                          // retires A-hi(t)
            __builtin_amdgcn_s_barrier();
            G256_MFMA(0, bl, 0);
            // P1: A-hi x B-lo
            G256_READ_A(cur, 1);
            if (more) G256_ISSUE(0, t + 1, nb);
            __builtin_amdgcn_s_barrier();
            G256_MFMA(1, bl, 0);
            // P2: A-hi x B-hi (A-hi fragments still in registers); B-lo's registers take the next K-tile's fragments
            if (more) {
                WD_LOAD_B(bl, 0, t + 1);
                G256_ISSUE(1, t + 1, nb);
                wait_vmcnt<8>();                      // retires B-hi(t)
            } else {
                wait_vmcnt<0>();
            }
            __builtin_amdgcn_s_barrier();
            G256_MFMA(1, bh, 1);
            // P3: A-lo x B-hi (A-lo read from LDS a second time)
            G256_READ_A(cur, 0);
            if (more) wait_vmcnt<2>();                // retires B-lo(t+1), A-lo(t+1)
            __builtin_amdgcn_s_barrier();
            G256_MFMA(0, bh, 1);
        }
#undef WD_LOAD_B
#undef WD_LD
    } else {
    // ---- prologue: K-tile 0 into buffer 0, in the order of first use
    G256_ISSUE(0, 0, 0);
    G256_ISSUE(2, 0, 0);
    G256_ISSUE(3, 0, 0);
    G256_ISSUE(1, 0, 0);
    wait_vmcnt<4>();                         // A-lo(0), B-lo(0) landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();     // group 1 runs one slot behind

    for (int t = 0; t < nk - 1; ++t) {
        const char* cur = smem + (t & 1) * G256_BUF;
        const int nb = (t + 1) & 1;
        // P0: A-lo x B-lo
        G256_READ_B(bl, cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        G256_READ_A(cur, 0);
        G256_ISSUE(0, t + 1, nb);
        wait_vmcnt<4>();                     // retires B-hi(t)
        __builtin_amdgcn_s_barrier();
        G256_MFMA(0, bl, 0);
        // P1: A-lo x B-hi
        G256_READ_B(bh, cur, 1);
        G256_ISSUE(2, t + 1, nb);
        wait_vmcnt<4>();                     // retires A-hi(t)
        __builtin_amdgcn_s_barrier();
        G256_MFMA(0, bh, 1);
        // P2: A-hi x B-hi
        G256_READ_A(cur, 1);
        G256_ISSUE(3, t + 1, nb);
        __builtin_amdgcn_s_barrier();
        G256_MFMA(1, bh, 1);
        // P3: A-hi x B-lo (B-lo fragments still in registers)
        G256_ISSUE(1, t + 1, nb);
        wait_vmcnt<4>();                     // retires A-lo(t+1), B-lo(t+1)
        __builtin_amdgcn_s_barrier();
        G256_MFMA(1, bl, 0);
    }
    {   // last K-tile: nothing left to prefetch, the counts shrink
        const char* cur = smem + ((nk - 1) & 1) * G256_BUF;
        G256_READ_B(bl, cur, 0);
        __builtin_amdgcn_sched_barrier(0);
        G256_READ_A(cur, 0);
        wait_vmcnt<2>();                     // retires B-hi(last)
        __builtin_amdgcn_s_barrier();
        G256_MFMA(0, bl, 0);
        G256_READ_B(bh, cur, 1);
        wait_vmcnt<0>();                     // retires A-hi(last)
        __builtin_amdgcn_s_barrier();
        G256_MFMA(0, bh, 1);
        G256_READ_A(cur, 1);
        __builtin_amdgcn_s_barrier();
        G256_MFMA(1, bh, 1);
        __builtin_amdgcn_s_barrier();
        G256_MFMA(1, bl, 0);
    }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();     // balance the stagger: every wave has the same barrier count
#undef G256_ISSUE
#undef G256_READ_A
#undef G256_READ_B
#undef G256_MFMA

    // ---- epilogue, staged through LDS (the K-loop's buffers are dead after the last barrier above).
    // Fragment-shaped stores (16 rows x 32 B per wave-instruction, 32 instructions per lane) made the
    // tail store-ISSUE-bound: ~12 us per round of tiles, 29 us for the f32 read-modify-write (r01,
    // tools/gemm_overhead.py). Instead every lane drops its values (bias / QuickGELU applied) into a
    // row-major LDS image of the tile — 16-byte chunks XOR-swizzled with row & 15 so that the 16 rows
    // of a fragment column do not share banks — and the tile leaves as whole rows, 16 B per lane,
    // 512 B..1 KiB contiguous per wave-instruction; the residual / positional add happens on that pass.
    f32x4 bz[2][2], cs[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + b * 128 + wn * 32 + nt * 16 + 4 * fg;
            bz[b][nt] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            if (epi_is_ln(EPI)) cs[b][nt] = *reinterpret_cast<const f32x4*>(g.colsum + n);
        }
    __syncthreads();
    if (epi_is_bf16_out(EPI)) {
        // image: 256 rows x 512 B
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) {
            const int row = (idx >> 2) * 128 + wm * 64 + (idx & 3) * 16 + fr;
            f32x2 st = {0.f, 1.f};
            if (epi_is_ln(EPI)) {
                const int m = m0 + row;
                const int nseg = g.K >> 8;
                st = ln_row_stats(g.ln_part_in + (size_t)(m < g.M ? m : g.M - 1) * 2 * nseg, nseg, g.K);
            }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    f32x4 v;
                    if (epi_is_ln(EPI)) v = ln_apply(acc[idx >> 2][idx & 3][b][nt], st.x, st.y, cs[b][nt], bz[b][nt]);
                    else v = acc[idx >> 2][idx & 3][b][nt] + bz[b][nt];
                    if (epi_is_qgelu(EPI)) {
                        v = quick_gelu4(v);
                    }
                    const int colbyte = (b * 128 + wn * 32 + nt * 16 + 4 * fg) * 2;
                    const int off = row * 512 + ((((colbyte >> 4) ^ (row & 15)) << 4) | (colbyte & 8));
                    *reinterpret_cast<uint2*>(smem + off) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
                }
        }
        __syncthreads();
        unsigned short* outp = static_cast<unsigned short*>(g.out);
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int row = wave * 32 + i * 2 + (lane >> 5);
            const int chunk = lane & 31;
            const uint4 v = *reinterpret_cast<const uint4*>(smem + row * 512 + ((chunk ^ (row & 15)) << 4));
            const int m = m0 + row;
            if (m < g.M) *reinterpret_cast<uint4*>(outp + (size_t)m * g.N + n0 + chunk * 8) = v;
        }
    } else {
        // two passes of 128 rows x 1 KiB (f32)
        float* outp = static_cast<float*>(g.out);
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (a) __syncthreads();
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int row = wm * 64 + mt * 16 + fr;
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const f32x4 v = acc[a][mt][b][nt] + bz[b][nt];
                        const int chunk = (b * 128 + wn * 32 + nt * 16 + 4 * fg) >> 2;
                        *reinterpret_cast<f32x4*>(smem + row * 1024 + ((chunk ^ (row & 15)) << 4)) = v;
                    }
            }
            __syncthreads();
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int row = wave * 16 + i;
                f32x4 v = *reinterpret_cast<const f32x4*>(smem + row * 1024 + ((lane ^ (row & 15)) << 4));
                const int m = m0 + a * 128 + row;
                if (m < g.M) {
                    size_t orow = (size_t)m;
                    if (EPI == EPI_PATCH_F32) {
                        const int b_ = m / g.np, p_ = m - b_ * g.np;
                        orow = (size_t)b_ * g.L + 1 + p_;
                        v += *reinterpret_cast<const f32x4*>(g.pos + (size_t)(1 + p_) * g.N + n0 + lane * 4);
                    }
                    float* dst = outp + orow * g.N + n0 + lane * 4;
                    if (EPI == EPI_BIAS_RESID_F32) v += *reinterpret_cast<const f32x4*>(dst);
                    *reinterpret_cast<f32x4*>(dst) = v;
                }
            }
        }
    }
}

}  // namespace clipmi
