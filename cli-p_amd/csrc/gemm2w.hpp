// gemm2w.hpp — the residual producer (EPI_BIAS_RESID_LN_F32) as TWO co-resident workgroups per CU (round 3;
// "overlap a CU's epilogue with its own K-loop").
//
// gemm256p's residual epilogue is ~20 us per 256 x 256 tile in which no wave of the CU issues an MFMA (hi / lo
// read-modify-write + statistics: 512 KB through the CU's memory pipe), behind an 18 us (out_proj, K = 768) or 70 us
// (c_proj, K = 3072) K-loop. The accumulators of a 256 x 256 tile fill the register file (128 of 256 registers per
// wave), so the same workgroup cannot start the next tile's K-loop under its own store pass. Here the CU holds two
// workgroups of 4 waves with a 128 x 256 tile each (wave tile 64 x 128: the same 128 accumulator registers), started
// half a K-loop apart: while one is in its store pass the other has the matrix pipe to itself.
//
// What that costs: 72 KiB of LDS per workgroup = a ring of THREE 24-KiB K-tiles of 32 (A 128 rows + W 256 rows of 64
// bytes; 64-deep tiles would need 96 KiB per workgroup), one barrier and 32 MFMAs per wave and K-tile, K-tiles t + 1 and
// t + 2 in flight under tile t's MFMAs.
//
// Arithmetic: per output element the MFMA chain is the other kernels' (D = Wfrag x Afrag, K-steps of 32 in increasing k
// into one accumulator), the store pass computes (acc + bias) + (hi + lo) and the canonical statistics with the shared
// device functions in gemm256p's order: identical bits (test_gemm_resid_ln_producer, algo 4).
#pragma once
#include "gemm256.hpp"

namespace clipmi {

constexpr int G2W_STAGE = (128 + 256) * 64;      // one K-tile of 32: A 8 KiB | W 16 KiB
constexpr int G2W_LDS = 3 * G2W_STAGE;           // 72 KiB: two workgroups per CU

// 64-byte LDS rows: the 16-byte chunk c of row r sits at position c ^ swz(r), swz = a permutation of ((r >> 2) & 3)
// (0, 1, 2, 3 -> 0, 2, 3, 1) chosen so that each of ds_read_b128's four lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19,
// 28-31}, ... MI355X_MICROARCH.md LDS table) meets 16 different 16-byte slots of the 256-byte bank row when lane
// (fr, fg) reads chunk fg of row R + fr: conflict-free fragment reads (the plain (r >> 2) & 3 is 2-way).
__device__ __forceinline__ int g2w_swz(int r) {
    const int q = (r >> 2) & 3;
    return ((((q >> 1) ^ q) & 1) << 1) | (q >> 1);
}

__global__ void __launch_bounds__(256, 2) gemm2w_resid_ln_kernel(GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;      // 2 x 2 waves of 64 (m) x 128 (n)
    const int fr = lane & 15, fg = lane >> 4;

    const int ntn = g.N >> 8, mtiles = (g.M + 127) >> 7;
    int bm, bn;
    gemm_tile_coords(blockIdx.x, gridDim.x, mtiles, ntn, 8, 4, bm, bn);
    const int m0 = bm << 7, n0 = bn << 8;
    const int K = g.K, nk = K >> 5;

    // the second workgroup of every CU starts half a K-loop late (dbg = n > 0: n - 1 units of ~3.7 us, development A/B)
    if (blockIdx.x >= NUM_CU && blockIdx.x < 2 * NUM_CU) {
        const int units = g.dbg > 0 ? g.dbg - 1 : (nk + 15) / 16;
        for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(127);
    }

    // ---- LDS-DMA: a piece = 16 rows x 64 B (lane l: row l >> 2, 16-byte chunk l & 3, XOR-swizzled with g2w_swz(row) on the
    // SOURCE address; the destination is lane-linear). A stage is 8 A pieces + 16 W pieces; wave w issues pieces 6 w .. 6 w + 5.
    const int prow = lane >> 2, pch = lane & 3;
    const unsigned short* src[6];
    int dst[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int p = wave * 6 + i;
        const bool isw = p >= 8;
        const int r = (isw ? p - 8 : p) * 16 + prow;
        const int chunk = pch ^ g2w_swz(r);
        if (isw) {
            src[i] = g.W + (size_t)(n0 + r) * K + chunk * 8;
        } else {
            int m = m0 + r;
            m = m < g.M ? m : g.M - 1;              // M tail: duplicate the last row, never stored
            src[i] = g.A + (size_t)m * K + chunk * 8;
        }
        dst[i] = p * 1024;
    }
    auto issue = [&](int kt, char* stage) {
#pragma unroll
        for (int i = 0; i < 6; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[i] + kt * 32),
                                             (__attribute__((address_space(3))) void*)(stage + dst[i]), 16, 0, 0);
    };

    // ---- fragment read offsets: row (16 t + fr) of the wave's slice, chunk fg ^ swz(row) = fg ^ swz(fr)
    const int sw = g2w_swz(fr);
    const int a_off = (wm * 64 + fr) * 64 + ((fg ^ sw) << 4);              // + mt * 1024
    const int b_off = 8192 + (wn * 128 + fr) * 64 + ((fg ^ sw) << 4);      // + nt * 1024

    f32x4 acc[4][8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    issue(0, smem);
    if (nk > 1) issue(1, smem + G2W_STAGE);
    char* cur = smem;
    char* nxt2 = smem + 2 * G2W_STAGE;                 // where K-tile t + 2 goes
    for (int t = 0; t < nk; ++t) {
        if (t + 1 < nk) wait_vmcnt<6>();               // this wave's pieces of K-tile t have landed (t + 1's may be in flight)
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();                  // ... everyone's have; K-tile t - 1 is fully consumed
        if (t + 2 < nk) issue(t + 2, nxt2);
        bf16x8 af[4], wf[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = *reinterpret_cast<const bf16x8*>(cur + a_off + i * 1024);
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[j] = *reinterpret_cast<const bf16x8*>(cur + b_off + j * 1024);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[j], af[i], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        nxt2 = cur;                                    // K-tile t + 3 goes where K-tile t was
        cur = cur + G2W_STAGE == smem + G2W_LDS ? smem : cur + G2W_STAGE;
    }

    // ---- store pass: two sub-passes of 64 rows. The wave group wm == h drops acc + bias into a 64-row x 1-KiB f32 image
    // (16-byte chunks XOR-swizzled with row & 15), then every wave takes 16 whole rows of it in two groups of eight:
    // old = hi + lo from global memory (requested before the image is read), o = (acc + bias) + old, hi' = bf16(o),
    // lo' = bf16(o - hi'), the rows' (sum, sum of squares) over this tile's 256 columns in the canonical order.
    f32x4 bz[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
        bz[j] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n0 + wn * 128 + j * 16 + 4 * fg) : f32x4{0.f, 0.f, 0.f, 0.f};
    const int nseg = g.N >> 8, seg = n0 >> 8;
    __syncthreads();                                   // the K-loop's last reads are done: the ring is free
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (wm == h) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = i * 16 + fr;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int chunk = wn * 32 + j * 4 + fg;
                    *reinterpret_cast<f32x4*>(smem + row * 1024 + ((chunk ^ (row & 15)) << 4)) = acc[i][j] + bz[j];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int grp = 0; grp < 2; ++grp) {
            const int lr0 = h * 64 + wave * 16 + grp * 8;          // first of this group's 8 tile rows
            uint2 hs[8], ls[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                int m = m0 + lr0 + i;
                m = m < g.M ? m : g.M - 1;
                const size_t e = (size_t)m * g.N + n0 + lane * 4;
                hs[i] = *reinterpret_cast<const uint2*>(g.xhi + e);
                ls[i] = *reinterpret_cast<const uint2*>(g.xlo + e);
            }
            float sa[8], sq[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = lr0 - h * 64 + i;                 // image row
                const f32x4 x = *reinterpret_cast<const f32x4*>(smem + row * 1024 + ((lane ^ (row & 15)) << 4));
                const f32x4 o = x + split_join(hs[i], ls[i]);
                uint2 nh, nl;
                split_make(o, nh, nl);
                const int m = m0 + lr0 + i;
                if (m < g.M) {
                    const size_t e = (size_t)m * g.N + n0 + lane * 4;
                    *reinterpret_cast<uint2*>(g.xhi + e) = nh;
                    *reinterpret_cast<uint2*>(g.xlo + e) = nl;
                }
                sa[i] = ln_lane_sum(o);
                sq[i] = ln_lane_sumsq(o);
            }
            // 8 rows x (sum, sum of squares) reduced TRANSPOSED over xor 32, 16, 8, then over 4, 2, 1: the same adds as
            // ln_wave_sum per row (gemm256p's store pass)
            const bool h32 = lane & 32, h16 = lane & 16, h8 = lane & 8;
            float a4[4], q4[4], a2[2], q2[2];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a4[j] = (h32 ? sa[j + 4] : sa[j]) + __shfl_xor(h32 ? sa[j] : sa[j + 4], 32);
                q4[j] = (h32 ? sq[j + 4] : sq[j]) + __shfl_xor(h32 ? sq[j] : sq[j + 4], 32);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                a2[j] = (h16 ? a4[j + 2] : a4[j]) + __shfl_xor(h16 ? a4[j] : a4[j + 2], 16);
                q2[j] = (h16 ? q4[j + 2] : q4[j]) + __shfl_xor(h16 ? q4[j] : q4[j + 2], 16);
            }
            float a1 = (h8 ? a2[1] : a2[0]) + __shfl_xor(h8 ? a2[0] : a2[1], 8);
            float q1 = (h8 ? q2[1] : q2[0]) + __shfl_xor(h8 ? q2[0] : q2[1], 8);
#pragma unroll
            for (int o_ = 4; o_ >= 1; o_ >>= 1) {
                a1 += __shfl_xor(a1, o_);
                q1 += __shfl_xor(q1, o_);
            }
            const int mrow = m0 + lr0 + (lane >> 3);              // the row this lane group ended up with
            if ((lane & 7) == 0 && mrow < g.M) *reinterpret_cast<f32x2*>(g.ln_part + ((size_t)mrow * nseg + seg) * 2) = f32x2{a1, q1};
        }
        if (h == 0) __syncthreads();                   // the image is free for the other wave group's rows
    }
}

}  // namespace clipmi
