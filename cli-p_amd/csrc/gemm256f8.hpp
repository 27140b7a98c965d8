// gemm256f8.hpp — the 256x256 pipelined NT GEMM of gemm256.hpp on OCP FP8 (e4m3) operands and the FP8 matrix
// cores (BASELINE.json configs[4]: "fp8 ViT-B/32 weights on CDNA4 fp8 MFMA"):
//     C[m][n] = a_scale[m] * w_scale[n] * sum_k A8[m][k] W8[n][k]  (+ bias, epilogues as gemm256)
// A8 = activations quantised per ROW (clipmi quantize_rows_fp8_kernel: scale = max|row| / 448, RNE), W8 = weights
// quantised per OUTPUT CHANNEL at pack time. Products of two e4m3 values are exact in f32 and the MFMA accumulates
// in f32, so the result equals an f32 matmul of the dequantised operands up to summation order.
//
// Same LDS geometry in BYTES as gemm256 (128-byte rows, 16-byte chunks XOR-swizzled with row & 7, LDS-DMA, counted
// vmcnt, staggered wave groups): a K-tile is 128 fp8 values instead of 64 bf16. Each 16-byte fragment a lane reads
// holds 16 consecutive k; v_mfma_f32_16x16x32_fp8_fp8 takes 8 per lane, so every fragment pair feeds TWO MFMAs (low
// halves, high halves): the k subsets of the two differ but agree between A and W, which is all a dot product
// needs. The non-scaled FP8 MFMA runs at the bf16 rate (MI355X_MICROARCH.md, matrix-core table): this path halves
// operand bytes, not MFMA cycles; the 2x rate needs the block-scaled (MX) forms, not built.
// Requires N % 256 == 0, K % 128 == 0, K >= 256.
#pragma once
#include "gemm256.hpp"

namespace clipmi {

typedef long i64x2v __attribute__((ext_vector_type(2)));
typedef int i32x8v __attribute__((ext_vector_type(8)));

// BSA ("block-scaled A", round 3): the activations carry one e8m0 scale per 32 consecutive k (the MX format; written by
// the PRODUCING kernel's epilogue or by quantize_rows_fp8mx_kernel - a row scale would need the whole row before the first
// byte can be written, a 32-block is local to any column tile) in g.a_bscale [rows padded to 256][K / 32]. The scaled MFMA
// takes it as its per-lane scale operand, so the block scales cost no
// matrix cycles; the tile's 256 x K/32 scale bytes arrive once by LDS-DMA behind the K-tile buffers. a_scale is unused.
// Which k a lane's scale reaches was measured (tools/bsa_probe.py): the instruction orders its 128 k as [first 16 bytes of
// lane groups 0..3 | second 16 bytes of groups 0..3], cuts THAT into four blocks of 32 and takes block s's scale from lane
// group s - so with this kernel's fragment layout (a lane holds the 16-byte chunks fg and 4 + fg of the 128-byte row) the
// hardware's block s is k = 32 s .. 32 s + 31 of the K-tile, and lane (fr, fg) supplies the scale of (row fr, block fg).
template <int EPI, bool MX, bool BSA = false>
__global__ void __launch_bounds__(512, 2) gemm256f8_nt_kernel(GemmArgs g) {
    static_assert(!BSA || MX, "block scales need the scaled MFMA form");
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
    const unsigned char* A8 = reinterpret_cast<const unsigned char*>(g.A);
    const unsigned char* W8 = reinterpret_cast<const unsigned char*>(g.W);
    const unsigned char* src[4][2];             // [A-lo, A-hi, B-lo, B-hi][piece]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = wave * 16 + i * 8 + srow;
        const int chunk = (spos ^ (row & 7)) * 16;          // bytes
        int ma = m0 + row, mb = m0 + 128 + row;
        ma = ma < g.M ? ma : g.M - 1;
        mb = mb < g.M ? mb : g.M - 1;
        src[0][i] = A8 + (size_t)ma * K + chunk;
        src[1][i] = A8 + (size_t)mb * K + chunk;
        src[2][i] = W8 + (size_t)(n0 + row) * K + chunk;
        src[3][i] = W8 + (size_t)(n0 + 128 + row) * K + chunk;
    }
    const int dma_off = wave * 16 * 128;
    // issue half-tile H (0..3 as in `src`) of K-tile kt into buffer `buf`
#define G256_ISSUE(H, kt, buf)                                                                                        \
    do {                                                                                                              \
        char* d_ = smem + (buf) * G256_BUF + (H) * G256_HALF + dma_off;                                               \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[H][0] + (kt) * 128),      \
                                         (__attribute__((address_space(3))) void*)(d_), 16, 0, 0);                    \
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src[H][1] + (kt) * 128),      \
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

    // BSA: the tile's scale bytes [256][K / 32] behind the K-tile buffers (rows past M: the array is padded to 256 rows)
    const int KB32 = K >> 5;
    const unsigned char* slab = reinterpret_cast<const unsigned char*>(smem) + G256_LDS;
    if (BSA) {
        const unsigned char* sg = g.a_bscale + (size_t)m0 * KB32 + lane * 16;
        for (int pc = wave; pc < (KB32 >> 2); pc += 8)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(sg + pc * 1024),
                                             (__attribute__((address_space(3))) void*)(smem + G256_LDS + pc * 1024), 16, 0, 0);
    }
    int sb[4] = {0x7f, 0x7f, 0x7f, 0x7f};      // e8m0 scale of this lane's (row, k-block) per mt of the current A half
    int kt_cur = 0;
    const int sb_off = (wm * 64 + fr) * KB32 + fg;

    i64x2v af[4][2];           // current A half: [mt][16-byte fragment]
    i64x2v bl[2][2], bh[2][2]; // B-lo / B-hi: [nt][16-byte fragment]

#define G256_READ_A(base, half)                                                                      \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                               \
        af[t_][0] = *reinterpret_cast<const i64x2v*>((base) + (half) * G256_HALF + offA + t_ * 2048 + c0); \
        af[t_][1] = *reinterpret_cast<const i64x2v*>((base) + (half) * G256_HALF + offA + t_ * 2048 + c1); \
        if (BSA) sb[t_] = slab[sb_off + ((half) * 128 + t_ * 16) * KB32 + kt_cur * 4];               \
    }
#define G256_READ_B(dst, base, half)                                                                 \
    _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) {                                               \
        dst[t_][0] = *reinterpret_cast<const i64x2v*>((base) + (half) * G256_HALF + offB + t_ * 2048 + c0); \
        dst[t_][1] = *reinterpret_cast<const i64x2v*>((base) + (half) * G256_HALF + offB + t_ * 2048 + c1); \
    }
    // MFMA slot of quadrant (A half a, B half b); D = Wfrag x Afrag (C^T tile, see gemm.hpp).
    //   MX = false: 32 x v_mfma_f32_16x16x32_fp8_fp8 (low and high 8 bytes of each fragment pair; f32 accumulation of
    //               exact products; the bf16 MFMA rate)
    //   MX = true:  8 x v_mfma_scale_f32_16x16x128_f8f6f4 with UNIT block scales (e8m0 127 = 2^0): one instruction takes
    //               both 16-byte fragments of a lane (32 of the 128 k), twice the k per cycle. Probed on hardware
    //               (tools/probe/mfma_mx_probe.hip): the same products, summed with ~2^-15 relative error per
    //               instruction (a narrower adder than the f32 chain), far below e4m3's own 2^-4.
#define G256_MFMA(a, bfr, b)                                                                         \
    do {                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        __builtin_amdgcn_s_setprio(1);                                                               \
        if constexpr (MX) {                                                                          \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                         \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                   \
                    const i64x2v b0_ = bfr[j_][0], b1_ = bfr[j_][1], a0_ = af[i_][0], a1_ = af[i_][1]; \
                    typedef long i64x4v_ __attribute__((ext_vector_type(4)));                        \
                    const i64x4v_ bw_ = {b0_.x, b0_.y, b1_.x, b1_.y}, aw_ = {a0_.x, a0_.y, a1_.x, a1_.y}; \
                    acc[a][i_][b][j_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(            \
                        __builtin_bit_cast(i32x8v, bw_), __builtin_bit_cast(i32x8v, aw_), acc[a][i_][b][j_], 0, 0, 0, \
                        0x7f7f7f7f, 0, BSA ? sb[i_] : 0x7f7f7f7f);                                   \
                }                                                                                    \
        } else {                                                                                     \
            _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)                                      \
                _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                     \
                    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                               \
                        acc[a][i_][b][j_] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bfr[j_][ks_].x, af[i_][ks_].x, acc[a][i_][b][j_], 0, 0, 0); \
                        acc[a][i_][b][j_] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bfr[j_][ks_].y, af[i_][ks_].y, acc[a][i_][b][j_], 0, 0, 0); \
                    }                                                                                \
        }                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        __builtin_amdgcn_s_barrier();                                                                \
    } while (0)

    const int nk = K >> 7;
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
        kt_cur = t;
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
        kt_cur = nk - 1;
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
    f32x4 bz[2][2], ws[2][2], cs[2][2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = n0 + b * 128 + wn * 32 + nt * 16 + 4 * fg;
            bz[b][nt] = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
            ws[b][nt] = *reinterpret_cast<const f32x4*>(g.w_scale + n);
            if (epi_is_ln(EPI)) cs[b][nt] = *reinterpret_cast<const f32x4*>(g.colsum + n);
        }
    float as[2][4];            // activation row scales of this lane's accumulator rows: [A half][mt]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            int m = m0 + a * 128 + wm * 64 + mt * 16 + fr;
            m = m < g.M ? m : g.M - 1;
            as[a][mt] = BSA ? 1.0f : g.a_scale[m];
        }
    __syncthreads();
    if (epi_is_bf16_out(EPI)) {
        // image: 256 rows x 512 B
#pragma unroll
        for (int idx = 0; idx < 8; ++idx) {
            const int row = (idx >> 2) * 128 + wm * 64 + (idx & 3) * 16 + fr;
            // LN-folded consumer (round 4; gemm.hpp): A = e4m3(x) with MX block scales, W = e4m3(W diag(gamma)): the epilogue is
            // rstd (acc w_scale - mean colsum) + cb with colsum of the ROUNDED e4m3 weights (weights.py ln_fold_terms_fp8)
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
                    if (epi_is_ln(EPI)) v = ln_apply(acc[idx >> 2][idx & 3][b][nt] * (ws[b][nt] * as[idx >> 2][idx & 3]), st.x, st.y, cs[b][nt], bz[b][nt]);
                    else v = acc[idx >> 2][idx & 3][b][nt] * (ws[b][nt] * as[idx >> 2][idx & 3]) + bz[b][nt];
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
        unsigned char* const out8 = static_cast<unsigned char*>(g.out);
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            const int row = wave * 32 + i * 2 + (lane >> 5);
            const int chunk = lane & 31;
            const uint4 v = *reinterpret_cast<const uint4*>(smem + row * 512 + ((chunk ^ (row & 15)) << 4));
            const int m = m0 + row;
            if (epi_is_qgelu(EPI) && g.out_bscale) {
                // the rows leave as e4m3 + MX block scales (the next GEMM's A operand) instead of bf16: gemm.hpp mx_pack_bf16x8,
                // the bytes the persistent kernel's storers and quantize_rows_fp8mx_kernel write
                unsigned sb_;
                const uint2 q8_ = mx_pack_bf16x8(v, sb_);
                if (m < g.M) {
                    *reinterpret_cast<uint2*>(out8 + (size_t)m * g.N + n0 + chunk * 8) = q8_;
                    if ((lane & 3) == 0) g.out_bscale[(size_t)m * (g.N >> 5) + ((n0 + chunk * 8) >> 5)] = (unsigned char)sb_;
                }
            } else if (m < g.M) {
                *reinterpret_cast<uint4*>(outp + (size_t)m * g.N + n0 + chunk * 8) = v;
            }
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
                        const f32x4 v = acc[a][mt][b][nt] * (ws[b][nt] * as[a][mt]) + bz[b][nt];
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
                    if (EPI == EPI_BIAS_RESID_F32 || EPI == EPI_BIAS_RESID_LN8) v += *reinterpret_cast<const f32x4*>(dst);
                    *reinterpret_cast<f32x4*>(dst) = v;
                }
                if constexpr (EPI == EPI_BIAS_RESID_LN8) {
                    // the new residual row also leaves as e4m3 + MX block scales (A operand of the next LN-folded FP8 GEMM) with
                    // its 256-column statistics partial; a wave holds the whole segment (4 columns per lane): gemm.hpp
                    // ln8_row_segment, shared with rows_mx_stats_kernel. Rows past M are computed (wave-wide shuffles) and dropped.
                    unsigned p4_, sb_;
                    float sm_, sq_;
                    ln8_row_segment(v, p4_, sb_, sm_, sq_);
                    if (m < g.M) {
                        *reinterpret_cast<unsigned*>(g.x8 + (size_t)m * g.N + n0 + lane * 4) = p4_;
                        if ((lane & 7) == 0) g.x8_bs[(size_t)m * (g.N >> 5) + ((n0 + lane * 4) >> 5)] = (unsigned char)sb_;
                        if (lane == 0) *reinterpret_cast<f32x2*>(g.ln_part + ((size_t)m * (g.N >> 8) + (n0 >> 8)) * 2) = f32x2{sm_, sq_};
                    }
                }
            }
        }
    }
}

}  // namespace clipmi
