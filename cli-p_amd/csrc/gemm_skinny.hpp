// gemm_skinny.hpp — the same bf16 "NT" GEMM (gemm.hpp) for M <= 128 rows: ONE prompt through the text tower
// (reference query-index.py:108: model.encode_text(texts) with a single text, M = 77 tokens) and ONE image
// (build-index.py:48-49 is B = 1, M = 50).
//
// Why a third kernel: with M <= 128 the tiled kernels have N / 128 workgroups to run (out_proj / c_proj of the text
// tower: FOUR), each walking K in 64-wide tiles behind a barrier: ~100 dependent launches of ~9 us, 0.88 ms per
// prompt for 76 MB of weights (r02). Here a wave owns 16 output columns x 16 rows and walks K alone: its two 16-byte
// fragments per 32-wide K-step come straight from global memory to registers (the weights are read once per launch and
// shared with nobody, A is a few hundred KB in L2: an LDS round trip would be overhead), several K-steps in flight,
// no barrier anywhere. (N / 16) x ceil(M / 16) waves: 640 for c_fc at M = 77.
//
// Arithmetic: the MFMA sequence per output element is the tiled kernels' (D = Wfrag x Afrag, K-steps of 32 in
// increasing k into one accumulator) and the epilogues are the shared device functions, so a row's result does not
// depend on which kernel computed it.
#pragma once
#include "gemm.hpp"

namespace clipmi {

// Workgroups of ONE wave when that still leaves the 256 CUs short of work (out_proj / c_proj of one prompt: 32 column
// strips) - a CU pulls ~25-60 GB/s through its memory pipe, so a launch's weight bytes must be spread over as many CUs as
// there are strips; four neighbouring strips per workgroup otherwise. DEPTH = K-steps of 32 in flight per wave (2 x 16 B
// per lane each): 16 covers K = 512 in one round trip, 32 (256 registers, one wave per SIMD - there are few waves anyway)
// walks K = 2048 in two.
template <int EPI, int DEPTH>
__global__ void __launch_bounds__(256) gemm_skinny_kernel(GemmArgs g) {
    constexpr int SKINNY_DEPTH = DEPTH;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int nstrips = g.N >> 4;
    const int strip = blockIdx.x * (blockDim.x >> 6) + wave;
    if (strip >= nstrips) return;
    const int n0 = strip << 4, m0 = blockIdx.y << 4;
    const int K = g.K;
    int am = m0 + fr;
    const bool valid = am < g.M;
    am = valid ? am : g.M - 1;                                   // M tail: duplicate the last row, masked at the store
    const bf16x8* wp = reinterpret_cast<const bf16x8*>(g.W + (size_t)(n0 + fr) * K + 8 * fg);
    const bf16x8* ap = reinterpret_cast<const bf16x8*>(g.A + (size_t)am * gemm_lda(g) + 8 * fg);
    const int nk = K >> 5;                                       // K-steps of 32: 4 bf16x8 apart

    // the epilogue's operands are requested first: their round trip then runs under the K-walk's instead of behind it (in a
    // chain of ~90 dependent launches of ~4 us every serial L2 round trip counts)
    const int n = n0 + 4 * fg;
    const f32x4 bz = g.bias ? *reinterpret_cast<const f32x4*>(g.bias + n) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    float part[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    f32x4 add = {0.f, 0.f, 0.f, 0.f};
    size_t orow = (size_t)am;
    if (epi_is_ln(EPI)) {
        cs = *reinterpret_cast<const f32x4*>(g.colsum + n);
        const int nseg = K >> 8;
        if (!g.ln_leaf_in) {
            const float* pp = g.ln_part_in + orow * 2 * nseg;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nseg) { part[2 * i] = pp[2 * i]; part[2 * i + 1] = pp[2 * i + 1]; }
        }
    } else if (EPI == EPI_PATCH_F32) {
        const int b_ = am / g.np, p_ = am - b_ * g.np;
        orow = (size_t)b_ * g.L + 1 + p_;
        add = *reinterpret_cast<const f32x4*>(g.pos + (size_t)(1 + p_) * g.N + n);
    } else if (EPI == EPI_BIAS_RESID_F32) {
        add = *reinterpret_cast<const f32x4*>(static_cast<const float*>(g.out) + orow * g.N + n);
    }
    // EPI_BIAS_RESID_LN_F32 (round 5): the lane's four values of the split residual row, updated in place in the epilogue
    uint2 xh = make_uint2(0u, 0u);
    unsigned xl = 0u;
    if (EPI == EPI_BIAS_RESID_LN_F32) {
        xh = *reinterpret_cast<const uint2*>(resid_hi(g.x3, orow, g.N) + n);
        xl = *reinterpret_cast<const unsigned*>(resid_lo(g.x3, orow, g.N) + n);
    }
    // LN-folded consumer behind such a producer: (mean, rstd) from the statistics LEAVES (GemmArgs.ln_leaf_in) by the canonical
    // tree of ln_wave_sum. A 256-column segment has 64 leaves; lane (fr, fg) takes leaves 4 t + fg, t < 16, of its row: tree
    // levels 32 / 16 / 8 / 4 pair t ^ 8 / 4 / 2 / 1 inside the lane, levels 2 / 1 pair lanes fg ^ 2 / fg ^ 1 (= lane ^ 32 / ^ 16).
    if (epi_is_ln(EPI) && g.ln_leaf_in) {
        const int nseg = K >> 8;
        const f32x2* lf = reinterpret_cast<const f32x2*>(g.ln_leaf_in) + orow * (size_t)(K >> 2) + fg;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < nseg) {
                f32x2 v[16];
#pragma unroll
                for (int t = 0; t < 16; ++t) v[t] = lf[64 * j + 4 * t];
                float sm[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float a[8], b[4];
#pragma unroll
                    for (int t = 0; t < 8; ++t) a[t] = v[t][c] + v[t + 8][c];          // level 32
#pragma unroll
                    for (int t = 0; t < 4; ++t) b[t] = a[t] + a[t + 4];                // level 16
                    const float c0 = b[0] + b[2], c1 = b[1] + b[3];                    // level 8
                    float d = c0 + c1;                                                 // level 4
                    d = lane_fold32(d, d);                                             // level 2
                    sm[c] = lane_fold16(d, d);                                         // level 1
                }
                part[2 * j] = sm[0];
                part[2 * j + 1] = sm[1];
            }
        }
    }

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    bf16x8 wf[SKINNY_DEPTH], af[SKINNY_DEPTH];
#pragma unroll
    for (int d = 0; d < SKINNY_DEPTH; ++d) {
        const int s = d < nk ? d : nk - 1;
        wf[d] = wp[4 * s];
        af[d] = ap[4 * s];
    }
    for (int s0 = 0; s0 < nk; s0 += SKINNY_DEPTH) {
#pragma unroll
        for (int d = 0; d < SKINNY_DEPTH; ++d) {
            if (s0 + d < nk) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[d], af[d], acc, 0, 0, 0);
            int s = s0 + d + SKINNY_DEPTH;
            s = s < nk ? s : nk - 1;                             // past the end: a harmless re-read, never multiplied
            wf[d] = wp[4 * s];
            af[d] = ap[4 * s];
        }
    }

    // epilogue (gemm.hpp's, per lane: row m = m0 + fr, columns n0 + 4 fg .. + 3)
    f32x4 v;
    if (epi_is_ln(EPI)) {
        const f32x2 st = ln_row_stats(part, K >> 8, K);
        v = ln_apply(acc, st.x, st.y, cs, bz);
    } else {
        v = acc + bz;
    }
    if (epi_is_bf16_out(EPI)) {
        if (epi_is_qgelu(EPI)) v = quick_gelu4(v);
        if (valid)
            *reinterpret_cast<uint2*>(static_cast<unsigned short*>(g.out) + orow * g.N + n) =
                make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
        return;
    }
    if (EPI == EPI_BIAS_RESID_LN_F32) {
        // (acc + bias) + old row - the tiled producers' add order - then the split form in place and this lane's leaf
        const f32x4 o = v + split_join(xh, xl);
        uint2 nh;
        unsigned nl;
        split_make(o, nh, nl);
        if (valid) {
            *reinterpret_cast<uint2*>(resid_hi(g.x3, orow, g.N) + n) = nh;
            *reinterpret_cast<unsigned*>(resid_lo(g.x3, orow, g.N) + n) = nl;
            *reinterpret_cast<f32x2*>(g.ln_leaf + (orow * (size_t)(g.N >> 2) + (size_t)(n >> 2)) * 2) = f32x2{ln_lane_sum(o), ln_lane_sumsq(o)};
        }
        return;
    }
    if (EPI == EPI_PATCH_F32 || EPI == EPI_BIAS_RESID_F32) v += add;
    if (valid) *reinterpret_cast<f32x4*>(static_cast<float*>(g.out) + orow * g.N + n) = v;
}

}  // namespace clipmi
