// gemm.hip — instantiation and launch of the bf16 NT GEMM (see gemm.hpp).
#include "gemm256p.hpp"
#include "gemm_skinny.hpp"
#ifdef CLIPMI_DEV
#endif
#include "gemm256f8.hpp"
#include <hip/hip_ext.h>
#include <cstdlib>

namespace clipmi {

// LDS opt-in (hipFuncSetAttribute) is remembered per (kernel, device): a second device in the same thread gets its own
static bool lds_opted(int* slots) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return false;
    if (slots[dev]) return true;
    slots[dev] = 1;
    return false;
}

template <int EPI>
static int launch_epi(const GemmArgs& g, hipStream_t st, GemmProbe* probe) {
    const int grid = (g.N / GEMM_BN) * ((g.M + GEMM_BM - 1) / GEMM_BM);
    if (probe && probe->wants(EPI)) {
        GemmProbe& p = *probe;
        // measurement probe: the events take the dispatch's own begin/end timestamps
        const int i = p.begin(EPI, 0, st, g.K);
        hipExtLaunchKernelGGL(gemm_bf16_nt_kernel<EPI>, dim3(grid), dim3(256), GEMM_LDS_BYTES, st, p.ev[2 * i],
                              p.ev[2 * i + 1], 0, g);
    } else {
        hipLaunchKernelGGL(gemm_bf16_nt_kernel<EPI>, dim3(grid), dim3(256), GEMM_LDS_BYTES, st, g);
    }
    CLIPMI_CHECK_LAUNCH("gemm_bf16_nt_kernel");
    return 0;
}

template <int EPI>
static int launch_epi256(const GemmArgs& g, hipStream_t st, GemmProbe* probe) {
    const int grid = (g.N / 256) * ((g.M + 255) / 256);
    static thread_local int opted[64];
    if (!lds_opted(opted)) {
        if (hipFuncSetAttribute((const void*)gemm256_bf16_nt_kernel<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                G256_LDS) != hipSuccess)
            return set_err(CLIPMI_EHIP, "hipFuncSetAttribute(gemm256, %d B LDS)", G256_LDS);
    }
    if (probe && probe->wants(EPI)) {
        GemmProbe& p = *probe;
        const int i = p.begin(EPI, 1, st, g.K);
        hipExtLaunchKernelGGL(gemm256_bf16_nt_kernel<EPI>, dim3(grid), dim3(512), G256_LDS, st, p.ev[2 * i],
                              p.ev[2 * i + 1], 0, g);
    } else {
        hipLaunchKernelGGL(gemm256_bf16_nt_kernel<EPI>, dim3(grid), dim3(512), G256_LDS, st, g);
    }
    CLIPMI_CHECK_LAUNCH("gemm256_bf16_nt_kernel");
    return 0;
}

#ifdef CLIPMI_DEV
// development: the W-direct form of gemm256 (gemm256.hpp WD = true), algo 5 of the test hooks
template <int EPI, int WD = 1>
static int launch_epi256wd(const GemmArgs& g, hipStream_t st) {
    const int grid = (g.N / 256) * ((g.M + 255) / 256);
    static thread_local int opted[64];
    if (!lds_opted(opted)) {
        if (hipFuncSetAttribute((const void*)gemm256_bf16_nt_kernel<EPI, WD>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                G256_LDS) != hipSuccess)
            return set_err(CLIPMI_EHIP, "hipFuncSetAttribute(gemm256 WD, %d B LDS)", G256_LDS);
    }
    hipLaunchKernelGGL((gemm256_bf16_nt_kernel<EPI, WD>), dim3(grid), dim3(512), G256_LDS, st, g);
    CLIPMI_CHECK_LAUNCH("gemm256_bf16_nt_kernel<WD>");
    return 0;
}
#endif

// persistent, role-split 256x256 kernel (gemm256p.hpp): pure-store epilogues only
static int persist_mode() {
    // development switch for A/B runs on one box: CLIPMI_GEMM_PERSIST=0 keeps every GEMM on gemm256
    static const int mode = (int)dev_knob("CLIPMI_GEMM_PERSIST", 1);
    return mode;
}

// LDS beyond the two K-tile buffers: bias[N]; FP8: w_scale[N] + the tile's 256 a_scale values (1 KiB); LN consumer: the
// tile's 256 colsum values (1 KiB) + K/256 statistics pairs per row (+ 8 B where the 4-pair read of the last row runs
// past them)
constexpr int g256p_lds(int epi, bool fp8, int N, int K, int dbg) {
    const int nseg = K >> 8;
    return G256_LDS + N * 4 * (fp8 ? 2 : 1) + (fp8 ? 1024 : epi_is_ln(epi) ? 1024 + 256 * nseg * 8 + (nseg < 4 ? 8 : 0) : 0) +
           ((dbg & 12) ? 8192 + 2048 : 0);
}

template <int EPI, bool FP8 = false>
static int launch_epi256p(const GemmArgs& g, hipStream_t st, GemmProbe* probe) {
    const int tiles = (g.N / 256) * ((g.M + 255) / 256);
    static const int grid_max = (int)dev_knob("CLIPMI_GEMM_GRID", NUM_CU);      // development: fewer workgroups than CUs
    const int grid = tiles < grid_max ? tiles : grid_max;
    const int lds = g256p_lds(EPI, FP8, g.N, g.K, g.dbg);
    if (lds > LDS_BYTES) return set_err(CLIPMI_EUNSUPPORTED, "gemm256p: %d B of LDS for N=%d", lds, g.N);
    static thread_local int opted[64];
    if (!lds_opted(opted)) {
        if (hipFuncSetAttribute((const void*)gemm256p_bf16_nt_kernel<EPI, FP8>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS_BYTES) != hipSuccess)
            return set_err(CLIPMI_EHIP, "hipFuncSetAttribute(gemm256p, %d B LDS)", LDS_BYTES);
    }
    if (probe && probe->wants(EPI)) {
        GemmProbe& p = *probe;
        const int i = p.begin(EPI, 2, st, g.K);
        hipExtLaunchKernelGGL((gemm256p_bf16_nt_kernel<EPI, FP8>), dim3(grid), dim3(512), lds, st, p.ev[2 * i],
                              p.ev[2 * i + 1], 0, g);
    } else {
        hipLaunchKernelGGL((gemm256p_bf16_nt_kernel<EPI, FP8>), dim3(grid), dim3(512), lds, st, g);
    }
    CLIPMI_CHECK_LAUNCH("gemm256p_bf16_nt_kernel");
    return 0;
}

// The residual producer of the LN-folded layers (EPI_BIAS_RESID_LN_F32) updates the split residual rows (x3) and leaves
// the new rows' statistics partials in g.ln_part. The persistent kernel's storers do all of it on their way out; every
// other kernel writes acc + bias as f32 into g.tmp_f32 and split_stats_kernel (add form) does the rest in one
// LayerNorm-sized pass. Both give the same bits (same adds in the same order, canonical statistics: gemm.hpp).

static int split_pct() {        // CLIPMI_GEMM_SPLIT=<pct>: split off the last round when it is less than pct % full (0: never)
    static const int pct = (int)dev_knob("CLIPMI_GEMM_SPLIT", 40);
    return pct;
}

// skinny kernel (gemm_skinny.hpp): M <= 128 rows, one wave per 16 columns x 16 rows
static bool skinny_ok(const GemmArgs& g) {
    static const bool off = dev_knob("CLIPMI_GEMM_SKINNY", 1) == 0;   // A/B aid
    return !off && g.M >= 1 && g.M <= SKINNY_MAX_M && g.N % 16 == 0 && g.K % 32 == 0 && g.K >= 32;
}

template <int EPI>
static int launch_skinny_t(const GemmArgs& g, hipStream_t st) {
    const int nstrips = g.N / 16, mtiles = (g.M + 15) / 16;
    const int waves = nstrips * mtiles <= 4 * NUM_CU ? 1 : 4;       // one wave per workgroup until every CU has four
    const dim3 grid((nstrips + waves - 1) / waves, mtiles);
    if (g.K > 512) hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 32>), grid, dim3(waves * 64), 0, st, g);
    else hipLaunchKernelGGL((gemm_skinny_kernel<EPI, 16>), grid, dim3(waves * 64), 0, st, g);
    CLIPMI_CHECK_LAUNCH("gemm_skinny_kernel");
    return 0;
}

bool gemm_resid_writes_leaves(int M, int N, int K) {
    GemmArgs g{};
    g.M = M; g.N = N; g.K = K;
    return skinny_ok(g) && N % 256 == 0 && N <= 1024;
}

static int launch_skinny(const GemmArgs& g, int epi, hipStream_t st) {
    if (!g.A || !g.W || (!g.out && epi != EPI_BIAS_RESID_LN_F32)) return set_err(CLIPMI_EINVAL, "gemm: NULL pointer");
    switch (epi) {
        case EPI_BIAS_BF16: return launch_skinny_t<EPI_BIAS_BF16>(g, st);
        case EPI_BIAS_QGELU_BF16: return launch_skinny_t<EPI_BIAS_QGELU_BF16>(g, st);
        case EPI_BIAS_RESID_F32: return launch_skinny_t<EPI_BIAS_RESID_F32>(g, st);
        case EPI_F32: return launch_skinny_t<EPI_F32>(g, st);
        case EPI_PATCH_F32: return launch_skinny_t<EPI_PATCH_F32>(g, st);
        case EPI_LN_BIAS_BF16: return launch_skinny_t<EPI_LN_BIAS_BF16>(g, st);
        case EPI_LN_BIAS_QGELU_BF16: return launch_skinny_t<EPI_LN_BIAS_QGELU_BF16>(g, st);
        case EPI_BIAS_RESID_LN_F32: return launch_skinny_t<EPI_BIAS_RESID_LN_F32>(g, st);     // in place + statistics leaves
    }
    return set_err(CLIPMI_EINVAL, "gemm_skinny: epilogue %d", epi);
}

// algo: 0 = choose by shape, 1 = force the 128x128 kernel, 2 = force the 256x256 kernel,
//       3 = force the persistent 256x256 kernel
int launch_gemm_algo(const GemmArgs& g, int epi, int algo, hipStream_t st, GemmProbe* probe) {
    if (epi_is_ln(epi) && ((!g.ln_part_in && !g.ln_leaf_in) || !g.colsum || !g.bias || g.K % 256 != 0 || g.K > 1024))
        return set_err(CLIPMI_EINVAL, "gemm: LN-folded epilogue %d needs ln_part_in, colsum, bias and K %% 256 == 0, K <= 1024", epi);
    if (epi == EPI_BIAS_RESID_LN_F32 &&
        (!g.x3 || !g.ln_part || !g.tmp_f32 || g.N % 256 != 0 || g.N > 1024 || ((size_t)g.x3 & 15) != 0))
        return set_err(CLIPMI_EINVAL, "gemm: EPI_BIAS_RESID_LN_F32 needs x3 (16-byte aligned), ln_part, tmp_f32 and N %% 256 == 0, N <= 1024");
    if (g.lda_bytes && (g.lda_bytes % 16 != 0 || g.lda_bytes < 2u * (unsigned)g.K))
        return set_err(CLIPMI_EINVAL, "gemm: lda_bytes %u (need a multiple of 16, >= 2 K)", g.lda_bytes);
#ifndef CLIPMI_DEV
    if (algo == 4 || algo == 5) return set_err(CLIPMI_EUNSUPPORTED, "algo %d exists in the development build only (libclipmi_dev.so)", algo);
#else
    if (algo == 5) {
        if (g.N % 256 != 0 || g.K % 64 != 0 || g.K < 128 || g.M < 1 || !g.A || !g.W || !g.out)
            return set_err(CLIPMI_EINVAL, "gemm256 WD: N %% 256 == 0, K %% 64 == 0, K >= 128");
        if (epi == EPI_BIAS_BF16) return launch_epi256wd<EPI_BIAS_BF16>(g, st);
        if (epi == EPI_BIAS_QGELU_BF16) return launch_epi256wd<EPI_BIAS_QGELU_BF16>(g, st);
        if (epi == EPI_F32) return launch_epi256wd<EPI_F32>(g, st);
        return set_err(CLIPMI_EINVAL, "gemm256 WD: epilogue %d", epi);
    }
    if (algo == 4) return set_err(CLIPMI_EUNSUPPORTED, "algo 4 (gemm2w, DESIGN 4.4g) was removed in round 5: measured slower in round 3, its sources are in the history");
#endif
    const bool ok256 = g.N % 256 == 0 && g.K % 64 == 0 && g.K >= 128;
    if (algo == 2 && !ok256) return set_err(CLIPMI_EINVAL, "gemm256: N=%d K=%d (need N %% 256 == 0, K %% 64 == 0, K >= 128)", g.N, g.K);
    const bool store_only = epi == EPI_BIAS_BF16 || epi == EPI_BIAS_QGELU_BF16 || epi == EPI_BIAS_RESID_F32 || epi_is_ln(epi) ||
                            epi == EPI_BIAS_RESID_LN_F32;
    const bool ok256p = ok256 && g.K % 128 == 0 && g.N <= G256P_MAX_N && store_only && g.K <= (1 << 20) &&
                        g256p_lds(epi, false, g.N, g.K, g.dbg) <= LDS_BYTES && (!epi_is_ln(epi) || (g.K % 256 == 0 && g.K <= 1024));
    if (algo == 3 && !ok256p)
        return set_err(CLIPMI_EINVAL, "gemm256p: M=%d N=%d K=%d epi=%d (need N %% 256 == 0, K %% 128 == 0, a store-only epilogue "
                       "and bias/colsum rows that fit LDS)", g.M, g.N, g.K, epi);
    // by shape: the 256x256 pipeline wins when its tiles fill the 256 CUs in whole rounds (r01 on MI355X,
    // M=25600: N=2304/3072 -> 781/841 TF vs 702/685; N=768 -> 300 tiles = 1.17 rounds, 408/769 vs 521/904)
    bool use256 = algo == 2 || algo == 3;
    bool use256p = algo == 3;
    if (algo == 0 && ok256 && g.M >= 1024) {
        const long long tiles = (long long)(g.N / 256) * ((g.M + 255) / 256);
        const long long rounds = (tiles + NUM_CU - 1) / NUM_CU;
        // calibrated with tools/gemm_rule.py: the 256x256 kernel wins from ~0.74-0.78 fill of its rounds
        // (earlier for long K, where its mainloop advantage outweighs the idle CUs of the last round)
        const long long pct = g.K >= 2048 ? 72 : 78;
        use256 = tiles * 100 >= rounds * NUM_CU * pct;
        // more than one round of tiles: the persistent kernel overlaps each tile's write-out with the next
        // tile's K-loop
        // (the split-residual producer also at <= one round: its fused store pass saves the whole split_stats pass that
        //  the other kernels need behind them; CLIPMI_GEMM_PERSIST=2 keeps the old rule for A/B runs)
        use256p = use256 && ok256p && persist_mode() != 0 &&
                  (tiles > NUM_CU || (epi == EPI_BIAS_RESID_LN_F32 && persist_mode() != 2));
        // A ragged last round (M = 25600 at N = 768: 300 tiles = 1.17 rounds; 78 k images/s at B = 512 against 102 k at 435 /
        // 870): the rows that fill WHOLE rounds of 256 x 256 tiles go to the persistent kernel, the remaining row tiles to
        // whatever the rule picks for them alone (128 x 128 tiles: a quarter of a big tile's time per round), instead of the
        // whole GEMM falling back. Same bits either way (every kernel's rows are independent and bit-identical). Not for the
        // patch GEMM (its epilogue maps the absolute row number).
        if (ok256p && persist_mode() != 0 && tiles > NUM_CU && epi != EPI_PATCH_F32 && (tiles % NUM_CU) != 0 &&
            (tiles % NUM_CU) * 100 < (long long)NUM_CU * split_pct()) {
            const int ntn = g.N / 256;
            const long long main_tiles = (tiles / NUM_CU) * NUM_CU;          // whole rounds
            const int rows_main = (int)(main_tiles / ntn) * 256;             // < M: the last round was ragged
            if (rows_main >= 256 && rows_main < g.M) {
                GemmArgs a = g, b = g;
                a.M = rows_main;
                b.M = g.M - rows_main;
                b.A = g.A + (size_t)rows_main * gemm_lda(g);
                const size_t oe = (size_t)rows_main * g.N;
                if (epi == EPI_BIAS_RESID_LN_F32) {
                    b.x3 = static_cast<char*>(g.x3) + (size_t)rows_main * resid_row_bytes(g.N); b.tmp_f32 = g.tmp_f32 + oe;
                    b.ln_part = g.ln_part + (size_t)rows_main * (g.N / 256) * 2;
                } else if (epi_is_bf16_out(epi)) {
                    b.out = static_cast<unsigned short*>(g.out) + oe;
                } else {
                    b.out = static_cast<float*>(g.out) + oe;
                }
                if (b.ln_part_in) b.ln_part_in = g.ln_part_in + (size_t)rows_main * (g.K / 256) * 2;
                if (int rc = launch_gemm_algo(a, epi, 3, st, probe)) return rc;
                return launch_gemm_algo(b, epi, 0, st, probe);
            }
        }
    }
    if (g.M < 1 || !g.A || !g.W || (!g.out && epi != EPI_BIAS_RESID_LN_F32)) return set_err(CLIPMI_EINVAL, "gemm: bad arguments");
    // a handful of rows (one prompt, one image): the skinny kernel, whatever the epilogue (the residual producer as
    // acc + bias into the f32 scratch rows + the split / statistics pass, like every non-persistent kernel)
    const bool skinny = algo == 0 && skinny_ok(g);
    if (skinny && epi != EPI_BIAS_RESID_LN_F32) return launch_skinny(g, epi, st);
    // one prompt / one image with a leaf buffer: the skinny kernel updates the split rows itself and leaves the statistics as
    // leaves for the skinny LN-folded consumer behind it (no scratch rows, no split / statistics launch)
    if (epi == EPI_BIAS_RESID_LN_F32 && skinny && g.ln_leaf) return launch_skinny(g, epi, st);
    if (g.ln_leaf_in && !(skinny && epi_is_ln(epi)))
        return set_err(CLIPMI_EINVAL, "gemm: ln_leaf_in is read by the skinny LN-folded consumers only (M <= %d)", SKINNY_MAX_M);
    if (epi == EPI_BIAS_RESID_LN_F32 && !use256p) {
        // not the persistent kernel: acc + bias as f32 into the scratch rows, then the add + split + statistics pass
        GemmArgs t = g;
        t.out = g.tmp_f32;
        GemmProbe* pr = (probe && probe->wants(EPI_BIAS_RESID_LN_F32)) ? probe : nullptr;
        const int saved_epi = pr ? pr->epi : -1;
        if (pr) pr->epi = EPI_F32;                       // the probe brackets the GEMM launch itself
        const int rc = skinny ? launch_skinny(t, EPI_F32, st) : use256 ? launch_epi256<EPI_F32>(t, st, pr) : launch_gemm(t, EPI_F32, st, pr);
        if (pr) pr->epi = saved_epi;
        if (rc) return rc;
        return launch_split_stats(g.tmp_f32, true, g.x3, g.ln_part, g.M, g.N, st);
    }
    if (!use256) return launch_gemm(g, epi, st, probe);
    if (use256p) {
        switch (epi) {
            case EPI_BIAS_BF16: return launch_epi256p<EPI_BIAS_BF16>(g, st, probe);
            case EPI_BIAS_QGELU_BF16: return launch_epi256p<EPI_BIAS_QGELU_BF16>(g, st, probe);
            case EPI_BIAS_RESID_F32: return launch_epi256p<EPI_BIAS_RESID_F32>(g, st, probe);
            case EPI_LN_BIAS_BF16: return launch_epi256p<EPI_LN_BIAS_BF16>(g, st, probe);
            case EPI_LN_BIAS_QGELU_BF16: return launch_epi256p<EPI_LN_BIAS_QGELU_BF16>(g, st, probe);
            case EPI_BIAS_RESID_LN_F32: return launch_epi256p<EPI_BIAS_RESID_LN_F32>(g, st, probe);
        }
        return set_err(CLIPMI_EINVAL, "gemm256p: epilogue %d", epi);
    }
    switch (epi) {
        case EPI_BIAS_BF16: return launch_epi256<EPI_BIAS_BF16>(g, st, probe);
        case EPI_BIAS_QGELU_BF16: return launch_epi256<EPI_BIAS_QGELU_BF16>(g, st, probe);
        case EPI_BIAS_RESID_F32: return launch_epi256<EPI_BIAS_RESID_F32>(g, st, probe);
        case EPI_F32: return launch_epi256<EPI_F32>(g, st, probe);
        case EPI_PATCH_F32: return launch_epi256<EPI_PATCH_F32>(g, st, probe);
        case EPI_LN_BIAS_BF16: return launch_epi256<EPI_LN_BIAS_BF16>(g, st, probe);
        case EPI_LN_BIAS_QGELU_BF16: return launch_epi256<EPI_LN_BIAS_QGELU_BF16>(g, st, probe);
    }
    return set_err(CLIPMI_EINVAL, "gemm: unknown epilogue %d", epi);
}

template <int EPI, bool MX>
static int launch_epi256f8(const GemmArgs& g, hipStream_t st) {
    const int grid = (g.N / 256) * ((g.M + 255) / 256);
    static thread_local int opted[64];
    if (!lds_opted(opted)) {
        if (hipFuncSetAttribute((const void*)gemm256f8_nt_kernel<EPI, MX>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                G256_LDS) != hipSuccess)
            return set_err(CLIPMI_EHIP, "hipFuncSetAttribute(gemm256f8, %d B LDS)", G256_LDS);
    }
    hipLaunchKernelGGL((gemm256f8_nt_kernel<EPI, MX>), dim3(grid), dim3(512), G256_LDS, st, g);
    CLIPMI_CHECK_LAUNCH("gemm256f8_nt_kernel");
    return 0;
}

// block-scaled activations (gemm256f8.hpp BSA): the tile's 256 x K/32 scale bytes sit behind the K-tile buffers
constexpr int G256F8_BSA_MAX_K = 4096;           // 32 KiB of scale bytes per tile: 160 KiB of LDS in all
template <int EPI>
static int launch_epi256f8_bsa(const GemmArgs& g, hipStream_t st) {
    const int grid = (g.N / 256) * ((g.M + 255) / 256);
    const int lds = G256_LDS + 256 * (g.K / 32);
    static thread_local int opted[64];
    if (!lds_opted(opted)) {
        if (hipFuncSetAttribute((const void*)gemm256f8_nt_kernel<EPI, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                G256_LDS + 256 * (G256F8_BSA_MAX_K / 32)) != hipSuccess)
            return set_err(CLIPMI_EHIP, "hipFuncSetAttribute(gemm256f8 block-scaled)");
    }
    hipLaunchKernelGGL((gemm256f8_nt_kernel<EPI, true, true>), dim3(grid), dim3(512), lds, st, g);
    CLIPMI_CHECK_LAUNCH("gemm256f8_nt_kernel(block-scaled A)");
    return 0;
}

// true when launch_gemm_fp8(.., EPI_BIAS_QGELU_BF16, ..) of this shape runs on the persistent kernel, whose store pass can
// emit e4m3 + MX block scales (GemmArgs::out_bscale) instead of bf16
bool gemm_fp8_emits_mx(int M, int N, int K) {
    const long long tiles = (long long)(N / 256) * ((M + 255) / 256);
    return M >= 1 && N % 256 == 0 && K % 256 == 0 && K >= 256 && tiles > NUM_CU && N <= 3840 && persist_mode() != 0;
}

// mx: 1 = the block-scaled MFMA form with unit scales (twice the rate), 0 = plain FP8 MFMA (f32 accumulation)
int launch_gemm_fp8(const GemmArgs& g, int epi, hipStream_t st, int mx) {
    if (g.M < 1 || g.N % 256 != 0 || g.K % 128 != 0 || g.K < 256)
        return set_err(CLIPMI_EINVAL, "gemm_fp8: M=%d N=%d K=%d (need N %% 256 == 0, K %% 128 == 0, K >= 256)", g.M, g.N, g.K);
    if (!g.A || !g.W || !g.out || (!g.a_scale && !g.a_bscale) || !g.w_scale) return set_err(CLIPMI_EINVAL, "gemm_fp8: NULL pointer");
    if (g.a_bscale) {            // MX activations: per-32-block e8m0 scales instead of a row scale
        if (g.K > G256F8_BSA_MAX_K) return set_err(CLIPMI_EINVAL, "gemm_fp8: block-scaled A needs K <= %d", G256F8_BSA_MAX_K);
        if (epi == EPI_BIAS_RESID_F32) return launch_epi256f8_bsa<EPI_BIAS_RESID_F32>(g, st);
        if (epi == EPI_F32) return launch_epi256f8_bsa<EPI_F32>(g, st);
#ifdef CLIPMI_DEV
        // FP8 tower with folded LayerNorms (round 4, development library: DESIGN 4.4c - parity-green, but slower than the
        // LayerNorm-pass tower while its GEMMs run on this non-persistent kernel): the residual producer that also emits
        // e4m3 + MX scales + statistics, and the consumers whose A operand those are
        if (epi == EPI_BIAS_RESID_LN8) {
            if (!g.x8 || !g.x8_bs || !g.ln_part || g.N > 1024)
                return set_err(CLIPMI_EINVAL, "gemm_fp8: EPI_BIAS_RESID_LN8 needs x8, x8_bs, ln_part and N <= 1024");
            return launch_epi256f8_bsa<EPI_BIAS_RESID_LN8>(g, st);
        }
        if (epi_is_ln(epi)) {
            if (!g.ln_part_in || !g.colsum || !g.bias || g.K % 256 != 0 || g.K > 1024)
                return set_err(CLIPMI_EINVAL, "gemm_fp8: LN-folded epilogue %d needs ln_part_in, colsum, bias, K %% 256 == 0, K <= 1024", epi);
            if (g.out_bscale && epi != EPI_LN_BIAS_QGELU_BF16)
                return set_err(CLIPMI_EINVAL, "gemm_fp8: e4m3 + block-scale output exists for the QuickGELU forms only");
            return epi == EPI_LN_BIAS_BF16 ? launch_epi256f8_bsa<EPI_LN_BIAS_BF16>(g, st) : launch_epi256f8_bsa<EPI_LN_BIAS_QGELU_BF16>(g, st);
        }
#endif
        return set_err(CLIPMI_EINVAL, "gemm_fp8: block-scaled A with epilogue %d", epi);
    }
    // more than one round of tiles: the persistent role-split kernel on FP8 operands (mx = 2 keeps gemm256f8 for tests)
    const long long tiles = (long long)(g.N / 256) * ((g.M + 255) / 256);
    if (g.out_bscale && !(mx == 1 && epi == EPI_BIAS_QGELU_BF16 && gemm_fp8_emits_mx(g.M, g.N, g.K)))
        return set_err(CLIPMI_EINVAL, "gemm_fp8: e4m3 + block-scale output exists for the persistent QuickGELU form only");
    if (mx == 1 && tiles > NUM_CU && g.K % 256 == 0 && g.N <= 3840 && persist_mode() != 0) {
        if (epi == EPI_BIAS_BF16) return launch_epi256p<EPI_BIAS_BF16, true>(g, st, nullptr);
        if (epi == EPI_BIAS_QGELU_BF16) return launch_epi256p<EPI_BIAS_QGELU_BF16, true>(g, st, nullptr);
        // (the residual epilogue stays on gemm256f8: its persistent FP8 form does not fit 256 registers)
    }
    switch (epi) {
        case EPI_BIAS_BF16: return mx ? launch_epi256f8<EPI_BIAS_BF16, true>(g, st) : launch_epi256f8<EPI_BIAS_BF16, false>(g, st);
        case EPI_BIAS_QGELU_BF16:
            return mx ? launch_epi256f8<EPI_BIAS_QGELU_BF16, true>(g, st) : launch_epi256f8<EPI_BIAS_QGELU_BF16, false>(g, st);
        case EPI_BIAS_RESID_F32:
            return mx ? launch_epi256f8<EPI_BIAS_RESID_F32, true>(g, st) : launch_epi256f8<EPI_BIAS_RESID_F32, false>(g, st);
        case EPI_F32: return mx ? launch_epi256f8<EPI_F32, true>(g, st) : launch_epi256f8<EPI_F32, false>(g, st);
    }
    return set_err(CLIPMI_EINVAL, "gemm_fp8: epilogue %d", epi);
}

int launch_gemm(const GemmArgs& g, int epi, hipStream_t st, GemmProbe* probe) {
    if (g.M < 1 || g.N < 1 || g.K < 1 || g.N % GEMM_BN != 0 || g.K % GEMM_BK != 0)
        return set_err(CLIPMI_EINVAL, "gemm: M=%d N=%d K=%d (need N %% 128 == 0, K %% 64 == 0)", g.M, g.N, g.K);
    if (!g.A || !g.W || !g.out) return set_err(CLIPMI_EINVAL, "gemm: NULL pointer");
    switch (epi) {
        case EPI_BIAS_BF16: return launch_epi<EPI_BIAS_BF16>(g, st, probe);
        case EPI_BIAS_QGELU_BF16: return launch_epi<EPI_BIAS_QGELU_BF16>(g, st, probe);
        case EPI_BIAS_RESID_F32: return launch_epi<EPI_BIAS_RESID_F32>(g, st, probe);
        case EPI_F32: return launch_epi<EPI_F32>(g, st, probe);
        case EPI_PATCH_F32: return launch_epi<EPI_PATCH_F32>(g, st, probe);
        case EPI_LN_BIAS_BF16: return launch_epi<EPI_LN_BIAS_BF16>(g, st, probe);
        case EPI_LN_BIAS_QGELU_BF16: return launch_epi<EPI_LN_BIAS_QGELU_BF16>(g, st, probe);
    }
    return set_err(CLIPMI_EINVAL, "gemm: unknown epilogue %d", epi);
}

}  // namespace clipmi

using namespace clipmi;

extern "C" int clipmi_dbg_gemm_bf16(const void* a_dev, const void* w_dev, const float* bias_dev, void* out_dev, int M,
                                    int N, int K, int epi, void* stream) {
    const int algo = (epi >> 8) & 7;      // test hook: bits 8-10 force a kernel (see launch_gemm_algo)
    epi &= 0xff;
    if (epi < 0 || epi > 3) return set_err(CLIPMI_EINVAL, "dbg_gemm: epi %d", epi);
    GemmArgs g{};
    g.A = static_cast<const unsigned short*>(a_dev);
    g.W = static_cast<const unsigned short*>(w_dev);
    g.bias = bias_dev;
    g.out = out_dev;
    g.M = M; g.N = N; g.K = K;
    static const int dbg_env = (int)dev_knob("CLIPMI_GEMM_DBG", 0);
    g.dbg = dbg_env;
    if (g.dbg & 12) { g.pos = bias_dev; g.bias = nullptr; }     // stamps land in the caller's "bias" buffer (>= 4 KiB)
    return launch_gemm_algo(g, epi, algo, as_stream(stream));
}

// Test hooks of the LN-folded layers (gemm.hpp). `epi` = EPI_LN_BIAS_BF16 (5) or EPI_LN_BIAS_QGELU_BF16 (6), bits 8-9
// force a kernel as in clipmi_dbg_gemm_bf16. part_dev: [M][K/256][2] statistics partials of the split rows x3_dev ([K bf16 hi | K u8 lo] each).
extern "C" int clipmi_dbg_gemm_ln(const void* x3_dev, const void* wg_dev, const float* cb_dev, const float* colsum_dev,
                                  const float* part_dev, void* out_dev, int M, int N, int K, int epi, void* stream) {
    const int algo = (epi >> 8) & 3;
    epi &= 0xff;
    if (!epi_is_ln(epi)) return set_err(CLIPMI_EINVAL, "dbg_gemm_ln: epi %d", epi);
    GemmArgs g{};
    g.A = static_cast<const unsigned short*>(x3_dev);                 // the hi halves of the split rows are the A operand
    g.lda_bytes = (unsigned)resid_row_bytes(K);
    g.W = static_cast<const unsigned short*>(wg_dev);
    g.bias = cb_dev; g.colsum = colsum_dev; g.ln_part_in = part_dev;
    g.out = out_dev;
    g.M = M; g.N = N; g.K = K;
    return launch_gemm_algo(g, epi, algo, as_stream(stream));
}

// One prompt / one image (M <= 128, the skinny kernels): clipmi_dbg_gemm_resid_ln_leaf updates x3 in place and writes the statistics
// leaves leaf_dev [M][N / 4][2] (no scratch rows, no partials); clipmi_dbg_gemm_ln_leaf is the LN-folded consumer reading them.
// Together they return the bits of clipmi_dbg_gemm_resid_ln + clipmi_dbg_gemm_ln.
extern "C" int clipmi_dbg_gemm_resid_ln_leaf(const void* a_dev, const void* w_dev, const float* bias_dev, void* x3_dev, float* leaf_dev,
                                             int M, int N, int K, void* stream) {
    if (!leaf_dev || !gemm_resid_writes_leaves(M, N, K))
        return set_err(CLIPMI_EINVAL, "dbg_gemm_resid_ln_leaf: M <= %d, N %% 256 == 0, N <= 1024, K %% 32 == 0", SKINNY_MAX_M);
    GemmArgs g{};
    g.A = static_cast<const unsigned short*>(a_dev);
    g.W = static_cast<const unsigned short*>(w_dev);
    g.bias = bias_dev;
    g.x3 = x3_dev; g.ln_leaf = leaf_dev;
    g.ln_part = leaf_dev; g.tmp_f32 = leaf_dev;          // unused on this path (the argument check wants them non-null)
    g.M = M; g.N = N; g.K = K;
    return launch_gemm_algo(g, EPI_BIAS_RESID_LN_F32, 0, as_stream(stream));
}

extern "C" int clipmi_dbg_gemm_ln_leaf(const void* x3_dev, const void* wg_dev, const float* cb_dev, const float* colsum_dev,
                                       const float* leaf_dev, void* out_dev, int M, int N, int K, int epi, void* stream) {
    if (!epi_is_ln(epi)) return set_err(CLIPMI_EINVAL, "dbg_gemm_ln_leaf: epi %d", epi);
    GemmArgs g{};
    g.A = static_cast<const unsigned short*>(x3_dev);
    g.lda_bytes = (unsigned)resid_row_bytes(K);
    g.W = static_cast<const unsigned short*>(wg_dev);
    g.bias = cb_dev; g.colsum = colsum_dev; g.ln_leaf_in = leaf_dev;
    g.out = out_dev;
    g.M = M; g.N = N; g.K = K;
    return launch_gemm_algo(g, epi, 0, as_stream(stream));
}

// x3 (split residual rows [N bf16 hi | N u8 lo], updated in place) += a @ w^T + bias; part = statistics partials of the
// new rows [M][N/256][2]; tmp_dev: f32 [M][N] scratch. algo as in clipmi_dbg_gemm_bf16 (3 = the persistent kernel's fused
// store pass; 1 / 2 = GEMM into tmp + split_stats_kernel).
extern "C" int clipmi_dbg_gemm_resid_ln(const void* a_dev, const void* w_dev, const float* bias_dev, void* x3_dev,
                                        float* part_dev, float* tmp_dev, int M, int N, int K, int algo, void* stream) {
    GemmArgs g{};
    g.A = static_cast<const unsigned short*>(a_dev);
    g.W = static_cast<const unsigned short*>(w_dev);
    g.bias = bias_dev;
    g.x3 = x3_dev;
    g.ln_part = part_dev; g.tmp_f32 = tmp_dev;
    g.M = M; g.N = N; g.K = K;
    static const int dbg_env = (int)dev_knob("CLIPMI_GEMM_DBG", 0);
    g.dbg = dbg_env;
    return launch_gemm_algo(g, EPI_BIAS_RESID_LN_F32, algo & 7, as_stream(stream));
}

// block-scaled activations: a_bscale_dev = e8m0 bytes [ceil(M / 256) * 256][K / 32]; epi 2 (residual) or 3 (plain f32)
extern "C" int clipmi_dbg_gemm_fp8_bsa(const void* a8_dev, const void* w8_dev, const void* a_bscale_dev, const float* w_scale_dev,
                                       const float* bias_dev, void* out_dev, int M, int N, int K, int epi, void* stream) {
    GemmArgs g{};
    g.A = static_cast<const unsigned short*>(a8_dev);
    g.W = static_cast<const unsigned short*>(w8_dev);
    g.a_bscale = static_cast<const unsigned char*>(a_bscale_dev); g.w_scale = w_scale_dev;
    g.bias = bias_dev;
    g.out = out_dev;
    g.M = M; g.N = N; g.K = K;
    return launch_gemm_fp8(g, epi, as_stream(stream), 1);
}

extern "C" int clipmi_dbg_gemm_fp8(const void* a8_dev, const void* w8_dev, const float* a_scale_dev, const float* w_scale_dev,
                                   const float* bias_dev, void* out_dev, int M, int N, int K, int epi, void* stream) {
    // test hook: bit 8 selects the plain (non-scaled) FP8 MFMA form, bit 9 keeps the MX form on the non-persistent kernel
    const int mx = (epi >> 8) & 1 ? 0 : ((epi >> 9) & 1 ? 2 : 1);
    epi &= 0xff;
    if (epi < 0 || epi > 3) return set_err(CLIPMI_EINVAL, "dbg_gemm_fp8: epi %d", epi);
    GemmArgs g{};
    g.A = static_cast<const unsigned short*>(a8_dev);
    g.W = static_cast<const unsigned short*>(w8_dev);
    g.a_scale = a_scale_dev; g.w_scale = w_scale_dev;
    g.bias = bias_dev;
    g.out = out_dev;
    g.M = M; g.N = N; g.K = K;
    return launch_gemm_fp8(g, epi, as_stream(stream), mx);
}
