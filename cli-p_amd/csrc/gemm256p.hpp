// gemm256p.hpp — persistent, role-split form of the 256x256x64 bf16 NT GEMM (gemm256.hpp) for the
// pure-store epilogues (bias -> bf16, bias + QuickGELU -> bf16) when a launch has more than one round
// of tiles (ViT-B/32 at B = 435: qkv = 765 tiles, c_fc = 1020 tiles on 256 CUs).
//
// Why: in gemm256 every round of tiles ends with all 256 CUs writing their 128 KiB of output at the same
// moment; the write burst (33.5 MB) drains at ~4.2 TB/s = ~8 us during which no MFMA runs, then the next
// round pays a cold prologue. vmcnt on gfx9 is ONE in-order counter per wave for loads
// and stores, so a wave that has issued stores cannot use a counted wait for younger DMA loads without
// also waiting for the stores' acknowledgements. Hence the split:
//   * waves with wm == 0 ("loaders") issue every LDS-DMA load (4 instructions per half-tile each) and
//     own all counted vmcnt waits; they never issue a store;
//   * waves with wm == 1 ("storers") issue every global store of the epilogue and never wait on vmcnt:
//     their stores drain while the next tile's K-loop runs.
// One workgroup per CU walks tiles v = blockIdx.x, +gridDim.x, ... (same XCD-aware tile map as gemm256);
// the last K-tile of a tile prefetches K-tile 0 of the NEXT tile into the other LDS buffer, so the K-loop
// of the next tile starts warm right after the epilogue has been issued.
//
// Arithmetic is identical to gemm256 (same MFMA order per output element): results are bit-identical.
// Requires N % 256 == 0, K % 128 == 0 (an even number of K-tiles keeps "K-tile t lives in buffer t & 1"
// true across tiles), 256 rows x K x 2 B < 2^31 (lane offsets are 32-bit), N <= 8192 (bias row kept in LDS).
//
// LDS: [buffer 0 | buffer 1] as in gemm256 (128 KiB) + bias[N] f32. The epilogue stages the tile through
// buffer 1 (dead after the last K-tile; buffer 0 holds the next tile's K-tile 0 in flight) in two passes
// of 128 rows x 512 B.
#pragma once
#include "gemm256f8.hpp"

// K-loop ablations of the persistent kernel, TIMING ONLY (wrong results; development library built with
// CLIPMI_EXTRA_CXXFLAGS=-DCLIPMI_GEMM_ABL=n, tools/gpu_gemm_kloop_abl.sh): bit 0 = the second A half's 8 fragment
// reads per K-tile are skipped (stale registers): 16 instead of 24 ds_read_b128 per wave and K-tile = the 256 B of LDS reads
// per MFMA a 128 x 128 wave tile would have; bit 1 = the B-hi half-tile's DMA fetches ONE 16-byte chunk for all lanes (same
// instruction count and LDS writes, a quarter less traffic beyond the CU); bit 2 = the same for A-hi (half the traffic with bit 1).
#ifndef CLIPMI_GEMM_ABL
#define CLIPMI_GEMM_ABL 0
#endif
#ifndef CLIPMI_GEMM_STAMPS
#define CLIPMI_GEMM_STAMPS 0      // 1: in-kernel time stamps (tools/gp_stamps.py); costs a few % of the K-loop
#endif

namespace clipmi {

constexpr int G256P_MAX_N = 8192;

// FP8 = true: e4m3 operands + scales as gemm256f8 (MX form: one unit-scale v_mfma_scale_f32_16x16x128_f8f6f4 per
// accumulator tile and K-tile of 128 bytes); w_scale[N] sits in LDS beside the bias row, the tile's 256 a_scale
// values arrive by one LDS-DMA instruction under the last K-tile.
// 16-byte output store of the store passes. CLIPMI_GEMM_NT_STORE (compile time, development A/B): bit 0 the bf16 outputs of the
// qkv / c_fc forms, bit 1 the split residual - with the non-temporal hint
#ifndef CLIPMI_GEMM_NT_STORE
#define CLIPMI_GEMM_NT_STORE 0
#endif
template <int BIT>
__device__ __forceinline__ void gp_store16(void* dst, uint4 v) {
    typedef unsigned gp_u4v __attribute__((ext_vector_type(4)));
    if constexpr (CLIPMI_GEMM_NT_STORE & BIT) __builtin_nontemporal_store(gp_u4v{v.x, v.y, v.z, v.w}, reinterpret_cast<gp_u4v*>(dst));
    else *reinterpret_cast<uint4*>(dst) = v;
}

template <int EPI, bool FP8 = false>
__global__ void __launch_bounds__(512, 2) gemm256p_bf16_nt_kernel(GemmArgs g) {
    static_assert(EPI == EPI_BIAS_BF16 || EPI == EPI_BIAS_QGELU_BF16 || EPI == EPI_BIAS_RESID_F32 || epi_is_ln(EPI) ||
                      EPI == EPI_BIAS_RESID_LN_F32, "store-only epilogues");
    // LN-folded forms (gemm.hpp "LN-folded linear layers"): LNC = consumer epilogue (the tile's row-statistics partials
    // and its 256 colsum values by LDS-DMA under the last K-tile), RLN = residual producer (the storers also
    // update the split residual hi / lo and write the tile's 256-column row-statistics partials)
    constexpr bool LNC = epi_is_ln(EPI);
    constexpr bool RESID = EPI == EPI_BIAS_RESID_F32 || EPI == EPI_BIAS_RESID_LN_F32;
    constexpr bool RLN = EPI == EPI_BIAS_RESID_LN_F32;
    static_assert(!(FP8 && (LNC || RLN)), "LN-folded epilogues exist for bf16 operands only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bool loader = wm == 0;
    const int fr = lane & 15, fg = lane >> 4;

    const int ntn = g.N >> 8, mtiles = (g.M + 255) >> 8, ntiles = ntn * mtiles;
    const int K = g.K;
    const unsigned KB = FP8 ? (unsigned)K : 2u * (unsigned)K;      // bytes per operand row
    const unsigned LDA = g.lda_bytes ? g.lda_bytes : KB;           // bytes between two A rows (split-residual rows: 3 K)
    const int nk = (int)(KB >> 7);
    const int stride = gridDim.x;
    int v = blockIdx.x;

    // ---- DMA addressing: buffer descriptors (wave-uniform, rebuilt per tile: base = the tile's first A row /
    // W row) + one 32-bit lane offset per piece + a scalar offset for (K-tile, half, piece). Loader wave wn
    // moves rows [32 wn, 32 wn + 32) of every half-tile as four 8-row pieces. A rows past M are clamped to
    // row M-1 (their products are never stored); W needs no clamp (N % 256 == 0).
    const int srow = lane >> 3, spos = lane & 7;
    const unsigned lane_chunk = (unsigned)(spos ^ srow) * 16u;          // row & 7 == srow for every piece
    const unsigned wvoff = (unsigned)(wn * 32 + srow) * KB + lane_chunk;
    unsigned avoff[2][4];                // [A half][piece], bytes from the tile's first A row
    auto tile_origin = [&](int vv, int& mm, int& nn) {
        int bm, bn;
        // (dbg bits 8-15 / 16-23: supertile shape GM x GN of the XCD-local tile order, development A/B; default 8 x 4)
        const int gm_ = (g.dbg >> 8) & 0xff, gn_ = (g.dbg >> 16) & 0xff;
        gemm_tile_coords(vv, ntiles, mtiles, ntn, gm_ ? gm_ : 8, gn_ ? gn_ : 4, bm, bn);
        mm = bm << 8;
        nn = bn << 8;
    };
    __amdgpu_buffer_rsrc_t rsA, rsW;
    auto set_tile = [&](int mm, int nn) {
        const int last = g.M - 1 - mm;                                  // >= 0: last valid local row
        // recompute the lane's eight row indices here (mbcnt = lane id without a live register): hoisted out of
        // the tile loop they, or `lane`, get spilled, and a scratch reload costs every wave a vmcnt(0)
        int sr_;
        asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(sr_));
        sr_ >>= 3;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int lr = h * 128 + wn * 32 + i * 8 + sr_;
                lr = lr < last ? lr : last;
                avoff[h][i] = (unsigned)lr * LDA + lane_chunk;
            }
        rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(g.A) + (size_t)mm * LDA), 0, 0x7fffffff, 0x00020000);
        rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(reinterpret_cast<const char*>(g.W) + (size_t)nn * KB), 0, 0x7fffffff, 0x00020000);
    };
    const int dma_off = wn * 32 * 128;
    const unsigned piece_stride = 8u * KB, half_stride = 128u * KB;
    // issue half-tile H (0 A-lo, 1 A-hi, 2 B-lo, 3 B-hi) of K-tile kt of the current descriptors into `buf`
#define P_ISSUE(H, kt, buf)                                                                                           \
    do {                                                                                                              \
        if (loader) {                                                                                                 \
            __attribute__((address_space(3))) char* d_ =                                                              \
                (__attribute__((address_space(3))) char*)(smem + (buf) * G256_BUF + (H) * G256_HALF + dma_off);       \
            const unsigned ko_ = (unsigned)(kt) * 128u;                                                               \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
                if ((H) == 3 && (CLIPMI_GEMM_ABL & 2))                                                                \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, d_ + i_ * 1024, 16, 0u, 0u, 0, 0);                  \
                else if ((H) == 1 && (CLIPMI_GEMM_ABL & 4))                                                           \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, d_ + i_ * 1024, 16, 0u, 0u, 0, 0);                  \
                else if ((H) < 2)                                                                                     \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, d_ + i_ * 1024, 16, avoff[(H) & 1][i_], ko_, 0, 0); \
                else                                                                                                  \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, d_ + i_ * 1024, 16, wvoff,                          \
                                                             ko_ + ((H) & 1) * half_stride + i_ * piece_stride, 0, 0); \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)
#define P_WAIT(N)                                                                                                     \
    do {                                                                                                              \
        if (loader) wait_vmcnt<N>();                                                                                  \
    } while (0)
    // EPI_BIAS_RESID_F32: the tile's residual rows (f32 out[m0 + r][n0 .. n0+255], 1 KiB each) come in by
    // LDS-DMA too: loader wn moves rows rbase + 4 wn + i, i < 4, into 16-KiB region H of buffer 0, lane-linear
    // (rows past M are clamped to row M-1: never stored).
    __amdgpu_buffer_rsrc_t rsO;
    int last_lr = 0;
    // RLN: a residual row is [N bf16 hi | N u8 lo] (gemm.hpp): a piece = [256 hi values | 256 lo values] of one row and
    // tile, 768 bytes: lanes 0-31 fetch hi, lanes 32-47 lo, lanes 48-63 nothing (the LDS row keeps its 1-KiB pitch). The
    // descriptor sits at the tile's first hi value; the lo lanes' offset to their bytes depends on the tile's column
    // (hi moves 2 bytes per column, lo 1), so the lane offset is rebuilt with the descriptor.
    unsigned resid_lane_off = (unsigned)lane * 16u;
    auto set_out = [&](int mm, int nn) {
        if constexpr (RLN) {
            rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(static_cast<char*>(g.x3) + (size_t)mm * (3u * (unsigned)g.N) + 2u * (unsigned)nn),
                                                    0, 0x7fffffff, 0x00020000);
            last_lr = g.M - 1 - mm;
            int l_;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l_));
            resid_lane_off = l_ < 32 ? (unsigned)l_ * 16u : 2u * (unsigned)g.N - (unsigned)nn + (unsigned)(l_ - 32) * 16u;
        } else if constexpr (RESID) {
            rsO = __builtin_amdgcn_make_buffer_rsrc((void*)(static_cast<float*>(g.out) + (size_t)mm * g.N + nn), 0, 0x7fffffff, 0x00020000);
            last_lr = g.M - 1 - mm;
        }
    };
    const unsigned resid_row_bytes = (unsigned)g.N * (RLN ? 3u : 4u);
#define P_ISSUE_RESID(H, rbase)                                                                                       \
    do {                                                                                                              \
        if (loader) {                                                                                                 \
            __attribute__((address_space(3))) char* d_ =                                                              \
                (__attribute__((address_space(3))) char*)(smem + (H) * G256_HALF + wn * 4096);                        \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                                        \
                int lr_ = (rbase) + wn * 4 + i_;                                                                      \
                lr_ = lr_ < last_lr ? lr_ : last_lr;                                                                  \
                if (!RLN || (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) < 48)            \
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsO, d_ + i_ * 1024, 16, resid_lane_off,                 \
                                                             (unsigned)lr_ * resid_row_bytes, 0, 0);                  \
            }                                                                                                         \
        }                                                                                                             \
    } while (0)

    // ---- fragment read offsets (as gemm256)
    const int sw = fr & 7;
    const int c0 = ((0 + fg) ^ sw) * 16, c1 = ((4 + fg) ^ sw) * 16;
    const int offA = (wm * 64 + fr) * 128;
    const int offB = 2 * G256_HALF + (wn * 32 + fr) * 128;

    f32x4 acc[2][4][2][2];     // [A half][mt][B half][nt]
    i64x2v af[4][2];           // bf16: 16-byte fragments [mt][ks]
    i64x2v bl[2][2], bh[2][2];
    typedef long i64x4v __attribute__((ext_vector_type(4)));
    i64x4v af8[4], bl8[2], bh8[2];   // FP8: both fragments of a lane as ONE 32-byte operand (8 consecutive registers)

#define P_READ_A(base, half)                                                                         \
    _Pragma("unroll") for (int t_ = 0; t_ < 4; ++t_) {                                               \
        const i64x2v lo_ = *reinterpret_cast<const i64x2v*>((base) + (half) * G256_HALF + offA + t_ * 2048 + c0); \
        const i64x2v hi_ = *reinterpret_cast<const i64x2v*>((base) + (half) * G256_HALF + offA + t_ * 2048 + c1); \
        if constexpr (FP8) af8[t_] = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3);                  \
        else { af[t_][0] = lo_; af[t_][1] = hi_; }                                                   \
    }
#define P_READ_B(dst, base, half)                                                                    \
    _Pragma("unroll") for (int t_ = 0; t_ < 2; ++t_) {                                               \
        const i64x2v lo_ = *reinterpret_cast<const i64x2v*>((base) + (half) * G256_HALF + offB + t_ * 2048 + c0); \
        const i64x2v hi_ = *reinterpret_cast<const i64x2v*>((base) + (half) * G256_HALF + offB + t_ * 2048 + c1); \
        if constexpr (FP8) dst##8[t_] = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3);               \
        else { dst[t_][0] = lo_; dst[t_][1] = hi_; }                                                 \
    }
#define P_MFMA(a, bfr, b)                                                                            \
    do {                                                                                             \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                           \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        __builtin_amdgcn_s_setprio(1);                                                               \
        if constexpr (FP8) {                                                                         \
            _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                         \
                _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                   \
                    acc[a][i_][b][j_] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(            \
                        __builtin_bit_cast(i32x8v, bfr##8[j_]), __builtin_bit_cast(i32x8v, af8[i_]), acc[a][i_][b][j_], 0, 0, 0, \
                        0x7f7f7f7f, 0, 0x7f7f7f7f);                                                  \
                }                                                                                    \
        } else {                                                                                     \
            _Pragma("unroll") for (int ks_ = 0; ks_ < 2; ++ks_)                                      \
                _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_)                                     \
                    _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                 \
                        acc[a][i_][b][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                 \
                            __builtin_bit_cast(bf16x8, bfr[j_][ks_]), __builtin_bit_cast(bf16x8, af[i_][ks_]), acc[a][i_][b][j_], 0, 0, 0); \
        }                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        __builtin_amdgcn_s_barrier();                                                                \
    } while (0)
    // one K-tile out of buffer CUR; with PRE, phase p issues half-tile [A-lo, B-lo, B-hi, A-hi][p] of
    // (current offsets, K-tile KT) into buffer NB. W0/W1/W3: loader vmcnt at P0/P1/P3 (-1 = none).
#define P_KSTAMP(i)                                                                                  \
    do {                                                                                             \
        if (CLIPMI_GEMM_STAMPS && (g.dbg & 8)) {                                                     \
            const unsigned long long c_ = clock64();                                                 \
            if (t == 5) kst[i] = c_;                                                                 \
        }                                                                                            \
    } while (0)
    // the DMA slot of a phase: half-tile H of (K-tile KT -> buffer NB), or - residual epilogue, last K-tile
    // of a tile - the residual rows that belong in region H of buffer 0 (sub-pass 0 = rows 0..31 of the tile
    // -> regions A-lo, B-lo; sub-pass 1 = rows 64..95 -> regions A-hi, B-hi; issue order = region order of
    // the K-loop, so the write-after-read distances are the K-loop's)
#define P_SLOT(H, KT, NB)                                                                            \
    do {                                                                                             \
        if (RESID && last_kt) P_ISSUE_RESID(H, ((H) & 1) * 64 + ((H) >> 1) * 16); \
        else P_ISSUE(H, KT, NB);                                                                     \
    } while (0)
#define P_KTILE(CUR, PRE, KT, NB, W0, W1, W3)                                                        \
    do {                                                                                             \
        P_KSTAMP(0);                                                                                 \
        P_READ_B(bl, CUR, 0);                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                           \
        P_READ_A(CUR, 0);                                                                            \
        if (PRE) P_SLOT(0, KT, NB);                                                                 \
        if ((W0) >= 0) P_WAIT(((W0) < 0 ? 0 : (W0)));                                                \
        __builtin_amdgcn_s_barrier();                                                                \
        P_KSTAMP(1);                                                                                \
        P_MFMA(0, bl, 0);                                                                            \
        P_KSTAMP(2);                                                                                \
        P_READ_B(bh, CUR, 1);                                                                        \
        if (PRE) P_SLOT(2, KT, NB);                                                                 \
        if ((W1) >= 0) P_WAIT(((W1) < 0 ? 0 : (W1)));                                                \
        __builtin_amdgcn_s_barrier();                                                                \
        P_KSTAMP(3);                                                                                \
        P_MFMA(0, bh, 1);                                                                            \
        P_KSTAMP(4);                                                                                \
        if (!(CLIPMI_GEMM_ABL & 1)) { P_READ_A(CUR, 1); }                                            \
        if (PRE) P_SLOT(3, KT, NB);                                                                 \
        __builtin_amdgcn_s_barrier();                                                                \
        P_KSTAMP(5);                                                                                \
        P_MFMA(1, bh, 1);                                                                            \
        P_KSTAMP(6);                                                                                \
        if (PRE) P_SLOT(1, KT, NB);                                                                 \
        if ((W3) >= 0) P_WAIT(((W3) < 0 ? 0 : (W3)));                                                \
        __builtin_amdgcn_s_barrier();                                                                \
        P_KSTAMP(7);                                                                                \
        P_MFMA(1, bl, 0);                                                                            \
        P_KSTAMP(8);                                                                                \
    } while (0)

    // ---- prologue: first tile's K-tile 0 in flight, then the bias row of the whole GEMM into LDS
    int m0, n0;
    tile_origin(v, m0, n0);
    set_tile(m0, n0);
    set_out(m0, n0);
    P_ISSUE(0, 0, 0);
    P_ISSUE(2, 0, 0);
    P_ISSUE(3, 0, 0);
    P_ISSUE(1, 0, 0);
    float* sbias = reinterpret_cast<float*>(smem + G256_LDS);
    for (int i = tid; i < g.N; i += 512) sbias[i] = g.bias ? g.bias[i] : 0.f;
    float* sws = sbias + g.N;                          // FP8: per-output-channel weight scales [N];
                                                       // LNC: colsum of the CURRENT tile's 256 columns (by LDS-DMA per tile:
                                                       // a whole-N row beside bias[N] would not fit for N = 4096)
    float* sas = sws + (FP8 ? g.N : LNC ? 256 : 0);    // FP8: the current tile's 256 activation-row scales;
                                                       // LNC: its 256 x nseg (sum, sum of squares) statistics partials
    if (FP8)
        for (int i = tid; i < g.N; i += 512) sws[i] = g.w_scale[i];
    const int nseg_in = K >> 8;                        // LNC: 256-column segments of the rows A came from
    __amdgpu_buffer_rsrc_t rsS, rsC;
    if (FP8) rsS = __builtin_amdgcn_make_buffer_rsrc((void*)g.a_scale, 0, (unsigned)g.M * 4u, 0x00020000);
    if (LNC) {
        rsS = __builtin_amdgcn_make_buffer_rsrc((void*)g.ln_part_in, 0, (unsigned)g.M * (unsigned)nseg_in * 8u, 0x00020000);
        rsC = __builtin_amdgcn_make_buffer_rsrc((void*)g.colsum, 0, (unsigned)g.N * 4u, 0x00020000);
    }
    // (the compiler's wait for these loads also retires the loaders' DMA above; harmless, once per launch.
    //  The tile-start barrier below publishes sbias: every wave reaches it after its ds_write + lgkmcnt(0).)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    char* const buf1 = smem + G256_BUF;
    unsigned short* const outp = static_cast<unsigned short*>(g.out);

    // development stamps (dbg & 4): [wave][tile][stamp] 100 MHz wall-clock ticks in LDS, dumped at the end
    unsigned long long* stamps = reinterpret_cast<unsigned long long*>(smem + G256_LDS + g.N * 4 * (FP8 ? 2 : 1) + (FP8 ? 1024 : LNC ? 1024 + 8200 : 0));
    int tile_i = 0;
    unsigned long long kst[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#define P_STAMP(k)                                                                                   \
    do {                                                                                             \
        if (CLIPMI_GEMM_STAMPS && (g.dbg & 4) && lane == 0 && tile_i < 4) stamps[(wave * 4 + tile_i) * 8 + (k)] = wall_clock64(); \
    } while (0)
    for (;;) {
        P_STAMP(0);
        // ---- tile start: A-lo(0), B-lo(0) of this tile landed (B-hi(0), A-hi(0) may still be in flight)
        P_WAIT(8);
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();     // group 1 runs one slot behind
        const int vn = v + stride;
        const bool has_next = vn < ntiles;
        int nm0 = 0, nn0 = 0;
        if (has_next) tile_origin(vn, nm0, nn0);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[a][i][b][j] = f32x4{0.f, 0.f, 0.f, 0.f};

        P_STAMP(1);
        // One K-tile body for every K-tile of every tile (a single copy keeps the 128 accumulators in one
        // register assignment). K-tile t < nk-1 prefetches K-tile t+1 of this tile; the last one prefetches
        // K-tile 0 of the NEXT tile (of this tile again when there is none: never read, retired before exit),
        // and its P3 wait already retires that tile's A-lo(0), B-lo(0).
        for (int t = 0; t < nk; ++t) {
            const char* cur = smem + (t & 1) * G256_BUF;
            const int nb = (t + 1) & 1;
            int ktn = t + 1;
            const bool last_kt = t == nk - 1;
            if (last_kt) {
                ktn = 0;
                if (has_next) set_tile(nm0, nn0);      // the current offsets are dead: K-tile nk-1 is issued
                if (FP8 && wave == 0) {                // a_scale[m0 .. m0+255] -> sas (rows past M read as 0: never stored);
                    // older than this K-tile's prefetch: retired by its P3 wait
                    unsigned lo_;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo_));
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (__attribute__((address_space(3))) char*)sas, 16,
                                                             (unsigned)m0 * 4u + lo_ * 16u, 0, 0, 0);
                }
                if (LNC && wave == 0) {                // statistics partials of rows m0 .. m0+255 -> sas: 2 nseg 1-KiB pieces
                    // (rows past M read as zeros: never stored); older than this K-tile's prefetch: retired by its waits
                    unsigned lo_;
                    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lo_));
                    const unsigned src0 = (unsigned)m0 * (unsigned)nseg_in * 8u + lo_ * 16u;
                    for (int pc = 0; pc < 2 * nseg_in; ++pc)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsS, (__attribute__((address_space(3))) char*)sas + pc * 1024, 16,
                                                                 src0, (unsigned)pc * 1024u, 0, 0);
                    // colsum[n0 .. n0 + 255] -> sws (one piece)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsC, (__attribute__((address_space(3))) char*)sws, 16,
                                                             (unsigned)n0 * 4u + lo_ * 16u, 0, 0, 0);
                }
            }
            P_KTILE(cur, 1, ktn, nb, 8, 8, 8);
        }
        P_STAMP(2);
        if (CLIPMI_GEMM_STAMPS && (g.dbg & 8) && tile_i == 1 && lane == 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) stamps[128 + wave * 9 + i] = kst[i];     // after the 4 x 8 x 4 tile stamps
        }
        if constexpr (LNC) {
            // The tile's statistics partials (256 rows x nseg (sum, sumsq) pairs, DMA'd by wave 0 under the last K-tile and
            // retired by its own counted waits) are folded IN PLACE into 256 (mean, rstd) pairs by wave 0 alone, four rows
            // per lane in two halves (rows l, l + 64, then l + 128, l + 192: a half's writes land on partials that half has
            // already read), BEFORE the barrier that ends the K-loop: wave group 1 is still in its last MFMA slot then, so
            // the fold costs no extra barrier and next to no time (a 256-thread fold behind two extra barriers cost c_fc
            // 3 us per launch; per-lane folding of the lane's own 8 rows cost 16-32 registers beside the 128 accumulators
            // and pushed the storers' store addresses into scratch). asm accesses throughout (a compiler-visible access
            // to an LDS-DMA destination waits vmcnt(0)); always 4 pairs are read: for nseg < 4 the extra ones belong to the
            // next row and are ignored.
            if (wave == 0) {
                const unsigned sas0_ = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sas;
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    f32x2 p0, p1, p2, p3, p4, p5, p6, p7;
                    const unsigned r0_ = (unsigned)(hf * 128 + lane), r1_ = r0_ + 64u;
                    asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:8\n\tds_read_b64 %2, %8 offset:16\n\t"
                                 "ds_read_b64 %3, %8 offset:24\n\tds_read_b64 %4, %9\n\tds_read_b64 %5, %9 offset:8\n\t"
                                 "ds_read_b64 %6, %9 offset:16\n\tds_read_b64 %7, %9 offset:24\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3), "=&v"(p4), "=&v"(p5), "=&v"(p6), "=&v"(p7)
                                 : "v"(sas0_ + r0_ * (unsigned)nseg_in * 8u), "v"(sas0_ + r1_ * (unsigned)nseg_in * 8u)
                                 : "memory");
                    const float pa[8] = {p0.x, p0.y, p1.x, p1.y, p2.x, p2.y, p3.x, p3.y};
                    const float pb[8] = {p4.x, p4.y, p5.x, p5.y, p6.x, p6.y, p7.x, p7.y};
                    const f32x2 ma = ln_row_stats(pa, nseg_in, K), mb = ln_row_stats(pb, nseg_in, K);
                    asm volatile("ds_write_b64 %0, %2\n\tds_write_b64 %1, %3\n\ts_waitcnt lgkmcnt(0)"
                                 ::"v"(sas0_ + r0_ * 8u), "v"(sas0_ + r1_ * 8u), "v"(ma), "v"(mb) : "memory");
                }
            }
        }
        if (wm == 0) __builtin_amdgcn_s_barrier();     // balance the stagger; all reads of buffer 1 are done

        // ---- epilogue through buffer 1: two passes of 128 rows x 512 B
        f32x2 lst[2][4];         // LNC: (mean, rstd) of the lane's 8 accumulator rows
        if constexpr (LNC) {
            const unsigned sas_ = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sas;
            const unsigned sa_ = sas_ + (wm * 64 + fr) * 8;
            asm volatile(
                "ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:128\n\tds_read_b64 %2, %8 offset:256\n\tds_read_b64 %3, %8 offset:384\n\t"
                "ds_read_b64 %4, %8 offset:1024\n\tds_read_b64 %5, %8 offset:1152\n\tds_read_b64 %6, %8 offset:1280\n\t"
                "ds_read_b64 %7, %8 offset:1408\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(lst[0][0]), "=&v"(lst[0][1]), "=&v"(lst[0][2]), "=&v"(lst[0][3]), "=&v"(lst[1][0]), "=&v"(lst[1][1]),
                  "=&v"(lst[1][2]), "=&v"(lst[1][3])
                : "v"(sa_)
                : "memory");
        }
        f32x4 bz[2][2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                bz[b][nt] = *reinterpret_cast<const f32x4*>(sbias + n0 + b * 128 + wn * 32 + nt * 16 + 4 * fg);
        // FP8: sc[a][mt][b][nt] multiplies the accumulator: w_scale of the lane's 4 columns x a_scale of its row
        f32x4 wsv[2][2];
        float asv[2][4];
        if constexpr (FP8) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    wsv[b][nt] = *reinterpret_cast<const f32x4*>(sws + n0 + b * 128 + wn * 32 + nt * 16 + 4 * fg);
            // asm reads: a compiler-visible read of an LDS-DMA destination is preceded by vmcnt(0)
            const unsigned sa_ = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sas + (wm * 64 + fr) * 4;
            asm volatile(
                "ds_read_b32 %0, %8\n\tds_read_b32 %1, %8 offset:64\n\tds_read_b32 %2, %8 offset:128\n\tds_read_b32 %3, %8 offset:192\n\t"
                "ds_read_b32 %4, %8 offset:512\n\tds_read_b32 %5, %8 offset:576\n\tds_read_b32 %6, %8 offset:640\n\t"
                "ds_read_b32 %7, %8 offset:704\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(asv[0][0]), "=&v"(asv[0][1]), "=&v"(asv[0][2]), "=&v"(asv[0][3]), "=&v"(asv[1][0]), "=&v"(asv[1][1]),
                  "=&v"(asv[1][2]), "=&v"(asv[1][3])
                : "v"(sa_)
                : "memory");
        }
        if constexpr (LNC) {
            // the tile's colsum values of this lane's columns (asm reads: sws is an LDS-DMA destination)
            const unsigned ca_ = (unsigned)(size_t)(__attribute__((address_space(3))) char*)sws + (unsigned)(wn * 32 + 4 * fg) * 4u;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:64\n\tds_read_b128 %2, %4 offset:512\n\t"
                         "ds_read_b128 %3, %4 offset:576\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(wsv[0][0]), "=&v"(wsv[0][1]), "=&v"(wsv[1][0]), "=&v"(wsv[1][1]) : "v"(ca_) : "memory");
        }
        // bias (and the FP8 scales / the LN-folded correction) are folded into the accumulators in place, before the
        // staging passes: the per-column / per-row factors are dead by the time the passes need registers for their LDS traffic
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        if constexpr (FP8) acc[a][mt][b][nt] = acc[a][mt][b][nt] * (wsv[b][nt] * asv[a][mt]) + bz[b][nt];
                        else if constexpr (LNC) acc[a][mt][b][nt] = ln_apply(acc[a][mt][b][nt], lst[a][mt].x, lst[a][mt].y, wsv[b][nt], bz[b][nt]);
                        else acc[a][mt][b][nt] = acc[a][mt][b][nt] + bz[b][nt];
                    }
#define P_SCALED(accv, a_, mt_, b_, nt_) (accv)
        if constexpr (RESID) {
            // Residual stream update out[m][n] = (acc + bias) + out[m][n] (the add order of gemm256: bit-identical).
            // Eight sub-passes sp of 32 tile rows ((sp>>2)*128 + (sp&1)*64 + ((sp>>1)&1)*32 ..+31: the rows the
            // wave group wm = sp&1 holds for A half sp>>2, mt in {2((sp>>1)&1), +1}):
            //   * that group drops acc + bias into a 32-KiB image at [96K,128K) (row = (mt&1)*16 + fr, 16-B chunk ^
            //     (row & 15));
            //   * the residual rows arrive by LDS-DMA, lane-linear, in one of THREE 32-KiB buffers used in turn
            //     X = regions {A-lo, B-lo}, Y = {A-hi, B-hi}, Z = [64K,96K): sub-passes 0, 1 were issued under the
            //     last K-tile, 2 at the start of the epilogue, sp+3 when sp has been written out - three sub-passes
            //     in flight cover the HBM round trip (two did not: 13 us per tile, r01 stamps);
            //   * X and Y, freed by sp = 6 and 7, receive the next tile's K-tile 0 in the K-loop's own order (A-lo,
            //     B-lo, then B-hi, A-hi), so the next K-loop's counted waits hold unchanged.
            // Loaders: DMA + counted waits only; storers: LDS reads, the add and the global stores (fire-and-forget:
            // they drain under the next tile's K-loop).
            const unsigned img_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem + 6 * G256_HALF;
            const unsigned resid_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
            unsigned stage_addr[2];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                stage_addr[nt] = img_base + fr * 1024 + (wn >> 1) * 256 + (((((wn & 1) * 8 + nt * 4 + fg) ^ fr)) << 4);
            float* const outf = static_cast<float*>(g.out);
            const int nseg = g.N >> 8, seg = n0 >> 8;
            P_ISSUE_RESID(4, 32);                          // sub-pass 2 (tile rows 32..63) -> Z
            P_ISSUE_RESID(5, 48);
#pragma unroll
            for (int sp = 0; sp < 8; ++sp) {
                const int h = sp & 1;
                const int lo = sp % 3 == 0 ? 0 : (sp % 3 == 1 ? 1 : 4), hi = sp % 3 == 0 ? 2 : (sp % 3 == 1 ? 3 : 5);
                if (wm == h) {
#define P_STAGE32(mt, b, nt)                                                                         \
    do {                                                                                             \
        const f32x4 x_ = P_SCALED(acc[sp >> 2][((sp >> 1) & 1) * 2 + (mt)][b][nt], sp >> 2, ((sp >> 1) & 1) * 2 + (mt), b, nt); \
        asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(stage_addr[nt]), "v"(x_), "n"((mt) * 16384 + (b) * 512) : "memory"); \
    } while (0)
#define P_STAGE32_MT(mt) P_STAGE32(mt, 0, 0); P_STAGE32(mt, 0, 1); P_STAGE32(mt, 1, 0); P_STAGE32(mt, 1, 1)
                    P_STAGE32_MT(0); P_STAGE32_MT(1);
#undef P_STAGE32_MT
#undef P_STAGE32
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                // residual sub-pass sp has landed; younger in flight: sp+1, sp+2 (sp <= 5), then sp+1 or the
                // first half of the next K-tile 0
                if (sp <= 5) P_WAIT(16); else P_WAIT(8);
                __builtin_amdgcn_s_barrier();
                if (!loader) {
                    // storer wn: rows r = 8 wn + i of the sub-pass (image row r; residual row r in region
                    // (r < 16 ? lo : hi) at (r & 15) KiB, lane-linear)
                    const unsigned ra = resid_base + (wn < 2 ? lo : hi) * G256_HALF + ((wn & 1) * 8) * 1024 + lane * 16;
                    unsigned aa[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        aa[i] = img_base + (wn * 8 + i) * 1024 + ((lane ^ ((wn & 1) * 8 + i)) << 4);
                    const int lr0 = (sp >> 2) * 128 + (sp & 1) * 64 + ((sp >> 1) & 1) * 32 + wn * 8;
                    float sa[8], sq[8];          // RLN: per-lane (sum, sum of squares) of the 8 new rows
                    if constexpr (!RLN) {
                        f32x4 x0, x1, x2, x3, x4, x5, x6, x7, r0, r1, r2, r3, r4, r5, r6, r7;
                        // one statement: 16 reads in flight and their wait (asm destinations are unprotected until it)
                        asm volatile(
                            "ds_read_b128 %0, %16\n\tds_read_b128 %1, %17\n\tds_read_b128 %2, %18\n\tds_read_b128 %3, %19\n\t"
                            "ds_read_b128 %4, %20\n\tds_read_b128 %5, %21\n\tds_read_b128 %6, %22\n\tds_read_b128 %7, %23\n\t"
                            "ds_read_b128 %8, %24\n\tds_read_b128 %9, %24 offset:1024\n\tds_read_b128 %10, %24 offset:2048\n\t"
                            "ds_read_b128 %11, %24 offset:3072\n\tds_read_b128 %12, %24 offset:4096\n\tds_read_b128 %13, %24 offset:5120\n\t"
                            "ds_read_b128 %14, %24 offset:6144\n\tds_read_b128 %15, %24 offset:7168\n\ts_waitcnt lgkmcnt(0)"
                            : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7),
                              "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7)
                            : "v"(aa[0]), "v"(aa[1]), "v"(aa[2]), "v"(aa[3]), "v"(aa[4]), "v"(aa[5]), "v"(aa[6]), "v"(aa[7]), "v"(ra)
                            : "memory");
                        const f32x4 xs[8] = {x0, x1, x2, x3, x4, x5, x6, x7};
                        const f32x4 rs[8] = {r0, r1, r2, r3, r4, r5, r6, r7};
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            const int m = m0 + lr0 + i;
                            if (m < g.M && !(g.dbg & 1))
                                *reinterpret_cast<f32x4*>(outf + (size_t)m * g.N + n0 + lane * 4) = xs[i] + rs[i];
                        }
                    } else {
                        // split residual (gemm.hpp): old rows from the DMA'd pieces ([256 bf16 hi | 256 u8 lo] per row),
                        // new rows o = (acc + bias) + old -> (hi', lo') by split_make, and this tile's share of the row
                        // statistics (canonical order). Two groups of four rows (12 reads in flight each): fewer live
                        // row registers than the plain form, which keeps this variant out of scratch (a reload costs a
                        // storer a vmcnt(0) = all its stores).
                        const unsigned ra2 = ra - lane * 8;                      // hi: + lane * 8 instead of lane * 16
                        const unsigned ra3 = ra - lane * 12 + 512;               // lo: + 512 + lane * 4
                        char* const x3 = static_cast<char*>(g.x3);
                        const unsigned row_pitch = 3u * (unsigned)g.N;
                        // where this lane's ONE 16-byte store of a row goes (bytes from the row's start): lanes 4k, 4k + 2 write
                        // hi of 8 columns each, lane 4k + 1 lo of the quad's 16 columns, lane 4k + 3 nothing
                        const unsigned st_off = (lane & 1) ? 2u * (unsigned)g.N + (unsigned)n0 + (unsigned)(lane & ~3) * 4u
                                                           : ((unsigned)n0 + (unsigned)(lane & ~1) * 4u) * 2u;
#pragma unroll
                        for (int hgrp = 0; hgrp < 2; ++hgrp) {
                            f32x4 x0, x1, x2, x3_;
                            uint2 h0, h1, h2, h3;
                            unsigned l0, l1, l2, l3;
                            const unsigned rb = ra2 + hgrp * 4096, rl = ra3 + hgrp * 4096;
                            asm volatile(
                                "ds_read_b128 %0, %12\n\tds_read_b128 %1, %13\n\tds_read_b128 %2, %14\n\tds_read_b128 %3, %15\n\t"
                                "ds_read_b64 %4, %16\n\tds_read_b64 %5, %16 offset:1024\n\tds_read_b64 %6, %16 offset:2048\n\t"
                                "ds_read_b64 %7, %16 offset:3072\n\t"
                                "ds_read_b32 %8, %17\n\tds_read_b32 %9, %17 offset:1024\n\tds_read_b32 %10, %17 offset:2048\n\t"
                                "ds_read_b32 %11, %17 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                                : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3_), "=&v"(h0), "=&v"(h1), "=&v"(h2), "=&v"(h3),
                                  "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
                                : "v"(aa[4 * hgrp]), "v"(aa[4 * hgrp + 1]), "v"(aa[4 * hgrp + 2]), "v"(aa[4 * hgrp + 3]), "v"(rb), "v"(rl)
                                : "memory");
                            const f32x4 xs[4] = {x0, x1, x2, x3_};
                            const uint2 hs[4] = {h0, h1, h2, h3};
                            const unsigned ls[4] = {l0, l1, l2, l3};
#pragma unroll
                            for (int i4 = 0; i4 < 4; ++i4) {
                                const int i = 4 * hgrp + i4;
                                const f32x4 o = xs[i4] + split_join(hs[i4], ls[i4]);
                                const int m = m0 + lr0 + i;
                                uint2 nh;
                                unsigned nl;
                                split_make(o, nh, nl);
                                // lanes l and l ^ 1 trade hi halves, lane 4k + 1 gathers the quad's four lo words (DPP quad_perm
                                // moves, no LDS traffic): ONE 16-byte store per storing lane and row
                                uint2 got;
                                got.x = (unsigned)__builtin_amdgcn_update_dpp(0, (int)nh.x, 0xB1, 0xf, 0xf, false);   // quad_perm [1,0,3,2]
                                got.y = (unsigned)__builtin_amdgcn_update_dpp(0, (int)nh.y, 0xB1, 0xf, 0xf, false);
                                const unsigned q0 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)nl, 0x00, 0xf, 0xf, false);   // [0,0,0,0]
                                const unsigned q2 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)nl, 0xAA, 0xf, 0xf, false);   // [2,2,2,2]
                                const unsigned q3 = (unsigned)__builtin_amdgcn_update_dpp(0, (int)nl, 0xFF, 0xf, 0xf, false);   // [3,3,3,3]
                                if (m < g.M && (lane & 3) != 3 && !(g.dbg & 1))
                                    gp_store16<2>(x3 + (size_t)m * row_pitch + st_off,
                                                  (lane & 1) ? make_uint4(q0, nl, q2, q3) : make_uint4(nh.x, nh.y, got.x, got.y));
                                sa[i] = ln_lane_sum(o);
                                sq[i] = ln_lane_sumsq(o);
                            }
                        }
                        // 8 rows x (sum, sum of squares) reduced TRANSPOSED over xor 32, 16, 8 (each lane keeps half of
                        // what it held), then over 4, 2, 1: the same adds as ln_wave_sum, by half-wave / row swaps and DPP
                        // moves (gemm.hpp lane_fold32 ..: no LDS round trips; rounds 2-4 ran 20 ds_bpermute in six dependent steps)
                        const bool h8 = lane & 8, bit2 = lane & 4;
                        float a4[4], q4[4], a2[2], q2[2];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            a4[j] = lane_fold32(sa[j], sa[j + 4]);
                            q4[j] = lane_fold32(sq[j], sq[j + 4]);
                        }
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            a2[j] = lane_fold16(a4[j], a4[j + 2]);
                            q2[j] = lane_fold16(q4[j], q4[j + 2]);
                        }
                        float a1 = (h8 ? a2[1] : a2[0]) + lane_xor8(h8 ? a2[0] : a2[1]);
                        float q1 = (h8 ? q2[1] : q2[0]) + lane_xor8(h8 ? q2[0] : q2[1]);
                        a1 += lane_xor4(a1, bit2); q1 += lane_xor4(q1, bit2);
                        a1 += lane_xor2(a1); q1 += lane_xor2(q1);
                        a1 += lane_xor1(a1); q1 += lane_xor1(q1);
                        const int mrow = m0 + lr0 + (lane >> 3);          // the row this lane group ended up with
                        if ((lane & 7) == 0 && mrow < g.M && !(g.dbg & 1))
                            *reinterpret_cast<f32x2*>(g.ln_part + ((size_t)mrow * nseg + seg) * 2) = f32x2{a1, q1};
                    }
                }
                __builtin_amdgcn_s_barrier();              // sub-pass sp is out of LDS: image and buffer may be refilled
                if (sp & 1) P_STAMP(3 + (sp >> 1));
                if (sp < 5) {
                    const int s3 = sp + 3;
                    const int rb = (s3 >> 2) * 128 + (s3 & 1) * 64 + ((s3 >> 1) & 1) * 32;
                    P_ISSUE_RESID(lo, rb);
                    P_ISSUE_RESID(hi, rb + 16);
                } else if (sp == 6) {
                    P_ISSUE(0, 0, 0);                      // next tile's K-tile 0 (this tile's again when there is none)
                    P_ISSUE(2, 0, 0);
                } else if (sp == 7) {
                    P_ISSUE(3, 0, 0);
                    P_ISSUE(1, 0, 0);
                }
            }
        } else {
            // Staging writes go through asm ds_write: a compiler-visible LDS store would be preceded by
            // vmcnt(0) (the compiler cannot prove it does not overlap the LDS-DMA in flight into buffer 0).
            const unsigned stage_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)buf1;
            unsigned stage_addr[2];
    #pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                stage_addr[nt] = stage_base + (wm * 64 + fr) * 512 +
                                 ((((wn * 4 + nt * 2 + (fg >> 1)) ^ fr) << 4) | ((fg & 1) * 8));
    #pragma unroll
            for (int a = 0; a < 2; ++a) {
                if (g.dbg & 2) { asm volatile("" :: "v"(acc[a][0][0][0]), "v"(acc[a][3][1][1])); continue; }
                if (a) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // storers: pass-0 rows are in registers
                    __builtin_amdgcn_s_barrier();
                }
                // row = wm*64 + mt*16 + fr (row & 15 == fr); 16-B chunk index = b*16 + ((wn*4 + nt*2 + fg/2) ^ fr):
                // two lane addresses (nt = 0, 1) + immediates mt*8192 + b*256
    #define P_STAGE(mt, b, nt)                                                                           \
        do {                                                                                             \
            f32x4 x_ = P_SCALED(acc[a][mt][b][nt], a, mt, b, nt);                                        \
            if (epi_is_qgelu(EPI)) x_ = quick_gelu4(x_);                                                  \
            const uint2 pk_ = make_uint2(pack_bf16x2(x_.x, x_.y), pack_bf16x2(x_.z, x_.w));              \
            asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(stage_addr[nt]), "v"(pk_), "n"((mt) * 8192 + (b) * 256) : "memory"); \
        } while (0)
    #define P_STAGE_MT(mt) P_STAGE(mt, 0, 0); P_STAGE(mt, 0, 1); P_STAGE(mt, 1, 0); P_STAGE(mt, 1, 1)
                P_STAGE_MT(0); P_STAGE_MT(1); P_STAGE_MT(2); P_STAGE_MT(3);
    #undef P_STAGE_MT
    #undef P_STAGE
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                P_STAMP(3 + 2 * a);
                if (!loader) {
                    // Storer wave wn writes out rows 16 k + (4 wn + 2 s + hi), k < 8, s < 2, hi = lane / 32: two
                    // rows x 512 B per instruction, and row & 15 does not depend on k, so the swizzled read address
                    // is one lane address per s + the immediate 8192 k. asm reads (8 in flight, one wait): a
                    // compiler-visible ds_read here is preceded by vmcnt(0) (LDS-DMA in flight), which on a storer
                    // means "wait for every store issued so far".
    #pragma unroll
                    for (int sidx = 0; sidx < 2; ++sidx) {
                        const int rlow = wn * 4 + sidx * 2 + (lane >> 5);
                        const unsigned ra = stage_base + rlow * 512 + (((lane & 31) ^ rlow) << 4);
                        uint4 x0, x1, x2, x3, x4, x5, x6, x7;
                        asm volatile(
                            "ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:8192\n\tds_read_b128 %2, %8 offset:16384\n\t"
                            "ds_read_b128 %3, %8 offset:24576\n\tds_read_b128 %4, %8 offset:32768\n\tds_read_b128 %5, %8 offset:40960\n\t"
                            "ds_read_b128 %6, %8 offset:49152\n\tds_read_b128 %7, %8 offset:57344\n\ts_waitcnt lgkmcnt(0)"
                            : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(x4), "=&v"(x5), "=&v"(x6), "=&v"(x7)
                            : "v"(ra)
                            : "memory");
                        const uint4 xs[8] = {x0, x1, x2, x3, x4, x5, x6, x7};
                        if constexpr (FP8 && EPI == EPI_BIAS_QGELU_BF16) {
                            if (g.out_bscale) {
                                // the rows leave as e4m3 with MX block scales (the next GEMM's A operand: gemm256f8.hpp BSA)
                                // instead of bf16: a lane holds 8 consecutive columns, the 4 lanes of a quad one 32-block -
                                // the same bytes quantize_rows_fp8mx_kernel makes of the bf16 rows (vit_kernels.hpp)
                                unsigned char* const out8 = static_cast<unsigned char*>(g.out);
    #pragma unroll
                                for (int k = 0; k < 8; ++k) {
                                    unsigned sb_;
                                    const uint2 q8_ = mx_pack_bf16x8(xs[k], sb_);          // gemm.hpp: shared by every producer
                                    const int m = m0 + a * 128 + k * 16 + rlow;
                                    if (m < g.M) {
                                        const int col = n0 + (lane & 31) * 8;
                                        *reinterpret_cast<uint2*>(out8 + (size_t)m * g.N + col) = q8_;
                                        if ((lane & 3) == 0) g.out_bscale[(size_t)m * (g.N >> 5) + (col >> 5)] = (unsigned char)sb_;
                                    }
                                }
                                continue;
                            }
                        }
    #pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const int m = m0 + a * 128 + k * 16 + rlow;
                            if (m < g.M && !(g.dbg & 1))
                                gp_store16<1>(outp + (size_t)m * g.N + n0 + (lane & 31) * 8, xs[k]);
                        }
                    }
                }
                P_STAMP(4 + 2 * a);
            }

        }
        P_STAMP(7);
        ++tile_i;
        if (!has_next) break;
        v = vn;
        m0 = nm0;
        n0 = nn0;
        set_out(m0, n0);
        // buffer 1 is refilled (K-tile 1 of the next tile) only after the tile-start barrier, which the
        // storers reach after their last staging read has returned
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    P_WAIT(0);                                         // the last tile's idle prefetch must not outlive the LDS
    if (CLIPMI_GEMM_STAMPS && (g.dbg & 12) && g.pos && (blockIdx.x == 0 || blockIdx.x == 100)) {
        __syncthreads();
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(const_cast<float*>(g.pos)) + (blockIdx.x ? 256 : 0);
        if (tid < 256) dst[tid] = stamps[tid];
    }
#undef P_SCALED
#undef P_STAMP
#undef P_ISSUE
#undef P_WAIT
#undef P_READ_A
#undef P_READ_B
#undef P_MFMA
#undef P_KTILE
}

}  // namespace clipmi
