// topk.hip — exact flat inner-product top-K over a packed [N][E] f32 matrix (gfx950).
//
// Replaces `index.search(features, k + offset + 1)` (reference query-index.py:111) with an exact
// streaming pass. HBM-bound: the matrix is read once per batch of <= 16
// queries; scores are produced by v_mfma_f32_16x16x4_f32 (exact f32, one fmaf chain per
// (row, query) in a FIXED order, restated in oracle/topk_oracle.c) and selected per wavefront
// with a threshold filter + bounded candidate buffer in LDS.
//
// Score order ("score order" in DESIGN.md): for row r, query q, E = 16*NT:
//     acc = 0; for t in [0,NT) for c in [0,4) for g in [0,4): k = 16t + 4g + c;
//         acc = fmaf(db[r][k], q[k], acc)
// (t,c) is one MFMA instruction; g is the MFMA's internal k index (lane >> 4).
//
// Ordering rule everywhere (wave select, block merge, cross-rank merge, oracle):
//     a beats b  <=>  a.score > b.score || (a.score == b.score && a.id < b.id); NaN never selected.
#include "common.hpp"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <type_traits>
#include <hip/hip_ext.h>

namespace clipmi {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Cache policy of the once-read streams (CLIPMI_NT_MASK, compile time): bit 0 the coarse copies of the 64-query
// passes, bit 1 the first wide form's rows (65-256 queries), bit 4 the second form's, bit 2 the re-scored f32 rows, bit 3 the exact
// f32 scan. A set bit = the non-temporal hint
// (`nt`): the 64-query int8 scan streams 5.2 GB per call and reads 6.15 -> 6.62 TB/s with it; the first wide form (one workgroup
// per block) gains 1-5 %; the others lose or do not move (bits 2, 3, 4 stay clear).
#ifndef CLIPMI_NT_MASK
#define CLIPMI_NT_MASK 3
#endif
template <int BIT>
__device__ __forceinline__ f32x4 ld16_f(const float* p) {
    if constexpr ((CLIPMI_NT_MASK >> BIT) & 1) return __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    else return *reinterpret_cast<const f32x4*>(p);
}

constexpr int SAMPLE_MIN_N = 65536;     // below this the pre-pass is not worth its launches
constexpr int LDS_LIMIT = 160 * 1024 - 512;

__device__ __forceinline__ bool beats(float sa, unsigned ia, float sb, unsigned ib) {
    return sa > sb || (sa == sb && ia < ib);
}

// LDS traffic of ONE wave is processed in issue order; this only has to stop the compiler from
// moving LDS accesses across the point and to drain the wave's outstanding LDS operations.
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// Keep the best min(n, K) of the n candidates in buf[0..n) (n = *cnt_p, wave-uniform), sorted by
// the ordering rule, using rank counting: rank(e) = #{j : cand_j beats cand_e}. Ids are unique,
// so ranks are a permutation. All 64 lanes of the wave must call this together.
__device__ void wave_compact(uint2* buf, uint2* scratch, int* cnt_p, float* thr_p, int K, int lane) {
    const int n = *cnt_p;
    for (int e = lane; e < n; e += 64) {
        const uint2 me = buf[e];
        const float ms = __uint_as_float(me.x);
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const uint2 o = buf[j];
            rank += beats(__uint_as_float(o.x), o.y, ms, me.y) ? 1 : 0;
        }
        if (rank < K) scratch[rank] = me;
    }
    wave_lds_sync();
    const int m = n < K ? n : K;
    for (int e = lane; e < m; e += 64) buf[e] = scratch[e];
    if (lane == 0) {
        *cnt_p = m;
        *thr_p = (n >= K) ? __uint_as_float(scratch[K - 1].x) : -INFINITY;
    }
    wave_lds_sync();
}

struct ScanArgs {
    const float* db;        // [nrows][E]
    long long nrows;        // >= 1
    const float* q;         // first query of this group, [QA][E]
    int QA;                 // active queries in this pass, 1..16*QG
    int K;
    int C;                  // candidate capacity per (wave, query): multiple of 16, >= K + 16
    int wave_bytes;         // LDS bytes per wave
    const float* thr_in;    // [QA] initial thresholds (valid lower bounds) or nullptr
    uint2* cand;            // [QA][cap] dense candidate lists (unsorted), cap >= NL * C
    unsigned* gcnt;         // [32] entries used per query; zeroed on the stream before the launch
    long long cap;
    const unsigned* run_if; // optional device flag: the kernel exits at once when *run_if == 0
    int q_total;            // gridDim.y > 1 only: queries over all groups; group y takes queries [y*QA, y*QA + QA) and
                            // the matching slices of q, thr_in, cand, gcnt
};

// PREPASS only changes the kernel's NAME (profilers average per name; the threshold pre-pass over
// a few thousand rows must not dilute the main scan's average duration).
// QG = 16-query column groups per pass (1 or 2): every DB fragment feeds QG independent MFMA
// accumulator chains. The pass stays HBM-bound up to QG = 2 (f32 MFMA: 2 x 128 instructions of 32
// cycles per 32 KiB tile per SIMD = 16 B/clk/CU against ~10 B/clk/CU of HBM), and two chains hide
// the 40-cycle dependent-accumulator latency of a single one.
template <int E, bool PREPASS, int QG>
__global__ void __launch_bounds__(256) scan_topk_f32_kernel(ScanArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = E / 16;      // float4 per lane per 16-row tile
    constexpr int NCH = NT / 8;     // chunks of 8 float4
    if (a.run_if && *a.run_if == 0) return;       // fallback launch that is not needed (uniform)
    if (gridDim.y > 1) {                          // the coarse path's fallback: all query groups in one launch
        const int y = blockIdx.y, left = a.q_total - y * a.QA;
        a.q += (size_t)y * a.QA * E;
        a.cand += (size_t)y * a.QA * a.cap;
        a.gcnt += y * a.QA;
        if (a.thr_in) a.thr_in += y * a.QA;
        a.QA = left < a.QA ? left : a.QA;
    }

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int col = lane & 15;      // query column of this lane inside a group (B operand / accumulator column)
    const int g = lane >> 4;        // MFMA k index supplied by this lane; accumulator row group

    // query image, one 1-KiB lane-linear piece per (group, t): entry [(qg*NT + t)*64 + lane] =
    // q[16*qg + col][16t+4g .. +3]
    f32x4* qimg = reinterpret_cast<f32x4*>(smem);
    for (int idx = tid; idx < QG * NT * 64; idx += blockDim.x) {
        const int l = idx & 63, t = (idx >> 6) % NT, qg = (idx >> 6) / NT;
        const int c_ = qg * 16 + (l & 15), g_ = l >> 4;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (c_ < a.QA) v = *reinterpret_cast<const f32x4*>(a.q + (size_t)c_ * E + 16 * t + 4 * g_);
        qimg[idx] = v;
    }
    char* wbase = smem + QG * NT * 1024 + (size_t)wave * a.wave_bytes;
    uint2* buf = reinterpret_cast<uint2*>(wbase);            // [QA][C]
    uint2* scratch = buf + (size_t)a.QA * a.C;                // [C]
    int* cnt = reinterpret_cast<int*>(scratch + a.C);         // [32]
    float* thr = reinterpret_cast<float*>(cnt + 32);          // [32]
    if (lane < 32) {
        cnt[lane] = 0;
        thr[lane] = (lane < a.QA) ? (a.thr_in ? a.thr_in[lane] : -INFINITY) : INFINITY;
    }
    __syncthreads();

    bool active[QG];
    float tau[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        active[qg] = qg * 16 + col < a.QA;
        tau[qg] = thr[qg * 16 + col];
    }
    const int C = a.C, K = a.K;

    const long long ntiles = (a.nrows + 15) >> 4;
    const long long wg = (long long)blockIdx.x * nwaves + wave;
    const long long tw = (long long)gridDim.x * nwaves;
    const long long last_row = a.nrows - 1;

    long long tile = wg;
    if (tile < ntiles) {
        f32x4 T[NT];
        {
            long long r = tile * 16 + col;
            r = r > last_row ? last_row : r;
            const float* p = a.db + r * E + 4 * g;
#pragma unroll
            for (int t = 0; t < NT; ++t) T[t] = ld16_f<3>(p + 16 * t);
        }
        while (true) {
            const long long nxt = tile + tw;
            const bool has_next = nxt < ntiles;
            long long rn = (has_next ? nxt : tile) * 16 + col;
            rn = rn > last_row ? last_row : rn;
            const float* pn = a.db + rn * E + 4 * g;

            f32x4 acc[QG];
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) acc[qg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = 8 * c + j;
                    f32x4 bq[QG];
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg) bq[qg] = qimg[(qg * NT + t) * 64 + lane];
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg) acc[qg] = __builtin_amdgcn_mfma_f32_16x16x4f32(T[t].x, bq[qg].x, acc[qg], 0, 0, 0);
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg) acc[qg] = __builtin_amdgcn_mfma_f32_16x16x4f32(T[t].y, bq[qg].y, acc[qg], 0, 0, 0);
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg) acc[qg] = __builtin_amdgcn_mfma_f32_16x16x4f32(T[t].z, bq[qg].z, acc[qg], 0, 0, 0);
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg) acc[qg] = __builtin_amdgcn_mfma_f32_16x16x4f32(T[t].w, bq[qg].w, acc[qg], 0, 0, 0);
                }
                // this chunk's registers are free: refill them with the next tile (stays in
                // flight across the candidate step below and the first chunks of the next pass).
                // The sched_barriers pin the refill HERE: left alone, hipcc sinks all refills
                // behind the tile's last MFMA and the wave then waits out the full HBM latency.
                // The empty asm consumes the chunk's last MFMA results and clobbers memory, so no
                // refill load can be hoisted above an MFMA that still reads the old registers
                // (which would cost a register copy behind a vmcnt(0) at the loop head).
#pragma unroll
                for (int qg = 0; qg < QG; ++qg) asm volatile("" : "+a"(acc[qg]) : : "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int t = 8 * c + j;
                    T[t] = ld16_f<3>(pn + 16 * t);
                }
                __builtin_amdgcn_sched_barrier(0);
            }

            // accumulator qg: column = query 16*qg + col, rows = tile*16 + 4g + r
            const long long row0 = tile * 16 + 4 * g;
            bool any = false;
            bool pp[QG][4];
#pragma unroll
            for (int qg = 0; qg < QG; ++qg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pp[qg][r] = active[qg] && (row0 + r <= last_row) && (acc[qg][r] >= tau[qg]);
                    any |= pp[qg][r];
                }
            if (__ballot(any)) {
#pragma unroll
                for (int qg = 0; qg < QG; ++qg) {
                    const int qi = qg * 16 + col;
                    uint2* qb = buf + (size_t)qi * C;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (pp[qg][r]) {
                            const int pos = atomicAdd(&cnt[qi], 1);
                            qb[pos] = make_uint2(__float_as_uint(acc[qg][r]), (unsigned)(row0 + r));
                        }
                }
                wave_lds_sync();
#pragma unroll
                for (int qg = 0; qg < QG; ++qg) {
                    const bool need = active[qg] && (g == 0) && (cnt[qg * 16 + col] > C - 16);
                    unsigned long long mask = __ballot(need);
                    if (mask) {
                        while (mask) {
                            const int qq = qg * 16 + __builtin_ctzll(mask);
                            mask &= mask - 1;
                            wave_compact(buf + (size_t)qq * C, scratch, &cnt[qq], &thr[qq], K, lane);
                        }
                        tau[qg] = thr[qg * 16 + col];
                    }
                }
            }
            if (!has_next) break;
            tile = nxt;
        }
    }

    // publish the block's surviving candidates (unsorted, possibly more than K per wave): ONE
    // returning atomic per (block, query), issued by up to 32 lanes at once — per-wave serial atomics
    // here cost ~200 us of same-address contention at the end of a 1M-row scan
    __syncthreads();
    unsigned* gbase = reinterpret_cast<unsigned*>(smem);            // query image is dead now
    auto wave_cnt = [&](int w) {
        return reinterpret_cast<int*>(smem + QG * NT * 1024 + (size_t)w * a.wave_bytes + (size_t)(a.QA + 1) * a.C * 8);
    };
    if (wave == 0 && lane < a.QA) {
        unsigned total = 0;
        for (int w = 0; w < nwaves; ++w) total += (unsigned)wave_cnt(w)[lane];
        gbase[lane] = total ? atomicAdd(&a.gcnt[lane], total) : 0u;
    }
    __syncthreads();
    for (int qq = 0; qq < a.QA; ++qq) {
        const int n = cnt[qq];
        if (n == 0) continue;
        unsigned base = gbase[qq];
        for (int w = 0; w < wave; ++w) base += (unsigned)wave_cnt(w)[qq];
        uint2* dst = a.cand + (size_t)qq * a.cap + base;
        for (int e = lane; e < n; e += 64) dst[e] = buf[(size_t)qq * C + e];
    }
}

// Level 1 of the coarse path's pre-pass: exact scores of the first `nrows` rows (a few thousand) for up to 64 queries
// in ONE launch, written out unfiltered as candidate entries cand[q][row] = (score bits, row) for the radix select to
// take the K-th best of. Same MFMA sequence per (row, query) as scan_topk_f32_kernel (score order of this file's
// header). blockIdx.y picks 16 of the queries (a 32-KiB f32 image): the 25 MB of sample rows are read up to four times
// (from L2 after the first), and in exchange a workgroup fits on a CU - LDS and registers - beside a coarse scan
// workgroup of another batch in flight; with the 128-KiB image of all 64 queries this launch waited for that whole scan
// to drain, and its row tile + 16 accumulators spilled 252 B. A NaN score is stored as -inf.
// QG = 2 (the wide pass: up to 1024 queries): 32 queries per workgroup. With 16, every (row tile, query group) unit loaded its
// 32 KiB of rows for 128 MFMAs - 8 B per cycle and SIMD, the CU's whole L2 -> register rate - and the launch took 287 us for
// 83 us of f32 MFMA work at Q = 1024; 32 queries per row tile halve the loads per MFMA and give each
// wave two independent accumulator chains.
template <int E, int QG = 1>
__global__ void __launch_bounds__(256) sample_scores_kernel(const float* __restrict__ db, long long nrows,
                                                            const float* __restrict__ q, int QA, uint2* __restrict__ cand,
                                                            long long cap, unsigned* __restrict__ gcnt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = E / 16;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;
    const int qbase = blockIdx.y * 16 * QG;
    f32x4* qimg = reinterpret_cast<f32x4*>(smem);
    // QG*NT*64 = 2048 entries over 256 threads, 8 loads in flight per thread (one at a time cost ~10 us per block)
    for (int base = tid; base < QG * NT * 64; base += 8 * 256) {
        f32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int idx = base + j * 256;
            const int l = idx & 63, t = (idx >> 6) % NT, qg = (idx >> 6) / NT;
            const int c_ = qbase + qg * 16 + (l & 15), g_ = l >> 4;
            v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c_ < QA) v[j] = *reinterpret_cast<const f32x4*>(q + (size_t)c_ * E + 16 * t + 4 * g_);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) qimg[base + j * 256] = v[j];
    }
    if (blockIdx.x == 0 && tid < 16 * QG && qbase + tid < QA) gcnt[qbase + tid] = (unsigned)nrows;
    __syncthreads();
    const long long ntiles = (nrows + 15) >> 4;
    const long long last_row = nrows - 1;
    for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntiles; tile += (long long)gridDim.x * 4) {
        long long r = tile * 16 + col;
        r = r > last_row ? last_row : r;
        const float* p = db + r * E + 4 * g;
        f32x4 T[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) T[t] = *reinterpret_cast<const f32x4*>(p + 16 * t);
        // all 32 loads of the row tile are in flight before the first MFMA: left to itself the scheduler sank them into the
        // MFMA chain two at a time (vmcnt(1) after every load: 32 dependent L2 round trips per tile, ~16 us per tile against
        // 1.7 us of MFMA - round 4, hipcc -S)
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[QG];
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) acc[qg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f32x4 bq[QG];
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) bq[qg] = qimg[(qg * NT + t) * 64 + lane];
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) acc[qg] = __builtin_amdgcn_mfma_f32_16x16x4f32(T[t].x, bq[qg].x, acc[qg], 0, 0, 0);
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) acc[qg] = __builtin_amdgcn_mfma_f32_16x16x4f32(T[t].y, bq[qg].y, acc[qg], 0, 0, 0);
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) acc[qg] = __builtin_amdgcn_mfma_f32_16x16x4f32(T[t].z, bq[qg].z, acc[qg], 0, 0, 0);
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) acc[qg] = __builtin_amdgcn_mfma_f32_16x16x4f32(T[t].w, bq[qg].w, acc[qg], 0, 0, 0);
        }
        // accumulator qg: column = query 16*qg + col, rows = tile*16 + 4g + r_: a lane's four rows of one query are
        // 32 contiguous bytes of that query's list (two 16-byte stores; the four lane groups complete the 128-byte line)
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            const int qi = qbase + qg * 16 + col;
            const long long row = tile * 16 + 4 * g;
            unsigned sb[4];
#pragma unroll
            for (int r_ = 0; r_ < 4; ++r_) {
                const float sc = acc[qg][r_];
                sb[r_] = __float_as_uint(sc == sc ? sc : -INFINITY);
            }
            if (qi < QA) {
                uint2* dst = cand + (size_t)qi * cap + row;          // cap and row are multiples of 4: 32-byte aligned
                if (row + 3 <= last_row) {
                    *reinterpret_cast<uint4*>(dst) = make_uint4(sb[0], (unsigned)row, sb[1], (unsigned)row + 1);
                    *reinterpret_cast<uint4*>(dst + 2) = make_uint4(sb[2], (unsigned)row + 2, sb[3], (unsigned)row + 3);
                } else {
#pragma unroll
                    for (int r_ = 0; r_ < 4; ++r_)
                        if (row + r_ <= last_row) dst[r_] = make_uint2(sb[r_], (unsigned)(row + r_));
                }
            }
        }
    }
}

// Composite 64-bit key: larger = better under the ordering rule. High word = order-preserving
// image of the f32 score (-0 folded into +0), low word = ~id (smaller id wins ties). Keys of
// distinct rows are distinct, so "the K largest keys" is exactly the rule's top-K.
__device__ __forceinline__ unsigned long long ckey(uint2 c) {
    const float s = __uint_as_float(c.x) + 0.0f;
    unsigned u = __float_as_uint(s);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned long long)(~c.y);
}

constexpr int SEL_THREADS = 512;
constexpr int SEL_STAGE = 12288;      // candidate entries a select block keeps in LDS (96 KiB); longer lists stream from L2
constexpr int SEL_FIXED = 32 * 8 + (256 + 8 + 8) * 4;
// large K leaves less room beside the K-entry result buffer; host (LDS size) and device (staging test) share this
__host__ __device__ constexpr int sel_stage_entries(int K, int limit = LDS_LIMIT) {
    return (limit - SEL_FIXED - K * 8) / 8 < SEL_STAGE ? ((limit - SEL_FIXED - K * 8) / 8 < 0 ? 0 : (limit - SEL_FIXED - K * 8) / 8)
                                                       : SEL_STAGE;
}

// One block per query: exact top-K of a dense, unsorted candidate list by MSB-first radix
// select on the composite key (8-bit digits, starting at the highest bit in which the list's
// keys differ), then a rank-count sort of the K survivors. All passes stream the list from
// global memory (it is L2-resident: a few thousand 8-byte entries) - unless it fits SEL_STAGE entries, in which case
// the block copies it into LDS once (loads batched four deep) and every pass runs out of LDS: the per-pass loop
// (load, LDS atomic, next load) otherwise pays one L2 round trip per 512 entries, ~17 us for a 7 k-entry list
// against ~5 us staged (seven selects per 64-query search).
__global__ void __launch_bounds__(SEL_THREADS) select_topk_kernel(const uint2* __restrict__ cand,
                                                                 unsigned* __restrict__ gcnt, long long cap,
                                                                 int K, long long id_base, float* out_s,
                                                                 long long* out_i, float* thr_out,
                                                                 const unsigned* run_if, unsigned* m_out = nullptr,
                                                                 int keep = 0, int stage_cap = -1, float* qmeta = nullptr,
                                                                 int qs = 64, unsigned* live_keys = nullptr,
                                                                 float* live_edges = nullptr, int dense = 0,
                                                                 unsigned* arm_fallback = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (run_if && *run_if == 0) return;
    // the sample's select of a coarse search: a query without a usable coarse bound (NaN / infinite component: its margin is +inf,
    // coarse_prep_kernel) arms the exact fallback for the call - the coarse lists canonicalise NaN scores to -inf, which only the
    // exact scan tells from a true -inf (the oracle never returns a NaN-scoring row). (Set here, not by coarse_prep_kernel: that
    // launch also CLEARS the flag.)
    if (arm_fallback && qmeta && threadIdx.x == 0 && qmeta[3 * qs + blockIdx.x] == INFINITY) *arm_fallback = 1u;
    unsigned long long* red = reinterpret_cast<unsigned long long*>(smem);   // [16] min, [16] max
    unsigned* hist = reinterpret_cast<unsigned*>(red + 32);                  // [256]
    unsigned* wsum = hist + 256;                                             // [8]
    unsigned* ctl = wsum + 8;                                                // [8] b, above, nsel
    uint2* sel = reinterpret_cast<uint2*>(ctl + 8);                          // [K]
    uint2* lent = sel + K;                                                   // [SEL_STAGE] staged copy of the list

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = blockIdx.x;
    long long M = gcnt[q];
    if (M > cap) M = cap;
    const uint2* src = cand + (size_t)q * cap;
    const int need = (int)(M < K ? M : K);
    // stage_cap: entries of LDS the launch reserved behind the K-entry result buffer (-1: the most K leaves room for)
    const bool staged = M <= (stage_cap < 0 ? sel_stage_entries(K) : stage_cap);
    // dense (the wide pass's sample lists: entry e IS row e - sample_scores_kernel writes cand[q][row] = (score, row)): only the
    // 4 score bytes of an entry are staged, the id is the index - 48 KiB instead of 96 for 12 288 rows, so three select blocks
    // fit on a CU instead of one (1024 of them per call of 1024 queries: 121 -> ~50 us)
    unsigned* const l32 = reinterpret_cast<unsigned*>(lent);
    if (staged && dense) {
        for (long long e0 = tid; e0 < M; e0 += 4 * SEL_THREADS) {
            unsigned v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long e = e0 + (long long)j * SEL_THREADS;
                v[j] = src[e < M ? e : M - 1].x;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long e = e0 + (long long)j * SEL_THREADS;
                if (e < M) l32[e] = v[j];
            }
        }
        __syncthreads();
    } else if (staged) {
        for (long long e0 = tid; e0 < M; e0 += 4 * SEL_THREADS) {
            uint2 v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long e = e0 + (long long)j * SEL_THREADS;
                v[j] = src[e < M ? e : M - 1];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const long long e = e0 + (long long)j * SEL_THREADS;
                if (e < M) lent[e] = v[j];
            }
        }
        __syncthreads();
    }
    // the passes below exist twice, once per address space of the list (a pointer select would make every access a
    // flat load, each waiting on both counters in a loop that also holds LDS atomics: 54 us per select, measured)
    auto passes = [&](auto ent) {

        unsigned long long T = 0;     // select keys >= T
        if (M > K) {
            unsigned long long kmin = ~0ull, kmax = 0ull;
            for (long long e = tid; e < M; e += SEL_THREADS) {
                const unsigned long long k = ckey(ent(e));
                kmin = k < kmin ? k : kmin;
                kmax = k > kmax ? k : kmax;
            }
    #pragma unroll
            for (int o = 32; o >= 1; o >>= 1) {
                const unsigned long long a = __shfl_xor(kmin, o), b = __shfl_xor(kmax, o);
                kmin = a < kmin ? a : kmin;
                kmax = b > kmax ? b : kmax;
            }
            if (lane == 0) { red[wave] = kmin; red[16 + wave] = kmax; }
            __syncthreads();
            for (int w = 0; w < SEL_THREADS / 64; ++w) {
                kmin = red[w] < kmin ? red[w] : kmin;
                kmax = red[16 + w] > kmax ? red[16 + w] : kmax;
            }
            const unsigned long long diff = kmin ^ kmax;          // != 0: M >= 2 distinct keys
            int hi_shift = 64 - __builtin_clzll(diff);              // low bits not fixed yet, 1..64
            unsigned long long prefix = hi_shift >= 64 ? 0ull : ((kmax >> hi_shift) << hi_shift);
            unsigned r = (unsigned)K;                               // wanted: the r largest keys matching prefix
            while (hi_shift > 0) {
                const int w = hi_shift < 8 ? hi_shift : 8;
                const int shift = hi_shift - w;
                const unsigned dmask = (1u << w) - 1u;
                if (tid < 256) hist[tid] = 0;
                __syncthreads();
                for (long long e = tid; e < M; e += SEL_THREADS) {
                    const unsigned long long k = ckey(ent(e));
                    const bool match = hi_shift >= 64 || (k >> hi_shift) == (prefix >> hi_shift);
                    if (match) atomicAdd(&hist[(unsigned)(k >> shift) & dmask], 1u);
                }
                __syncthreads();
                // inclusive suffix sums S[t] = sum_{u >= t} hist[u] over 256 bins (waves 0..3)
                unsigned h = 0, S = 0;
                if (tid < 256) {
                    h = hist[tid];
                    S = h;
    #pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const unsigned v = __shfl_down(S, o);
                        if (lane + o < 64) S += v;
                    }
                    if (lane == 0) wsum[wave] = S;
                }
                __syncthreads();
                if (tid < 256) {
                    for (int w2 = wave + 1; w2 < 4; ++w2) S += wsum[w2];
                    const unsigned above = S - h;                    // count in bins > tid
                    if (S >= r && above < r) { ctl[0] = (unsigned)tid; ctl[1] = above; ctl[2] = h; }
                }
                __syncthreads();
                const unsigned b = ctl[0], above = ctl[1], cb = ctl[2];
                prefix |= (unsigned long long)b << shift;
                hi_shift = shift;
                r -= above;
                __syncthreads();
                if (cb == r) break;                                  // the whole bin is selected
            }
            T = prefix;
        }

        if (tid == 0) ctl[4] = 0;
        __syncthreads();
        for (long long e = tid; e < M; e += SEL_THREADS) {
            const uint2 c = ent(e);
            if (ckey(c) >= T) {
                const unsigned pos = atomicAdd(&ctl[4], 1u);
                if (pos < (unsigned)need) sel[pos] = c;
            }
        }
        __syncthreads();
        uint2* dst_keep = keep ? const_cast<uint2*>(src) : nullptr;      // every read of src is behind the barrier above
        for (int e = tid; e < need; e += SEL_THREADS) {
            const uint2 me = sel[e];
            const unsigned long long mk = ckey(me);
            int rank = 0;
            for (int j = 0; j < need; ++j) rank += ckey(sel[j]) > mk ? 1 : 0;
            if (dst_keep) dst_keep[rank] = me;
            if (out_s) {
                out_s[(size_t)q * K + rank] = __uint_as_float(me.x);
                out_i[(size_t)q * K + rank] = id_base + (long long)me.y;
            }
            if (thr_out && rank == K - 1) thr_out[q] = __uint_as_float(me.x);
            if (live_keys) {                                   // the sample's best and K-th best: the live scan's ladder
                if (rank == 0) ctl[5] = me.x;
                if (rank == K - 1) ctl[6] = me.x;
            }
            // coarse search: the next segment's scan threshold, (tau - margin) / t_q (coarse_prep_kernel's qmeta)
            // (an infinite margin - a non-finite query, coarse_prep_kernel - means "no threshold", not inf - inf)
            if (qmeta && rank == K - 1)
                qmeta[2 * qs + q] = qmeta[3 * qs + q] == INFINITY ? -INFINITY : (__uint_as_float(me.x) - qmeta[3 * qs + q]) * qmeta[q];
        }
        if (out_s)
            for (int e = need + tid; e < K; e += SEL_THREADS) {
                out_s[(size_t)q * K + e] = -FLT_MAX;
                out_i[(size_t)q * K + e] = -1;
            }
        if (thr_out && need < K && tid == 0) thr_out[q] = -INFINITY;
        if (qmeta && need < K && tid == 0) qmeta[2 * qs + q] = -INFINITY;
    };
    if (staged && dense) passes([&](long long e) -> uint2 { return make_uint2(l32[e], (unsigned)e); });
    else if (staged) passes([&](long long e) -> uint2 { return lent[e]; });
    else passes([&](long long e) -> uint2 { return src[e]; });
    if (live_keys) {
        // live-threshold scan (scan_coarse_live_kernel): exact threshold = the sample's K-th best, ladder of LIVE_NB edges
        // above it in steps of an eighth of the sample top-K's spread; with fewer than K sample rows nothing is filtered
        __syncthreads();
        if (tid == 0) {
            float e0 = -FLT_MAX, dl = 1.0f, tex = -INFINITY, tsc = -INFINITY;
            const float best = need >= K ? __uint_as_float(ctl[5]) : 0.f, kth = need >= K ? __uint_as_float(ctl[6]) : 0.f;
            if (need >= K && fabsf(best) < FLT_MAX && fabsf(kth) < FLT_MAX) {      // finite (NaN compares false)
                e0 = kth;
                dl = (best - kth) * 0.125f;
                const float floor_ = fabsf(kth) * 1e-6f + 1e-12f;
                if (!(dl > floor_)) dl = floor_;
                tex = kth;
                tsc = (kth - qmeta[3 * qs + q]) * qmeta[q];
            }
            live_keys[q] = ((__float_as_uint(tsc) & 0x80000000u) ? ~__float_as_uint(tsc) : (__float_as_uint(tsc) | 0x80000000u));
            live_keys[64 + q] = ((__float_as_uint(tex) & 0x80000000u) ? ~__float_as_uint(tex) : (__float_as_uint(tex) | 0x80000000u));
            live_edges[q] = e0;
            live_edges[64 + q] = dl;
        }
    }
    // leave the counter zeroed for the next scan of this call (every thread read M at entry; the
    // barriers above order that read before this store) — saves a memset node per scan
    // keep = 1 (segmented coarse scan): the K selected entries stay at the head of the list and the next segment's
    // survivors are appended behind them; keep bit 1: add to the measurement counter instead of overwriting it
    if (tid == 0) {
        gcnt[q] = keep & 1 ? (unsigned)need : 0u;
        if (m_out) m_out[q] = (keep & 2 ? m_out[q] : 0u) + (unsigned)M;        // list length(s), for measurement hooks
    }
}

__global__ void fill_empty_kernel(float* out_s, long long* out_i, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { out_s[i] = -FLT_MAX; out_i[i] = -1; }
}

// Cross-rank merge: R lists of K (score f32, id i64; id < 0 = empty) per query, rank counting
// straight from LDS. One block per query.
__global__ void __launch_bounds__(256) merge_lists_i64_kernel(const float* __restrict__ scores, const long long* __restrict__ ids,
                                                              long long rstride_s, long long rstride_i,
                                                              int R, int Q, int K, float* out_s, long long* out_i) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int q = blockIdx.x, tid = threadIdx.x;
    const int n = R * K;
    long long* sid = reinterpret_cast<long long*>(smem);          // [n]
    float* ss = reinterpret_cast<float*>(sid + n);                 // [n]
    int* nvalid = reinterpret_cast<int*>(ss + n);
    if (tid == 0) *nvalid = 0;
    __syncthreads();
    int myvalid = 0;
    for (int e = tid; e < n; e += blockDim.x) {
        const int r = e / K, k = e - r * K;
        const size_t src = (size_t)q * K + k;
        const long long id = ids[(size_t)r * rstride_i + src];
        const float s = scores[(size_t)r * rstride_s + src];
        const bool ok = id >= 0 && s == s;
        sid[e] = ok ? id : -1;
        ss[e] = s;
        myvalid += ok ? 1 : 0;
    }
    atomicAdd(nvalid, myvalid);
    __syncthreads();
    for (int e = tid; e < n; e += blockDim.x) {
        const long long mi = sid[e];
        if (mi < 0) continue;
        const float ms = ss[e];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const long long oi = sid[j];
            const float os = ss[j];
            rank += (oi >= 0 && (os > ms || (os == ms && oi < mi))) ? 1 : 0;
        }
        if (rank < K) {
            out_s[(size_t)q * K + rank] = ms;
            out_i[(size_t)q * K + rank] = mi;
        }
    }
    const int nv = *nvalid;
    for (int e = (nv < K ? nv : K) + tid; e < K; e += blockDim.x) {
        out_s[(size_t)q * K + e] = -FLT_MAX;
        out_i[(size_t)q * K + e] = -1;
    }
}

// =================================================================================================
// Coarse-then-exact path (SURVEY.md §7 option (b)): a bf16 copy of the matrix is scanned with bf16
// MFMA (half the HBM bytes, 64 queries per pass) keeping a PROVABLE SUPERSET of the exact top-K; the
// survivors are re-scored in exact f32 with the same fmaf order as the exact kernel and selected by the
// same radix select, so results stay bit-exact against the oracle.
//
// Superset argument. Let s(r) be the exact (fmaf-chain) score of row r and c(r) the bf16-MFMA score of
// the bf16-rounded operands. bf16 RNE has relative error u = 2^-9 per operand, so the exact products
// differ by at most (2u + u^2)|a_k b_k|; both accumulations (f32, 512 terms, any order) add at most
// ~2 * 512 * 2^-24 * sum|a_k b_k|. With sum|a_k b_k| <= ||row|| ||q||:
//        |c(r) - s(r)| <= 0.004 * ||row|| * ||q||  <=  margin_q := 0.0041 * R_max * ||q||
// (R_max = largest row norm, kept with the bf16 copy). tau0_q = exact K-th best of the first S rows is
// a lower bound of the final K-th best score; every row of the true top-K has s >= tau0_q, hence
// c >= tau0_q - margin_q =: the coarse threshold. If a candidate list overflows its capacity a device
// flag makes the (otherwise early-exiting) exact-scan fallback launches run and overwrite the result.
// =================================================================================================
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {
    f32x2v v = {a, b};
    bf16x2v r = __builtin_convertvector(v, bf16x2v);
    return __builtin_bit_cast(unsigned, r);
}

// ---- int8 coarse copy (clipmi_quantize_rows_i8 / clipmi_topk_ip_coarse_i8) -------------------------------
// Row r: s_r = (max |x| over r's 32-row block) / 127, q_rk = rint(x_rk / s_r) in [-127, 127], a_r = 1.001 * ||x_r - s_r q_r||_2.
// Query: t_q, p_qk the same way, f_q = y - t_q p_q. The int8 MFMA gives the EXACT integer D = q_r . p_q
// (|D| <= 512 * 127^2 < 2^24), and
//      x.y = s_r t_q D + e_r.y + (s_r q_r).f_q      =>      |x.y - s_r t_q D| <= a_r ||y|| + (R_max + A_max) ||f_q||
// (Cauchy-Schwarz twice; ||s_r q_r|| <= ||x_r|| + a_r). A row can be in the exact top-K only if its exact score is
// >= tau_q (the exact K-th best of a sample), so the scan keeps row r for query q when
//      s_r t_q D + a_r * 1.001 ||y|| >= tau_q - 1.001 (R_max + A_max) ||f_q|| - 1e-4 R_max ||y||
// (the last term covers f32 rounding of both sides and of the exact kernel's fmaf chain: < 512 * 2^-23 relative).
// Divided by t_q > 0 so that the kernel's test is one convert + one multiply + one fma + one compare per pair.
// (A second int8 digit of the query - y = t p + u p' + f', a second MFMA chain per row tile - removes the query half of
//  the margin and was measured: 45 % fewer rows reach the exact re-scoring pass (-30 us per 64 queries at 10 M rows), but
//  with 128 MFMAs per 32-row step the one wave per SIMD is MFMA/VALU-bound in the early segments: scans +75 us. Rejected.)
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Layout of the copy (round 3): 32-row blocks, each [E / 32 k-steps][64 lanes][16 bytes] = the A operand of
// v_mfma_i32_32x32x32_i8 as it lies in registers - lane l of k-step s holds bytes 32 s + 16 (l >> 5) .. + 15 of row
// (l & 31) of the block - so a wave reads one k-step of a block as ONE contiguous KiB. The 32 rows of a block share
// their scale (s = the block's max |x| / 127): a wave's compare then has one scale per step and becomes an integer
// test (scan_coarse_wide_kernel). meta[r] = (s_block, a_r); bmeta[block] = (s_block, max a_r of the block), stored
// behind the row meta. Rows past N in the last block are zero.
// Round 4: the copy is a PERMUTATION of the rows. Slot t of the copy holds row perm[t] (perm = null: the identity); callers
// order the rows by their largest |component| (IndexFlatIP.matrix_i8), so that the 32 rows of a block have nearly the same
// maximum and the block's scale is (almost) each row's own: the error norms a_r - and with them the coarse bound's margin -
// drop back to the per-row-scale values of round 2 (x 1.265 -> x 1.000 on unit rows; re-scored rows per query 5.8 k -> 5.1 k
// at 10 M rows). The scans work on slots; slot_rows[t] = the row a slot holds (0xffffffff for the padding slots N .. N32 - 1),
// stored behind the block meta, is read only for survivors (rescore_pairs_kernel, rescore_pairs16_kernel), which leave the lists as row ids.
__global__ void __launch_bounds__(256) quantize_rows_i8_kernel(const float* __restrict__ db, long long N, int E,
                                                               const unsigned* __restrict__ perm,
                                                               signed char* __restrict__ out, float2* __restrict__ meta,
                                                               float2* __restrict__ bmeta, unsigned* __restrict__ slot_rows) {
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long blk = blockIdx.x, r0 = blk * 32;
    float mx = 0.f;
    for (int i = 0; i < 8; ++i) {
        const long long t = r0 + wave * 8 + i;                     // slot
        const long long r = t < N ? (perm ? (long long)perm[t] : t) : N;
        if (r < N)
            for (int k = lane * 4; k < E; k += 256) {
                const float4 v = *reinterpret_cast<const float4*>(db + (size_t)r * E + k);
                mx = fmaxf(fmaxf(mx, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
            }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float s = mx > 0.f ? mx / 127.0f : 1.0f;
    const float inv = 1.0f / s;
    signed char* oblk = out + (size_t)blk * 32 * E;
    float amax_w = 0.f;
    for (int i = 0; i < 8; ++i) {
        const int rloc = wave * 8 + i;
        const long long t = r0 + rloc;                             // slot
        const long long r = t < N ? (perm ? (long long)perm[t] : t) : N;
        float ee = 0.f;
        for (int k = lane * 4; k < E; k += 256) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < N) v = *reinterpret_cast<const float4*>(db + (size_t)r * E + k);
            float qv[4] = {rintf(v.x * inv), rintf(v.y * inv), rintf(v.z * inv), rintf(v.w * inv)};
            const float xv[4] = {v.x, v.y, v.z, v.w};
            unsigned pk = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                qv[j] = fminf(fmaxf(qv[j], -127.f), 127.f);
                const float e = xv[j] - s * qv[j];
                ee = fmaf(e, e, ee);
                pk |= ((unsigned)(int)qv[j] & 0xffu) << (8 * j);
            }
            const int ks = k >> 5, h = (k >> 4) & 1, b = k & 15;
            *reinterpret_cast<unsigned*>(oblk + (size_t)ks * 1024 + (h * 32 + rloc) * 16 + b) = pk;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ee += __shfl_xor(ee, o);
        const float a_r = r < N ? sqrtf(ee) * 1.001f : 0.f;
        if (lane == 0) {
            meta[t] = make_float2(s, a_r);
            slot_rows[t] = r < N ? (unsigned)r : 0xffffffffu;
        }
        amax_w = fmaxf(amax_w, a_r);
    }
    if (lane == 0) red[4 + wave] = amax_w;
    __syncthreads();
    if (tid == 0) bmeta[blk] = make_float2(s, fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7])));
}

// Per-query constants of a coarse search, once per group of <= qs queries (one wave per query), qmeta [4][qs]
// (qs = 64 for the 64-query passes, the padded query count of a wide pass):
//   [0]    1 / t_q (int8; t_q = max|y| / 127)            or 1 (bf16)
//   [qs]   1.001 ||y|| / t_q (int8)                       or 0
//   [2 qs] the value the scan compares against, (tau - margin) * [0] - written by select_topk_kernel each time a new exact
//          K-th best tau is known (-inf until then)
//   [3 qs] margin: 1.001 (R_max + A_max) ||f_q|| + 1e-4 R_max ||y|| (int8, f_q = y - t_q p_q) or 1.001 * 0.0041 R_max ||q||
//         (bf16); +inf when it is not a number (the threshold then stays -inf: everything survives, the result stays exact)
// and the query image the scan kernels copy into LDS (entry [(qg*KS + s)*64 + lane] = 16 bytes of query 16 qg + (lane & 15):
// bf16 k = 32 s + 8 g .. + 7, or int8 k = 64 s + 16 g .. + 15, g = lane >> 4; zero for queries >= QA). Block 0 also clears
// the call's control words (candidate counters, overflow flag): no memset node.
// q2 (int8, 64-query passes; round 4): the query as TWO int8 digits, y = t p1 + (t / 254) p2 + f' - the second image sits
// COARSE_IMG2 entries behind the first and the margin is built from ||f'|| (1 / 254 of ||f||): the query's half of the
// bound's margin is gone, the scan pays a second MFMA per fragment (it is HBM-latency bound with the matrix pipe 16 % busy).
constexpr int COARSE_IMG2 = 4 * 8 * 64;          // entries of a full 64-query int8 image (32 KiB)
template <bool I8>
__global__ void __launch_bounds__(256) coarse_prep_kernel(const float* __restrict__ q, int E, float rmax, float amax, int QA,
                                                          float* qmeta, uint4* qimage, unsigned* ctl, int nctl, int qs,
                                                          int wide = 0, int q2 = 0) {
    constexpr int KS = 512 / (I8 ? 64 : 32);
    const int lane = threadIdx.x & 63;
    const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (blockIdx.x == 0)
        for (int i = threadIdx.x; i < nctl; i += 256) ctl[i] = 0u;
    const int qg = qi >> 4, col = qi & 15;
    // wide pass (int8): the B operand of v_mfma_i32_32x32x32_i8 - 32-query groups, 16 k-steps of 32 bytes, entry
    // [(G*16 + s)*64 + h*32 + n] = bytes 32 s + 16 h .. + 15 of query 32 G + n
    const int widx = ((qi >> 5) * 16 + (lane >> 1)) * 64 + (lane & 1) * 32 + (qi & 31);
    if (qi >= QA) {                                            // padding query of a used query group: zero image
        if (wide) { if (lane < 32) qimage[widx] = make_uint4(0u, 0u, 0u, 0u); }
        else if (lane < KS * 4) {
            qimage[(qg * KS + (lane >> 2)) * 64 + (lane & 3) * 16 + col] = make_uint4(0u, 0u, 0u, 0u);
            if (I8 && q2) qimage[COARSE_IMG2 + (qg * KS + (lane >> 2)) * 64 + (lane & 3) * 16 + col] = make_uint4(0u, 0u, 0u, 0u);
        }
        return;
    }
    const float* y = q + (size_t)qi * E;
    float mx = 0.f, ss = 0.f;
    for (int k = lane; k < E; k += 64) { const float v = y[k]; mx = fmaxf(mx, fabsf(v)); ss = fmaf(v, v, ss); }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { mx = fmaxf(mx, __shfl_xor(mx, o)); ss += __shfl_xor(ss, o); }
    const float t = mx > 0.f ? mx / 127.0f : 1.0f;
    const float inv = I8 ? 1.0f / t : 1.0f;
    const float t2 = t * (1.0f / 254.0f), inv2 = 1.0f / t2;             // second digit's step (q2)
    // both digits of component v and what they leave; the query images below use the same expressions
    auto digits = [&](float v, float& p1, float& p2) -> float {
        p1 = fminf(fmaxf(rintf(v * inv), -127.f), 127.f);
        const float r1 = fmaf(-t, p1, v);
        if (!q2) { p2 = 0.f; return r1; }
        p2 = fminf(fmaxf(rintf(r1 * inv2), -127.f), 127.f);
        return fmaf(-t2, p2, r1);
    };
    float ff = 0.f;
    if (I8) {
        for (int k = lane; k < E; k += 64) {
            float p1, p2;
            const float f = digits(y[k], p1, p2);
            ff = fmaf(f, f, ff);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) ff += __shfl_xor(ff, o);
    }
    if (lane == 0) {
        const float Y = sqrtf(ss), F = sqrtf(ff) * 1.001f;
        float margin = I8 ? 1.001f * (rmax + amax) * F + 1e-4f * rmax * Y : 0.0041f * rmax * Y * 1.001f;
        // A query with a NaN or infinite component has no usable coarse bound: its margin is +inf, which the selects turn into a
        // threshold of -inf whatever the K-th best is (inf - inf would be a NaN that passes nothing) - every row then passes, the
        // lists overflow and the exact fallback answers the call, as the contract promises for any data. The other two
        // constants stay finite so that the scans' left-hand sides do (round 5: an infinite component used to give inv = 0 and
        // Y inv = NaN, and such a query came back empty - test_non_finite_queries_through_the_permuted_int8_copy).
        const bool finite_q = fabsf(Y) < INFINITY && fabsf(mx) < INFINITY && fabsf(margin) < INFINITY;      // false for NaN too
        if (!finite_q) margin = INFINITY;
        qmeta[qi] = finite_q ? inv : 1.0f;
        qmeta[qs + qi] = I8 && finite_q ? 1.001f * Y * inv : 0.f;
        qmeta[2 * qs + qi] = -INFINITY;
        qmeta[3 * qs + qi] = margin;
    }
    if (I8 && wide) {
        if (lane < 32) {
            const float* p = y + 32 * (lane >> 1) + 16 * (lane & 1);
            unsigned w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned pk = 0;
#pragma unroll
                for (int b_ = 0; b_ < 4; ++b_) {
                    const float pq = fminf(fmaxf(rintf(p[4 * j + b_] * inv), -127.f), 127.f);
                    pk |= ((unsigned)(int)pq & 0xffu) << (8 * b_);
                }
                w[j] = pk;
            }
            qimage[widx] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        return;
    }
    if (lane < KS * 4) {
        const int s_ = lane >> 2, g_ = lane & 3;
        uint4 v;
        if (I8) {
            const float* p = y + 64 * s_ + 16 * g_;
            unsigned w[4], w2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                unsigned pk = 0, pk2 = 0;
#pragma unroll
                for (int b_ = 0; b_ < 4; ++b_) {
                    float p1, p2;
                    (void)digits(p[4 * j + b_], p1, p2);
                    pk |= ((unsigned)(int)p1 & 0xffu) << (8 * b_);
                    pk2 |= ((unsigned)(int)p2 & 0xffu) << (8 * b_);
                }
                w[j] = pk;
                w2[j] = pk2;
            }
            v = make_uint4(w[0], w[1], w[2], w[3]);
            if (q2) qimage[COARSE_IMG2 + (qg * KS + s_) * 64 + g_ * 16 + col] = make_uint4(w2[0], w2[1], w2[2], w2[3]);
        } else {
            const float* p = y + 32 * s_ + 8 * g_;
            v = make_uint4(pack2_bf16(p[0], p[1]), pack2_bf16(p[2], p[3]), pack2_bf16(p[4], p[5]), pack2_bf16(p[6], p[7]));
        }
        qimage[(qg * KS + s_) * 64 + g_ * 16 + col] = v;
    }
}

// A coarse scan appends (CAND_SLOT, slot of the copy); the re-scoring pass turns such an entry into (exact score bits, row id).
// No score has these bits (a NaN score is stored as -inf), so an entry that was re-scored before - the K best a select keeps at
// the head of the list - is recognised and keeps its row id.
// INVARIANT: every writer of cand[].x - sample_scores_kernel, rescore_pairs_kernel, rescore_pairs16_kernel and the
// heads select_topk_kernel keeps (it copies .x unchanged) - stores either CAND_SLOT with a slot, or the RAW bits of a score that is
// not a NaN (NaN -> -inf before the store). Never an fkey()-encoded key, never raw bits of an unchecked score: 0xffffffff is a NaN
// pattern and would be read as "slot". tests/test_topk_gpu.py::test_non_finite_queries_through_the_permuted_int8_copy holds it.
constexpr unsigned CAND_SLOT = 0xffffffffu;
constexpr size_t RESCORE16_LDS = 512 * 4 + 4 * 16 * 68 * 4;      // rescore_pairs16_kernel: the query + four waves' tiles of 16 row chunks

struct CoarseArgs {
    const void* dbc;             // coarse copy of the matrix: bf16 [nrows][E] or int8 [nrows][E]
    const float2* rmeta;         // int8 only: per row (scale s_r, error norm a_r >= ||x_r - s_r q_r||), padded to 32 rows
    const float* qmeta;          // [4][64], see coarse_prep_kernel: 1/t_q | 1.001 ||y|| / t_q | scaled threshold | margin
    const uint4* qimage;         // the query image in the scan's LDS layout (coarse_prep_kernel)
    long long row0;              // this launch scans rows [row0, row0 + nrows) of the copy; row0 % 32 == 0 (ids stay global)
    long long nrows;
    const float* q;              // f32 [QA][E]
    int QA;                      // 1..16*QG
    const float* tauc;           // [QA]
    uint2* cand;                 // [QA][cap]: .y = row id (.x filled by the re-scoring pass)
    unsigned* gcnt;              // [128]
    long long cap;
    unsigned* overflow;          // set to 1 when a list would exceed cap
    int abl;                     // development ablation (CLIPMI_COARSE_ABL=1): never append - timing only, results wrong
};

// Per-wave (query, row) list in LDS. One (row tile, query group) block of a step appends at most 4 rows x 64 lanes =
// 256 pairs and the flush test runs after every such block, so the list holds COARSE_FLUSH + 256 entries: it cannot
// overrun by construction (and the append is bound-checked all the same: a miss arms the exact fallback). 6 KiB per
// wave keeps a 64-query int8 scan workgroup at 56 KiB of LDS, so the small kernels of ANOTHER batch in flight (exact
// re-scoring 72 KiB, selects, the sample scan 64 KiB) fit on the same CU instead of waiting for the scan to drain.
constexpr int COARSE_QS = 64;           // qmeta stride of the 64-query passes
constexpr int COARSE_FLUSH = 512;       // flush when more than this many are pending
constexpr int COARSE_LIST = COARSE_FLUSH + 256;
constexpr size_t COARSE_WAVE_BYTES = (size_t)COARSE_LIST * 8 + 16;

// publish a wave's n pending pairs: one global atomic per pair (rare path: ~1 pair in 10^4 passes the threshold)
__device__ __noinline__ void coarse_flush(const uint2* list, int n, unsigned* gcnt, uint2* cand, long long cap, unsigned* overflow) {
    const int lane = threadIdx.x & 63;
    wave_lds_sync();
    for (int e = lane; e < n; e += 64) {
        const uint2 c = list[e];
        const unsigned pos = atomicAdd(&gcnt[c.x], 1u);
        if ((long long)pos < cap) cand[(size_t)c.x * cap + pos] = make_uint2(CAND_SLOT, c.y);
        else *overflow = 1u;
    }
    wave_lds_sync();
}

// PREPASS only changes the kernel's NAME (the level-2 pre-pass over S2 rows must not dilute the profiler's
// per-name average of the main scan).
// Q2 (int8 only): two query digits (coarse_prep_kernel q2) - 2 QG MFMAs per fragment, D = D1 + D2 / 254.
// 16 bytes of the streamed coarse copy (CLIPMI_NT_MASK bit 0: non-temporal)
__device__ __forceinline__ uint4 ld_stream(const char* p) {
    typedef unsigned u4v __attribute__((ext_vector_type(4)));
    if constexpr (CLIPMI_NT_MASK & 1) {
        const u4v v = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p));
        return make_uint4(v[0], v[1], v[2], v[3]);
    } else
        return *reinterpret_cast<const uint4*>(p);
}

template <int E, int QG, bool PREPASS, bool I8, bool Q2 = false>
__global__ void __launch_bounds__(256) scan_coarse_kernel(CoarseArgs a) {
    static_assert(!Q2 || I8, "the second query digit belongs to the int8 copy");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = E / (I8 ? 64 : 32);   // MFMA k-steps per 16-row tile (16 bytes per lane and step either way)
    constexpr int ROWB = I8 ? E : 2 * E;     // bytes per row of the coarse copy
    constexpr int SLOTS = 2 * KS;       // one step = two row tiles (32 rows): SLOTS 16-byte fragments per lane
    constexpr int NIMG = QG * KS * 64;                  // 16-byte entries of one query image
    constexpr int ND = Q2 ? 2 : 1;                      // query digits
    constexpr int IMG_BYTES = ND * NIMG * 16;
    static_assert(COARSE_LIST >= COARSE_FLUSH + 4 * 64, "a block's appends must fit behind a pending flush");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwaves = blockDim.x >> 6;
    const int col = lane & 15, g = lane >> 4;

    // query image: built once per search by coarse_prep_kernel (each of 256 workgroups x 3 launches used to read the 128 KiB
    // of f32 queries and quantise them again); here it is a 16-byte-per-lane copy, 8 loads in flight
    uint4* qimg = reinterpret_cast<uint4*>(smem);
    {
        static_assert(NIMG % (8 * 256) == 0 || NIMG < 8 * 256, "image copy loop");
        for (int base = tid; base < NIMG; base += 8 * 256) {
            uint4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = base + j * 256 < NIMG ? a.qimage[base + j * 256] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (base + j * 256 < NIMG) qimg[base + j * 256] = v[j];
            if constexpr (Q2) {                                  // the second digit's image, NIMG entries behind the first in LDS
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = base + j * 256 < NIMG ? a.qimage[COARSE_IMG2 + base + j * 256] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (base + j * 256 < NIMG) qimg[NIMG + base + j * 256] = v[j];
            }
        }
    }
    uint2* list = reinterpret_cast<uint2*>(smem + IMG_BYTES + (size_t)wave * COARSE_WAVE_BYTES);
    int npend = 0;                                     // wave-uniform: pairs pending in `list` (positions by ballot prefix)
    __syncthreads();

    bool active[QG];
    float tau[QG], yt[QG];
#pragma unroll
    for (int qg = 0; qg < QG; ++qg) {
        active[qg] = qg * 16 + col < a.QA;
        tau[qg] = active[qg] ? a.qmeta[2 * COARSE_QS + qg * 16 + col] : INFINITY;
        yt[qg] = (I8 && active[qg]) ? a.qmeta[COARSE_QS + qg * 16 + col] : 0.f;
    }

    // steps are counted from row 0 of the copy (row0 % 32 == 0), so row = 32 step + ... is the global row id
    const long long step0 = a.row0 >> 5;
    const long long nsteps = step0 + ((a.nrows + 31) >> 5);
    const long long wg = (long long)blockIdx.x * nwaves + wave;
    const long long tw = (long long)gridDim.x * nwaves;
    const long long last_row = a.row0 + a.nrows - 1;

    long long step = step0 + wg;
    if (step < nsteps) {
        uint4 T[SLOTS];
        // bf16 copy: row-major, 64 bytes per row and k-step; int8 copy: 32-row blocks of [16 k-steps of 32 B][64 lanes][16 B]
        // (quantize_rows_i8_kernel), padded to whole blocks: this lane's 16 bytes of MFMA k-step s are bytes 64 s + 16 g
        // of row 16 rt + col = 32-byte k-step 2 s + (g >> 1), half g & 1
        constexpr int KSTRIDE = I8 ? 2048 : 64;
        auto frag_ptr = [&](long long st, int rt) {
            if (I8)
                return static_cast<const char*>(a.dbc) + st * (32 * ROWB) + (g >> 1) * 1024 + ((g & 1) * 32 + rt * 16 + col) * 16;
            long long r = st * 32 + rt * 16 + col;
            r = r > last_row ? last_row : r;
            return static_cast<const char*>(a.dbc) + r * ROWB + 16 * g;
        };
        // int8: (s_r, a_r) of this lane's accumulator rows 4g .. 4g+3 of both row tiles (rmeta is padded to 32 rows)
        uint4 M[4] = {};
        auto load_meta = [&](long long st, uint4* m) {
            if (I8) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const uint4* mp = reinterpret_cast<const uint4*>(a.rmeta + st * 32 + rt * 16 + 4 * g);
                    m[2 * rt] = mp[0];
                    m[2 * rt + 1] = mp[1];
                }
            }
        };
        {
            const char* p0 = frag_ptr(step, 0);
            const char* p1 = frag_ptr(step, 1);
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_) {
                T[s_] = ld_stream(p0 + KSTRIDE * s_);
                T[KS + s_] = ld_stream(p1 + KSTRIDE * s_);
            }
            load_meta(step, M);
        }
        using AccT = typename std::conditional<I8, i32x4, f32x4>::type;
        while (true) {
            const long long nxt = step + tw;
            const bool has_next = nxt < nsteps;
            const char* pn[2] = {frag_ptr(has_next ? nxt : step, 0), frag_ptr(has_next ? nxt : step, 1)};
            uint4 MN[4] = {};
            load_meta(has_next ? nxt : step, MN);

            AccT acc[2][ND * QG];                       // [rt][digit * QG + qg]
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int qg = 0; qg < ND * QG; ++qg) acc[rt][qg] = AccT{0, 0, 0, 0};
            // One slot at a time: the LDS reads of the NEXT slot's query fragments, this slot's MFMAs, then the refill of
            // this slot's register with the next step's fragment. A load can only be issued once the MFMAs reading its
            // register have been, and its data is needed one step later, so a step lasts (HBM latency + the MFMA time
            // between two refills): refilling after every slot instead of every 8 (the first version) keeps that second
            // term at QG MFMAs - main scan 629-648 -> 619 us at 10 M rows. The sched_group_barriers state the order; no
            // fake dependency on the accumulators is needed.
            uint4 B[2][ND * QG];
            auto load_b = [&](int s_, uint4* b) {
#pragma unroll
                for (int d = 0; d < ND; ++d)
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg) b[d * QG + qg] = qimg[d * NIMG + (qg * KS + s_) * 64 + lane];
            };
            __builtin_amdgcn_sched_barrier(0);
            load_b(0, B[0]);
#pragma unroll
            for (int slot = 0; slot < SLOTS; ++slot) {
                const int rt = slot / KS;
                // the second row tile re-reads the fragments from LDS (left to itself the compiler keeps all of the first
                // tile's in registers: up to 128 more VGPRs, and no other kernel's wave fits on the SIMD any more)
                if (slot + 1 == KS) asm volatile("" ::: "memory");
                if (slot + 1 < SLOTS) load_b((slot + 1) % KS, B[(slot + 1) & 1]);
                const uint4* b = B[slot & 1];
#pragma unroll
                for (int qg = 0; qg < ND * QG; ++qg) {
                    if constexpr (I8)
                        acc[rt][qg] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, T[slot]),
                                                                            __builtin_bit_cast(i32x4, b[qg]), acc[rt][qg], 0, 0, 0);
                    else
                        acc[rt][qg] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, T[slot]),
                                                                              __builtin_bit_cast(bf16x8, b[qg]), acc[rt][qg], 0, 0, 0);
                }
                T[slot] = ld_stream(pn[rt] + KSTRIDE * (slot % KS));
            }
            constexpr int NBQ = ND * QG;
            __builtin_amdgcn_sched_group_barrier(0x100, NBQ, 0);               // slot 0's fragments
#pragma unroll
            for (int slot = 0; slot < SLOTS; ++slot) {
                if (slot + 1 < SLOTS) __builtin_amdgcn_sched_group_barrier(0x100, NBQ, 0);   // DS reads: next slot's fragments
                __builtin_amdgcn_sched_group_barrier(0x008, NBQ, 0);                         // MFMAs of this slot
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                           // VMEM read: the refill
            }
            __builtin_amdgcn_sched_barrier(0);

            // value compared with tau[qg]: the bf16 score itself, or (int8) D * s_r + a_r * 1.001 ||y|| / t_q
            float val[2][QG][4];
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                const float sr[4] = {__uint_as_float(M[2 * rt].x), __uint_as_float(M[2 * rt].z), __uint_as_float(M[2 * rt + 1].x),
                                     __uint_as_float(M[2 * rt + 1].z)};
                const float ar[4] = {__uint_as_float(M[2 * rt].y), __uint_as_float(M[2 * rt].w), __uint_as_float(M[2 * rt + 1].y),
                                     __uint_as_float(M[2 * rt + 1].w)};
#pragma unroll
                for (int qg = 0; qg < QG; ++qg)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float d = (float)acc[rt][qg][r];
                        if constexpr (Q2) d = fmaf((float)acc[rt][QG + qg][r], 1.0f / 254.0f, d);
                        val[rt][qg][r] = I8 ? fmaf(d, sr[r], ar[r] * yt[qg]) : d;
                    }
            }
            bool any = false;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                for (int qg = 0; qg < QG; ++qg)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        any |= active[qg] && (step * 32 + rt * 16 + 4 * g + r <= last_row) && (val[rt][qg][r] >= tau[qg]);
            // Appends (round 4): positions by ballot prefix into a wave-uniform count, as the wide passes do. The first form took one
            // returning LDS atomic per pair and, per (row tile, query group) block, a wave-level LDS sync + a read of the counter for
            // the flush test - eight of each on ~60 % of the steps at 64 queries. With all appends compiled out (CLIPMI_COARSE_ABL=1)
            // the scan is 10-45 us of 0.8 ms faster: that is what any append path can still gain. Blocks without a hit cost one ballot.
            if (__ballot(any) && !a.abl) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg) {
                        bool p[4], anyb = false;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            p[r] = active[qg] && (step * 32 + rt * 16 + 4 * g + r <= last_row) && (val[rt][qg][r] >= tau[qg]);
                            anyb |= p[r];
                        }
                        if (!__ballot(anyb)) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const unsigned long long m = __ballot(p[r]);
                            if (m) {
                                const int pos = npend + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                                // <= 256 appends per block behind at most COARSE_FLUSH pending: inside the list (static_assert above)
                                if (p[r]) list[pos] = make_uint2((unsigned)(qg * 16 + col), (unsigned)(step * 32 + rt * 16 + 4 * g + r));
                                npend += __builtin_popcountll(m);
                            }
                        }
                        if (npend > COARSE_FLUSH) {
                            coarse_flush(list, npend, a.gcnt, a.cand, a.cap, a.overflow);
                            npend = 0;
                        }
                    }
            }
            if (!has_next) break;
            step = nxt;
#pragma unroll
            for (int i = 0; i < 4; ++i) M[i] = MN[i];
        }
    }
    // final publication, aggregated per BLOCK: one returning global atomic per (block, query) issued by
    // 64 lanes at once, then entries scatter to base + an LDS-counted offset. (Per-entry global atomics
    // here meant ~166 k same-address atomics at the end of every scan: ~250 us fixed, as much as
    // streaming 1.25 M rows.)
    __syncthreads();
    constexpr int QMAXC = 16 * QG;                                   // <= blockDim.x for every launch shape
    unsigned* hist = reinterpret_cast<unsigned*>(smem);              // query image is dead now: [QMAXC] counts
    unsigned* gbase = hist + QMAXC;                                  // global base per query
    unsigned* hoff = hist + 2 * QMAXC;                               // running offsets
    if (tid < QMAXC) { hist[tid] = 0; hoff[tid] = 0; }
    __syncthreads();
    const int n_mine = npend;
    for (int e = lane; e < n_mine; e += 64) atomicAdd(&hist[list[e].x], 1u);
    __syncthreads();
    if (tid < QMAXC) gbase[tid] = hist[tid] ? atomicAdd(&a.gcnt[tid], hist[tid]) : 0u;
    __syncthreads();
    for (int e = lane; e < n_mine; e += 64) {
        const uint2 c = list[e];
        const unsigned pos = gbase[c.x] + atomicAdd(&hoff[c.x], 1u);
        if ((long long)pos < a.cap) a.cand[(size_t)c.x * a.cap + pos] = make_uint2(CAND_SLOT, c.y);
        else *a.overflow = 1u;
    }
}

// Exact f32 re-scoring of the coarse survivors: (exact score bits, row id) for every entry of the candidate lists. A lane that
// owns a (query, row) pair runs the SAME fmaf chain as the MFMA scan (t, c, g order) - one serial chain over the row's 512
// components, so the score bits equal the exact kernel's and the oracle's; that chain is why a row cannot be split over lanes and
// the rows go through a per-wave LDS tile: fetched COOPERATIVELY (4 rows x 256 B contiguous per wave-instruction), read back by
// the owning lane. (A first version with one thread fetching its own 2-KB row: ~1 ms for 175 k pairs, 4 uncoalesced loads in flight.)
//
// Two forms. rescore_pairs16_kernel (round 4, the wide passes' lists: up to 1 024 queries x ~500 pairs): a wave takes SIXTEEN pairs
// and issues all 32 loads of their whole rows (32 KB in flight per wave) before it touches the first chunk - one HBM round trip
// per batch, 19 KB of LDS per block, three waves per SIMD; the compute runs on lanes 0-15's rows four times over (lanes l and
// l + 16 k read the same LDS words: broadcasts), 4 x the issue slots per pair and still far below the gathers' time. One call of
// 1 024 queries at 10 M rows: 6.26 -> 6.08-6.12 ms (same box). rescore_pairs_kernel (below): 64 pairs per wave in 8 chunks, kept
// for the 64-query passes, where the two gather at the same rate and this one costs the other call in flight less.
template <int E>
__global__ void __launch_bounds__(256) rescore_pairs16_kernel(const float* __restrict__ db, const float* __restrict__ q,
                                                            uint2* cand, const unsigned* __restrict__ gcnt, long long cap,
                                                            const unsigned* __restrict__ slot_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RS = 68;                                  // floats per staged row chunk (64 + 4 pad: conflict-free b128 reads)
    constexpr int NCH = E / 64;                             // 256-byte chunks per row
    float* qs = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* st = qs + E + wave * 16 * RS;
    const int qi = blockIdx.y;
    for (int k = threadIdx.x; k < E; k += 256) qs[k] = q[(size_t)qi * E + k];
    __syncthreads();
    long long M = gcnt[qi];
    if (M > cap) M = cap;
    uint2* lst = cand + (size_t)qi * cap;
    const int r16 = lane & 15, j4 = lane >> 4;
    for (long long base = ((long long)blockIdx.x * 4 + wave) * 16; base < M; base += (long long)gridDim.x * 4 * 16) {
        const long long my = base + r16;
        const uint2 ent = lst[my < M ? my : M - 1];
        // a fresh survivor names a SLOT of the (permuted) int8 copy: its row id comes from slot_rows (one more dependent read,
        // for survivors only); entries kept from earlier segments are row ids already
        const unsigned id_my = (ent.x == CAND_SLOT && slot_rows) ? slot_rows[ent.y] : ent.y;
        f32x4 nx[4][NCH];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned rid = __shfl(id_my, 4 * i + j4);
            const float* rowp = db + (size_t)rid * E + r16 * 4;
#pragma unroll
            for (int c = 0; c < NCH; ++c) nx[i][c] = ld16_f<2>(rowp + c * 64);
        }
        float acc = 0.f;
#pragma unroll
        for (int chunk = 0; chunk < NCH; ++chunk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(st + (4 * i + j4) * RS + r16 * 4) = nx[i][chunk];
            wave_lds_sync();
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                f32x4 v[4];
#pragma unroll
                for (int g_ = 0; g_ < 4; ++g_) v[g_] = *reinterpret_cast<const f32x4*>(st + r16 * RS + 16 * tt + 4 * g_);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int g_ = 0; g_ < 4; ++g_)
                        acc = __builtin_fmaf(v[g_][c], qs[64 * chunk + 16 * tt + 4 * g_ + c], acc);
                // pin the chain here: with the chunk loop unrolled the compiler sinks all 512 fmaf below the last chunk (acc
                // is only used at the end) and keeps every chunk's LDS reads alive - 512 registers + 550 spilled to scratch
                asm volatile("" : "+v"(acc) :: "memory");
            }
            wave_lds_sync();
        }
        if (j4 == 0 && my < M) lst[my] = make_uint2(__float_as_uint(acc == acc ? acc : -INFINITY), id_my);      // NaN never ranks
    }
}

// The 64-query passes' form (rounds 2-3): 64 pairs per wave, 8 chunks of 16 loads with one chunk of prefetch. Same-box A/B
// (CLIPMI_RESCORE=16 / 64, development library; profiles/r04_search_rescore_ab.txt): its lists (2 500 / 1 500 / 700 pairs per
// query) are gathered at the same ~5 TB/s by either form - 58.6 / 41.6 / 29.3 us against 58.1 / 47.8 / 27.2 at 10 M rows - and
// with two calls in flight this one disturbs the other call's streaming scan less (0.976 against 1.005 ms per call).
constexpr size_t RESCORE_LDS = 512 * 4 + 4 * 64 * 68 * 4;
template <int E>
__global__ void __launch_bounds__(256) rescore_pairs_kernel(const float* __restrict__ db, const float* __restrict__ q,
                                                            uint2* cand, const unsigned* __restrict__ gcnt, long long cap,
                                                            const unsigned* __restrict__ slot_rows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int RS = 68;                                  // floats per staged row chunk (64 + 4 pad: conflict-free b128 reads)
    float* qs = reinterpret_cast<float*>(smem);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float* st = qs + E + wave * 64 * RS;
    const int qi = blockIdx.y;
    for (int k = threadIdx.x; k < E; k += 256) qs[k] = q[(size_t)qi * E + k];
    __syncthreads();
    long long M = gcnt[qi];
    if (M > cap) M = cap;
    uint2* lst = cand + (size_t)qi * cap;
    for (long long base = ((long long)blockIdx.x * 4 + wave) * 64; base < M; base += (long long)gridDim.x * 4 * 64) {
        const long long my = base + lane;
        const uint2 ent = lst[my < M ? my : M - 1];
        // a fresh survivor names a SLOT of the (permuted) int8 copy: its row id comes from slot_rows (one more dependent read,
        // for survivors only); entries kept from earlier segments are row ids already
        const unsigned id_my = (ent.x == CAND_SLOT && slot_rows) ? slot_rows[ent.y] : ent.y;
        float acc = 0.f;
        // chunk c+1's 16 loads are issued (to registers) before chunk c is consumed from LDS: one
        // HBM round trip per chunk stays, but it overlaps the previous chunk's LDS reads and fmaf chain
        f32x4 nx[16];
        const float* rowp[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const unsigned rid = __shfl(id_my, 4 * i + (lane >> 4));
            rowp[i] = db + (size_t)rid * E + (lane & 15) * 4;
            nx[i] = ld16_f<2>(rowp[i]);
        }
        for (int chunk = 0; chunk < E / 64; ++chunk) {
#pragma unroll
            for (int i = 0; i < 16; ++i)
                *reinterpret_cast<f32x4*>(st + (4 * i + (lane >> 4)) * RS + (lane & 15) * 4) = nx[i];
            if (chunk + 1 < E / 64) {
#pragma unroll
                for (int i = 0; i < 16; ++i) nx[i] = ld16_f<2>(rowp[i] + (chunk + 1) * 64);
            }
            wave_lds_sync();
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                f32x4 v[4];
#pragma unroll
                for (int g_ = 0; g_ < 4; ++g_) v[g_] = *reinterpret_cast<const f32x4*>(st + lane * RS + 16 * tt + 4 * g_);
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int g_ = 0; g_ < 4; ++g_)
                        acc = __builtin_fmaf(v[g_][c], qs[64 * chunk + 16 * tt + 4 * g_ + c], acc);
            }
            wave_lds_sync();
        }
        if (my < M) lst[my] = make_uint2(__float_as_uint(acc == acc ? acc : -INFINITY), id_my);      // NaN never ranks
    }
}

// =================================================================================================
// Live-threshold scan (int8 copy, <= 64 queries): ONE launch over all rows, exact re-scoring INSIDE the scan.
//
// The segmented form above re-scores and selects between launches: 13 launches per call, ~250 us of latency-bound side
// kernels beside 0.88 ms of scans at 10 M rows, and 5.8 k exactly re-scored rows per query because a segment's threshold
// is frozen at its start. Here a workgroup is 4 SCANNER waves (the loop of scan_coarse_kernel) + 4 RE-SCORING waves.
// Scanners push the pairs that pass the coarse test into a ring in LDS and never wait for anything else; re-scoring waves
// pop them, drop what a since-risen threshold has made stale (the pair carries its coarse upper bound), gather row and
// query from global memory, run the score-order fmaf chain (the SAME bits as rescore_pairs_kernel and the oracle), and
// append (exact score, row) to the query's global candidate list when it reaches the query's exact threshold.
// Thresholds rise DURING the scan through a per-query ladder in global memory: NB buckets of exact scores above the
// sample's K-th best (edge_b = e0 + b delta, delta = an eighth of the sample top-K's spread); a re-scored row bumps its
// bucket, and whoever bumps re-reads the query's 32 counters: the highest edge with >= K rows at or above it IS a lower
// bound of the final K-th best (K rows score at least that), published with atomicMax as an order-preserving key - the
// exact threshold for the re-scoring waves, (edge - margin) / t_q for the scanners, who re-read their 64 keys every step.
// Exactness never depends on the ladder's timing: every published threshold is a valid lower bound, a true top-K row
// passes the coarse test against any of them and its exact score reaches any of them.
// Every spin is bounded (a wave that waits too long sets the overflow flag = the exact fallback answers, still exact).
// =================================================================================================
constexpr int LIVE_NB = 32;                      // ladder buckets per query
#ifdef CLIPMI_DEV                                // (LIVE_NB also sizes the product's control block: it stays outside)
constexpr int LIVE_QN = 4096;                    // ring entries (32 KiB)
constexpr unsigned LIVE_EMPTY = 0xffffffffu;
constexpr int LIVE_ROW_BITS = 26;                // ring entry .y = (query << 26) | row: shards of < 2^26 rows
constexpr int LIVE_WB = 24;                      // 32-row steps per batch of scanner work a workgroup draws (>= its scanner waves)
constexpr int LIVE_SPIN_LIMIT = 1 << 20;         // polls before a waiting lane gives up (~ tens of ms): overflow -> fallback
#endif

__device__ __forceinline__ unsigned fkey(float f) {             // order-preserving key of a float (-0 and +0 differ: harmless)
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float fkey_inv(unsigned k) {
    return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k);
}

struct LiveArgs {
    const signed char* dbc;      // int8 copy (32-row blocks)
    const float2* rmeta;         // per row (block scale, error norm)
    const unsigned* slot_rows;   // the row each slot of the copy holds (quantize_rows_i8_kernel)
    const float* db;             // f32 [N][512]
    const float* q;              // f32 [QA][512]
    const float* qmeta;          // [4][64]: 1/t | Y/t | (unused here) | margin
    const uint4* qimage;         // 16x16x64 image of the queries (coarse_prep_kernel)
    long long nrows;             // rows [0, nrows)
    int QA, K;
    unsigned* tau_key;           // [64] key of the scaled coarse threshold (live)
    unsigned* tex_key;           // [64] key of the exact threshold (live)
    const float* edge0;          // [64] ladder origin = the sample's exact K-th best
    const float* delta;          // [64] ladder step > 0
    unsigned* hist;              // [64][LIVE_NB]
    uint2* cand;                 // [QA][cap] (exact score bits, row)
    unsigned* gcnt;              // [64]
    long long cap;
    unsigned* overflow;
    unsigned* chunk_ctr;         // work counter of the scanner waves (zero at launch)
    int wb;                      // 32-row steps per work batch (>= the scanner waves of a workgroup)
    int abl;                     // development ablation (CLIPMI_LIVE_ABL): 1 scanners push nothing, 2 re-scoring waves drop every pair
    unsigned* stats;             // [12] pushed | push spins | popped | stale | re-scored | inserted | batches | ladder rises
};

#ifdef CLIPMI_DEV      // experimental: compiled into the development library only
template <int QG, int NSCAN = 4>
__global__ void __launch_bounds__(512) scan_coarse_live_kernel(LiveArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int E = 512, KS = E / 64, ROWB = E, SLOTS = 2 * KS, NIMG = QG * KS * 64, IMG_BYTES = NIMG * 16;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int col = lane & 15, g = lane >> 4;

    const unsigned long long t_begin = a.stats ? wall_clock64() : 0ull;
    uint4* qimg = reinterpret_cast<uint4*>(smem);
    for (int base = tid; base < NIMG; base += 512) qimg[base] = a.qimage[base];
    uint2* ring = reinterpret_cast<uint2*>(smem + IMG_BYTES);                      // [LIVE_QN]
    int* ctl = reinterpret_cast<int*>(ring + LIVE_QN);                             // [0] tail, [1] head, [2] scanners done, [3] abort, [4] batches released
    unsigned* ltau = reinterpret_cast<unsigned*>(ctl + 16);                        // [64] this workgroup's copy of the threshold keys
    unsigned* ltex = ltau + 64;                                                    // [64] ... and of the exact thresholds' keys
    f32x4* lcon = reinterpret_cast<f32x4*>(ltex + 64);                             // [64] ladder origin, ladder step, query scale, query slack
    int* wqb = reinterpret_cast<int*>(lcon + 64);                                  // [8] first step of work batch b (slot b & 7); tags in ctl[8..15]
    for (int i = tid; i < LIVE_QN; i += 512) ring[i] = make_uint2(0u, LIVE_EMPTY);
    if (tid < 8) { ctl[tid] = 0; ctl[8 + tid] = tid == 0 ? 0 : -1; }
    if (tid == 0) {
        // scanner work: batches of a.wb steps. Batch 0 of a workgroup is its index in the grid, every further one comes from
        // a counter (zeroed with the call's control words); batch 1 is drawn here, batch b + 2 by the wave that takes the first
        // step of batch b (ctl[5] = the workgroup's step tickets)
        wqb[0] = (int)blockIdx.x * a.wb;
        wqb[1] = (int)(gridDim.x + atomicAdd(a.chunk_ctr, 1u)) * a.wb;
    }
    if (tid < 64) {
        const bool on = tid < a.QA;
        ltau[tid] = on ? a.tau_key[tid] : 0u;
        ltex[tid] = on ? a.tex_key[tid] : 0u;
        lcon[tid] = on ? f32x4{a.edge0[tid], a.delta[tid], a.qmeta[tid], a.qmeta[3 * COARSE_QS + tid]} : f32x4{0.f, 1.f, 1.f, 0.f};
    }
    __syncthreads();
    if (tid == 0) ctl[9] = 1;                       // batch 1's base is in place (written before the barrier)

    if (wave < NSCAN) {
        // ------------------------------------------------------------------ scanner
        bool active[QG];
        float tau[QG], yt[QG];
#pragma unroll
        for (int qg = 0; qg < QG; ++qg) {
            active[qg] = qg * 16 + col < a.QA;
            tau[qg] = INFINITY;
            yt[qg] = active[qg] ? a.qmeta[COARSE_QS + qg * 16 + col] : 0.f;
        }
        // work: the workgroup's waves take 32-row steps one by one from its current batch (an LDS ticket); batches come from a
        // device-wide counter two ahead. CUs and XCDs stream at different rates: with a fixed interleave the last wave finished
        // ~90 us (10 %) after the average one; per-wave draws from the counter queue a slow returning atomic in front of the
        // wave's row loads (vmcnt is in order) - here a wave publishes the batch it drew at its NEXT visit.
        const long long nsteps = (a.nrows + 31) >> 5;
        int pend_b = -1;
        unsigned grabbed = 0u;
        auto take = [&]() -> long long {
            if (pend_b >= 0) {
                if (lane == 0) {
                    wqb[pend_b & 7] = (int)(gridDim.x + grabbed) * a.wb;
                    *reinterpret_cast<volatile int*>(&ctl[8 + (pend_b & 7)]) = pend_b;
                }
                pend_b = -1;
            }
            int t_ = 0;
            if (lane == 0) t_ = atomicAdd(&ctl[5], 1);
            t_ = __builtin_amdgcn_readfirstlane(t_);
            const int b_ = t_ / a.wb, o_ = t_ - b_ * a.wb;
            if (o_ == 0) {
                if (lane == 0) grabbed = atomicAdd(a.chunk_ctr, 1u);
                pend_b = b_ + 2;
            }
            int sp_ = 0;
            while (*reinterpret_cast<volatile int*>(&ctl[8 + (b_ & 7)]) != b_) {
                __builtin_amdgcn_s_sleep(2);
                if (*reinterpret_cast<volatile int*>(&ctl[3]) || ++sp_ > LIVE_SPIN_LIMIT) {
                    *a.overflow = 1u;                      // the exact fallback answers
                    ctl[3] = 1;
                    return -1;
                }
            }
            const long long s_ = (long long)*reinterpret_cast<volatile int*>(&wqb[b_ & 7]) + o_;
            return s_ < nsteps ? s_ : -1;
        };
        long long step = take();
        unsigned st_push = 0, st_spin = 0, tm_a = 0, tm_b = 0, tm_c = 0;
        const bool timed = a.stats != nullptr;
        if (step >= 0) {
            uint4 T[SLOTS];
            auto frag_ptr = [&](long long st, int rt) {
                return reinterpret_cast<const char*>(a.dbc) + st * (32 * ROWB) + (g >> 1) * 1024 + ((g & 1) * 32 + rt * 16 + col) * 16;
            };
            uint4 M[4];
            auto load_meta = [&](long long st, uint4* m) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const uint4* mp = reinterpret_cast<const uint4*>(a.rmeta + st * 32 + rt * 16 + 4 * g);
                    m[2 * rt] = mp[0];
                    m[2 * rt + 1] = mp[1];
                }
            };
            {
                const char* p0 = frag_ptr(step, 0);
                const char* p1 = frag_ptr(step, 1);
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    T[s_] = *reinterpret_cast<const uint4*>(p0 + 2048 * s_);
                    T[KS + s_] = *reinterpret_cast<const uint4*>(p1 + 2048 * s_);
                }
                load_meta(step, M);
            }
            // thresholds: every step reads the WORKGROUP's copy of the keys in LDS; scanner wave 0 refreshes that copy from the
            // published keys every 4th step (request in one step, LDS update in the next), re-scoring waves raise it when they
            // raise a key. (All 1024 scanner waves re-reading the published keys - two cache lines, agent scope: served by the
            // fabric, not the XCD's L2 - every step made those two lines the scan's bottleneck: 7.5 us per step, 2.3 ms per scan.)
            unsigned tk[QG], tkn[QG];
#pragma unroll
            for (int qg = 0; qg < QG; ++qg) tkn[qg] = 0u;
            int it = 0;
            while (true) {
#pragma unroll
                for (int qg = 0; qg < QG; ++qg) tk[qg] = active[qg] ? *reinterpret_cast<volatile unsigned*>(ltau + qg * 16 + col) : 0u;
                if (wave == 0) {
                    if ((it & 3) == 0) {
#pragma unroll
                        for (int qg = 0; qg < QG; ++qg)
                            tkn[qg] = active[qg] ? __hip_atomic_load(a.tau_key + qg * 16 + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    } else if ((it & 3) == 1) {
#pragma unroll
                        for (int qg = 0; qg < QG; ++qg)
                            if (g == 0 && active[qg] && tkn[qg] > tk[qg]) atomicMax(ltau + qg * 16 + col, tkn[qg]);
                    }
                }
                ++it;
                const long long nxt = take();
                const bool has_next = nxt >= 0;
                const char* pn[2] = {frag_ptr(has_next ? nxt : step, 0), frag_ptr(has_next ? nxt : step, 1)};
                uint4 MN[4];
                load_meta(has_next ? nxt : step, MN);

                i32x4 acc[2][QG];
#pragma unroll
                for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg) acc[rt][qg] = i32x4{0, 0, 0, 0};
                uint4 B[2][QG];
                auto load_b = [&](int s_, uint4* b) {
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg) b[qg] = qimg[(qg * KS + s_) * 64 + lane];
                };
                const unsigned long long tk0 = timed ? wall_clock64() : 0ull;
                __builtin_amdgcn_sched_barrier(0);
                load_b(0, B[0]);
                // one query fragment from LDS feeds BOTH 16-row tiles (the LDS pipe is the scanner's second bound: read per
                // MFMA, 4 scanner waves keep it ~70 % busy)
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    if (s_ + 1 < KS) load_b(s_ + 1, B[(s_ + 1) & 1]);
                    const uint4* b = B[s_ & 1];
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                        for (int qg = 0; qg < QG; ++qg)
                            acc[rt][qg] = __builtin_amdgcn_mfma_i32_16x16x64_i8(__builtin_bit_cast(i32x4, T[rt * KS + s_]),
                                                                                __builtin_bit_cast(i32x4, b[qg]), acc[rt][qg], 0, 0, 0);
                    T[s_] = *reinterpret_cast<const uint4*>(pn[0] + 2048 * s_);
                    T[KS + s_] = *reinterpret_cast<const uint4*>(pn[1] + 2048 * s_);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, QG, 0);
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    if (s_ + 1 < KS) __builtin_amdgcn_sched_group_barrier(0x100, QG, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 2 * QG, 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                unsigned long long tk1 = 0ull;
                if (timed) { asm volatile("" :: "v"(acc[1][QG - 1][0])); tk1 = wall_clock64(); tm_a += (unsigned)(tk1 - tk0); }

#pragma unroll
                for (int qg = 0; qg < QG; ++qg) tau[qg] = active[qg] ? fkey_inv(tk[qg]) : INFINITY;
                float val[2][QG][4];
                int total = 0;
                const int lim = (int)(a.nrows - step * 32);           // rows of this step that exist (32 except in the last block)
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) {
                    const float sr[4] = {__uint_as_float(M[2 * rt].x), __uint_as_float(M[2 * rt].z), __uint_as_float(M[2 * rt + 1].x),
                                         __uint_as_float(M[2 * rt + 1].z)};
                    const float ar[4] = {__uint_as_float(M[2 * rt].y), __uint_as_float(M[2 * rt].w), __uint_as_float(M[2 * rt + 1].y),
                                         __uint_as_float(M[2 * rt + 1].w)};
#pragma unroll
                    for (int qg = 0; qg < QG; ++qg)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            val[rt][qg][r] = fmaf((float)acc[rt][qg][r], sr[r], ar[r] * yt[qg]);
                            if (lim < 32 && rt * 16 + 4 * g + r >= lim) val[rt][qg][r] = __uint_as_float(0x7fc00000u);   // NaN: passes no test
                            total += __popcll(__ballot(val[rt][qg][r] >= tau[qg]));
                        }
                }
                if (a.abl == 1) total = 0;
                unsigned long long tk2 = 0ull;
                if (timed) { asm volatile("" :: "s"(total)); tk2 = wall_clock64(); tm_b += (unsigned)(tk2 - tk1); }
                if (total) {
                    // ONE LDS atomic per step reserves `total` ring slots; the pairs go in by ballot prefix (scalar masks)
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&ctl[0], total);
                    base = __builtin_amdgcn_readfirstlane(base);
                    // room: re-scoring waves release their 64-slot batches IN ORDER (ctl[4] = batches released), so every slot
                    // below 64 * ctl[4] is free again. One LDS read per step instead of one read-and-wait per pushed pair.
                    bool room = true;
                    {
                        int spins = 0;
                        while (base + total > 64 * *reinterpret_cast<volatile int*>(&ctl[4]) + LIVE_QN) {
                            __builtin_amdgcn_s_sleep(8);
                            ++st_spin;
                            if (*reinterpret_cast<volatile int*>(&ctl[3]) || ++spins > LIVE_SPIN_LIMIT) {
                                *a.overflow = 1u;          // somebody gave up waiting: the exact fallback answers
                                ctl[3] = 1;
                                room = false;
                                break;
                            }
                        }
                    }
                    if (room) {
                        const unsigned row0 = (unsigned)(step * 32) + 4 * g;
#pragma unroll
                        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
                            for (int qg = 0; qg < QG; ++qg)
#pragma unroll
                                for (int r = 0; r < 4; ++r) {
                                    const unsigned long long m_ = __ballot(val[rt][qg][r] >= tau[qg]);
                                    if (m_) {
                                        if ((m_ >> lane) & 1ull)
                                            ring[(base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m_ >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m_, 0u))) &
                                                 (LIVE_QN - 1)] = make_uint2(__float_as_uint(val[rt][qg][r]),
                                                                             ((unsigned)(qg * 16 + col) << LIVE_ROW_BITS) | (row0 + rt * 16 + r));
                                        base += __popcll(m_);
                                    }
                                }
                        st_push += (unsigned)total;
                    }
                }
                if (timed) { wave_lds_sync(); tm_c += (unsigned)(wall_clock64() - tk2); }
                if (!has_next) break;
                step = nxt;
#pragma unroll
                for (int i = 0; i < 4; ++i) M[i] = MN[i];
            }
        }
        wave_lds_sync();                                   // this wave's ring writes are done (LDS is in order per wave)
        if (lane == 0) atomicAdd(&ctl[2], 1);
        if (a.stats) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) st_spin += __shfl_xor(st_spin, o);
            st_spin >>= 6;
            if (lane == 0) {
                atomicAdd(a.stats + 0, st_push); atomicAdd(a.stats + 1, st_spin);
                const unsigned dt = (unsigned)(wall_clock64() - t_begin);
                atomicMax(a.stats + 8, dt); atomicAdd(a.stats + 10, dt >> 10);
                atomicAdd(a.stats + 15, 1u);
                atomicAdd(a.stats + 12, tm_a >> 4); atomicAdd(a.stats + 13, tm_b >> 4); atomicAdd(a.stats + 14, tm_c >> 4);
            }
        }
        return;
    }

    // ---------------------------------------------------------------------- re-scoring wave
    // Claims 64 ring slots at a time, drops the pairs whose upper bound the threshold has passed since, and re-scores the rest
    // EIGHT pairs per round: 8 lanes per pair request the pair's whole row and query vector (2 x 2 KB, 32 loads of 16 B per
    // lane, ALL in flight together: one memory round trip per round), the wave transposes them through LDS in two halves and
    // lanes 0..7 walk one pair each through the score-order fmaf chain (t, c, g). (Earlier forms: per-lane gathers kept the
    // CU's address unit busy; 64 pairs per batch in 16 dependent 128-byte stages took ~100 us per batch - the scan waited
    // 340 us for the re-scoring waves to drain.)
    volatile int* vctl = ctl;
    unsigned st_pop = 0, st_stale = 0, st_resc = 0, st_ins = 0, st_batch = 0, st_rise = 0, st_phase = 0;
    const int pslot = lane >> 3, piece = lane & 7;
    for (;;) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&ctl[1], 64);
        base = __shfl(base, 0);
        {
            const int dn = vctl[2];                       // read `done` BEFORE `tail`: a finished scanner's reservations are all in
            const int tail = vctl[0];
            if ((dn == NSCAN && base >= tail) || (vctl[3] && dn == NSCAN)) break;   // nothing at or behind this index will ever come
        }
        const int idx = base + lane;
        uint2* sl = ring + (idx & (LIVE_QN - 1));
        bool pend = true;
        int spins = 0;
        while (__ballot(pend)) {
            // wait for this lane's entry - but never with filled lanes in hand for long: a scanner may be waiting for
            // exactly those slots (phased processing; see the deadlock note in DESIGN.md 4.1e)
            bool got = false;
            uint2 e = make_uint2(0u, LIVE_EMPTY);
            for (int poll = 0; poll < 64; ++poll) {
                if (pend && !got) {
                    e.y = *reinterpret_cast<volatile unsigned*>(&sl->y);
                    if (e.y != LIVE_EMPTY) {
                        e.x = *reinterpret_cast<volatile unsigned*>(&sl->x);
                        got = true;
                    } else {
                        const int dn = vctl[2];
                        const int tail = vctl[0];
                        if (dn == NSCAN && idx >= tail) pend = false;
                        else if (vctl[3] || ++spins > LIVE_SPIN_LIMIT) { *a.overflow = 1u; ctl[3] = 1; pend = false; }
                    }
                }
                const unsigned long long waiting = __ballot(pend && !got), have = __ballot(got);
                if (!waiting || (have && poll >= 32)) break;
                __builtin_amdgcn_s_sleep(4);
            }
            if (!__ballot(got)) continue;
            // ---- the lanes that hold a pair
            if (got) { *sl = make_uint2(0u, LIVE_EMPTY); pend = false; ++st_pop; }
            const unsigned qi = got ? (e.y >> LIVE_ROW_BITS) : 0u;
            const unsigned slot_ = e.y & ((1u << LIVE_ROW_BITS) - 1u);
            const unsigned row = (got && a.slot_rows) ? a.slot_rows[slot_] : slot_;       // the copy is a permutation of the rows
            bool live_ = got && a.abl != 2;
            bool scored = false;
            float myscore = 0.f;
            if (lane == 0) ++st_phase;
            while (true) {
                // (re-)check against the workgroup's threshold copy: it may have risen during the previous round
                if (live_ && !(__uint_as_float(e.x) >= fkey_inv(*reinterpret_cast<volatile unsigned*>(ltau + qi)))) { live_ = false; ++st_stale; }
                unsigned long long t_ = __ballot(live_);
                if (!t_) break;
                int mysrc = -1, mygrp = -1;
#pragma unroll
                for (int p_ = 0; p_ < 8; ++p_)
                    if (t_) {
                        const int s_ = __builtin_ctzll(t_);
                        if (pslot == p_) mysrc = s_;
                        if (lane == s_) { live_ = false; mygrp = p_; }          // taken by this round
                        t_ &= t_ - 1;
                    }
                if (lane == 0) ++st_batch;
                unsigned prow = __shfl(row, mysrc < 0 ? 0 : mysrc), pq = __shfl(qi, mysrc < 0 ? 0 : mysrc);
                if (mysrc < 0) { prow = 0u; pq = 0u; }          // idle lane groups re-read row 0 / query 0 (cache hits), stored nowhere
                f32x4 R[16], QV[16];
                {
                    const float* rp = a.db + (size_t)prow * E + piece * 4;
                    const float* qp = a.q + (size_t)pq * E + piece * 4;
#pragma unroll
                    for (int j = 0; j < 16; ++j) R[j] = *reinterpret_cast<const f32x4*>(rp + 32 * j);
#pragma unroll
                    for (int j = 0; j < 16; ++j) QV[j] = *reinterpret_cast<const f32x4*>(qp + 32 * j);
                }
                // the chain walks the pair's 8 lanes: a 16-float block lives in one quad (4 lanes x float4), element order
                // (c, g) = component-major across the quad's lanes, so the running sum hops lane g-1 -> g by a quad rotation
                // (DPP) before every fmaf; all lanes execute every step, the TRUE sum is the one travelling. Between blocks
                // the true sum is broadcast to the group (one ds_bpermute per 16 elements). No LDS staging.
                float acc = 0.f;
#pragma unroll
                for (int j = 0; j < 16; ++j)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        if (j || tt) acc = __shfl(acc, (lane & ~7) | (tt ? 3 : 7));
#pragma unroll
                        for (int c = 0; c < 4; ++c)
#pragma unroll
                            for (int g_ = 0; g_ < 4; ++g_) {
                                const float rot = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(acc), 0x93, 0xf, 0xf, true));
                                acc = __builtin_fmaf(R[j][c], QV[j][c], rot);
                            }
                    }
                // lane 7 of every group holds its pair's exact score: back to the lane the pair came from
                const float res = __shfl(acc, mygrp < 0 ? 0 : 8 * mygrp + 7);
                if (mygrp >= 0) { myscore = res; scored = true; ++st_resc; }
            }
            // ---- the phase's scores, all lanes at once: list + ladder histogram (the constants come from LDS; the only
            // dependent fabric round trips of a phase are the counters' returns and the ladder rows)
            const unsigned kt = lane < a.QA ? __hip_atomic_load(a.tau_key + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const unsigned kx = lane < a.QA ? __hip_atomic_load(a.tex_key + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            const bool ins = scored && myscore == myscore && myscore >= fkey_inv(*reinterpret_cast<volatile unsigned*>(ltex + qi));
            if (ins) {
                ++st_ins;
                const f32x4 cn = lcon[qi];
                float x = (myscore - cn.x) / cn.y;
                x = x < 0.f ? 0.f : (x > (float)(LIVE_NB - 1) ? (float)(LIVE_NB - 1) : x);
                int b = (int)x;
                if (b > 0 && !(myscore >= fmaf((float)b, cn.y, cn.x))) --b;        // the bucket's edge must not exceed the score
                if (myscore >= fmaf((float)b, cn.y, cn.x)) atomicAdd(&a.hist[qi * LIVE_NB + b], 1u);
                const unsigned pos = atomicAdd(&a.gcnt[qi], 1u);
                if ((long long)pos < a.cap) a.cand[(size_t)qi * a.cap + pos] = make_uint2(__float_as_uint(myscore), row);
                else *a.overflow = 1u;
            }
            if (lane < a.QA) { atomicMax(ltau + lane, kt); atomicMax(ltex + lane, kx); }
            // ladder: for up to four of the queries that just received a row, the highest edge with >= K rows at or above it
            unsigned long long todo = __ballot(ins);
            unsigned lq[4];
            unsigned hv[4];
            int nl = 0;
#pragma unroll
            for (int r_ = 0; r_ < 4; ++r_) {
                lq[r_] = 0u;
                hv[r_] = 0u;
                if (todo) {
                    const int src = __builtin_ctzll(todo);
                    lq[r_] = __shfl(qi, src);
                    todo &= ~__ballot(ins && qi == lq[r_]);
                    hv[r_] = lane < LIVE_NB ? __hip_atomic_load(a.hist + lq[r_] * LIVE_NB + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                    nl = r_ + 1;
                }
            }
#pragma unroll
            for (int r_ = 0; r_ < 4; ++r_)
                if (r_ < nl) {
                    unsigned v = hv[r_];
#pragma unroll
                    for (int o = 1; o < LIVE_NB; o <<= 1) {
                        const unsigned t2 = __shfl_down(v, o);
                        if (lane + o < LIVE_NB) v += t2;
                    }
                    const unsigned long long okm = __ballot(lane < LIVE_NB && v >= (unsigned)a.K);
                    if (okm && lane == 0) {
                        const int bmax = 63 - __builtin_clzll(okm);
                        const f32x4 cn = lcon[lq[r_]];
                        const float tnew = fmaf((float)bmax, cn.y, cn.x);
                        const unsigned nk = fkey((tnew - cn.w) * cn.z), nx = fkey(tnew);
                        if (nx > *reinterpret_cast<volatile unsigned*>(ltex + lq[r_])) {
                            ++st_rise;
                            atomicMax(a.tex_key + lq[r_], nx);
                            atomicMax(a.tau_key + lq[r_], nk);
                            atomicMax(ltex + lq[r_], nx);
                            atomicMax(ltau + lq[r_], nk);
                        }
                    }
                }
        }
        // release the batch in claim order: scanners may then reuse every slot below 64 * ctl[4]
        if (lane == 0) {
            int sp = 0;
            while (vctl[4] != (base >> 6)) {
                __builtin_amdgcn_s_sleep(2);
                if (vctl[3] || ++sp > LIVE_SPIN_LIMIT) { *a.overflow = 1u; ctl[3] = 1; break; }
            }
            ctl[4] = (base >> 6) + 1;
        }
    }
    if (a.stats) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            st_pop += __shfl_xor(st_pop, o); st_stale += __shfl_xor(st_stale, o); st_resc += __shfl_xor(st_resc, o);
            st_ins += __shfl_xor(st_ins, o); st_batch += __shfl_xor(st_batch, o); st_rise += __shfl_xor(st_rise, o);
        }
        if (lane == 0) {
            atomicAdd(a.stats + 2, st_pop); atomicAdd(a.stats + 3, st_stale); atomicAdd(a.stats + 4, st_resc);
            atomicAdd(a.stats + 5, st_ins); atomicAdd(a.stats + 6, st_batch); atomicAdd(a.stats + 7, st_rise);
            atomicMax(a.stats + 9, (unsigned)(wall_clock64() - t_begin));
            atomicAdd(a.stats + 11, st_phase);
        }
    }
}

#endif  // CLIPMI_DEV

// =================================================================================================
// Wide coarse pass (int8 copy only): ONE stream of the copy for up to WIDE_MAX_Q = 1024 queries of one search call
// (reference: query-index.py:111 is one index.search call; SURVEY.md §8d "DB streamed once per batch").
//
// The 64-query pass above is HBM-bound with the matrix cores 17 % busy; a search of Q >> 64 queries as Q / 64 such
// passes streams the copy Q / 64 times. Here the queries are cut into tiles of <= 256 (a 128-KiB int8 image in LDS)
// and every workgroup keeps ONE tile resident for the whole launch. Workgroups with the same `row lane` and different
// tiles walk the same rows of the copy in the same order; they sit on one XCD (blockIdx % 8 labels the XCD's share of
// the grid - speed only), so the copy crosses HBM about once per launch and the other tiles' reads are L2 / Infinity
// Cache hits. Per 32-row step a wave holds the rows' fragments in registers (16 KiB) and walks the tile's 64-query sets:
// per set and k-step 4 query fragments from LDS feed 8 MFMAs (both 16-row tiles: half an LDS read per MFMA, so the LDS
// pipe is at 50 % when the matrix pipe is full); the fragments' registers are refilled with the next step's rows during
// the LAST set. Two waves per SIMD (512 threads): one wave's compare epilogue and its wait for the next rows run under
// the other's MFMAs. The pass is matrix-bound: 2 N Q 512 int8 operations per launch.
// Survivors go to the same per-query lists as the 64-query pass's (entries appended by ballot prefix, no LDS atomics).
// =================================================================================================
// meta of the int8 copy: row entries for N rounded up to 32 rows (+32), then one (scale, largest error norm) pair per block (+1)
inline size_t i8_row_meta_entries(int64_t N) { return (size_t)((N + 31) / 32 * 32 + 32); }
inline size_t i8_block_meta_entries(int64_t N) { return (size_t)((N + 31) / 32 + 1); }
// ... and behind the block meta one u32 per slot: the row it holds (quantize_rows_i8_kernel)
inline const unsigned* i8_slot_rows(const float2* rmeta, int64_t N) {
    return reinterpret_cast<const unsigned*>(rmeta + i8_row_meta_entries(N) + i8_block_meta_entries(N));
}

constexpr int WIDE_MAX_Q = 1024;                 // queries of one wide launch (4 tiles of 256)
constexpr int WIDE_TILE_SETS = 4;                // 64-query sets per tile: 4 x 32 KiB of image
constexpr int WIDE_FLUSH = 128;                  // pending pairs per wave that trigger a flush
constexpr int WIDE_LIST = WIDE_FLUSH + 256;      // a (row tile, query group) block appends <= 256 behind a pending flush
constexpr size_t WIDE_WAVE_BYTES = (size_t)WIDE_LIST * 8;
constexpr long long WIDE_CAP = 1ll << 15;        // candidate slots per query of a wide pass (overflow arms the exact fallback)

struct WideArgs {
    const signed char* dbc;      // int8 copy, 32-row blocks of [16][64][16 B] (quantize_rows_i8_kernel)
    const float2* rmeta;         // per row (block scale, error norm), padded to 32 rows
    const float2* bmeta;         // per 32-row block (scale, largest error norm)
    const float* qmeta;          // [4][qs]
    const uint4* qimage;         // all queries' image, 64-query sets of 32 KiB (coarse_prep_kernel, wide form)
    long long row0, nrows;       // rows [row0, row0 + nrows), row0 % 32 == 0
    int Q, qs;
    int nqt, spt;                // query tiles, 64-query sets per tile (<= WIDE_TILE_SETS)
    uint2* cand;                 // [Q][cap]
    unsigned* gcnt;              // [qs]
    long long cap;
    unsigned* overflow;
    int map_mode, pf_mode;       // development knobs (CLIPMI_WIDE_MAP / CLIPMI_WIDE_PF)
};

typedef int i32x16 __attribute__((ext_vector_type(16)));

__device__ __noinline__ void wide_flush(const uint2* list, int n, int qbase, unsigned* gcnt, uint2* cand, long long cap,
                                        unsigned* overflow) {
    const int lane = threadIdx.x & 63;
    wave_lds_sync();
    for (int e = lane; e < n; e += 64) {
        const uint2 c = list[e];
        const unsigned q = (unsigned)qbase + c.x;
        const unsigned pos = atomicAdd(&gcnt[q], 1u);
        if ((long long)pos < cap) cand[(size_t)q * cap + pos] = make_uint2(CAND_SLOT, c.y);
        else *overflow = 1u;
    }
    wave_lds_sync();
}

template <int WAVES, int ABL = 0>      // ABL: development ablations (1 no compare, 2 few LDS reads, 3 no row loads)
__global__ void __launch_bounds__(WAVES * 64) scan_coarse_wide_kernel(WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = 16;                                    // k-steps of 32 bytes
    constexpr int SET_ENTRIES = 2 * KS * 64;                  // 16-byte entries of one 64-query set's image (32 KiB)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;                   // accumulator column (query of a 32-group), row half

    // workgroup -> (query tile, row lane): blocks b, b + 8, ... share an XCD (observed dispatch; speed only)
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, jb = gridDim.x >> 3;
    const int rlx = jb / a.nqt;                               // row lanes per XCD label
    if (j >= rlx * a.nqt) return;
    const int qt = j % a.nqt, nrl = rlx * 8;
    const int rl = a.map_mode == 1 ? x * rlx + (j / a.nqt) : (j / a.nqt) * 8 + x;
    const int set0 = qt * a.spt;
    int nsets = ((a.Q + 63) >> 6) - set0;
    nsets = nsets > a.spt ? a.spt : nsets;
    if (nsets < 1) return;
    const int qbase = set0 * 64;

    uint4* qimg = reinterpret_cast<uint4*>(smem);
    {
        const uint4* src = a.qimage + (size_t)set0 * SET_ENTRIES;
        const int ne = nsets * SET_ENTRIES;                   // multiple of 2048
        for (int base = tid; base < ne; base += 8 * WAVES * 64) {
            uint4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = base + i * WAVES * 64 < ne ? src[base + i * WAVES * 64] : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (base + i * WAVES * 64 < ne) qimg[base + i * WAVES * 64] = v[i];
        }
    }
    float* ltau = reinterpret_cast<float*>(smem + (size_t)a.spt * SET_ENTRIES * 16);       // [256] scaled thresholds
    float* lyt = ltau + 64 * WIDE_TILE_SETS;                                               // [256] 1.001 ||y|| / t_q
    for (int i = tid; i < 64 * WIDE_TILE_SETS; i += WAVES * 64) {
        const bool act = i < nsets * 64 && qbase + i < a.Q;
        ltau[i] = act ? a.qmeta[2 * a.qs + qbase + i] : INFINITY;
        lyt[i] = act ? a.qmeta[a.qs + qbase + i] : 0.f;
    }
    uint2* list = reinterpret_cast<uint2*>(reinterpret_cast<char*>(lyt + 64 * WIDE_TILE_SETS) + (size_t)wave * WIDE_WAVE_BYTES);
    __syncthreads();

    // block indices fit 32 bits (N < 2^32 rows) and are wave-uniform: kept in scalar registers, so the block meta is a
    // scalar load (a vector load here put a vmcnt(0) at the head of the group loop - behind the 16 KB prefetch) and the row
    // loads are scalar base + one lane offset
    const int blk0 = (int)(a.row0 >> 5);
    const int nblk = blk0 + (int)((a.nrows + 31) >> 5);
    const int tw = nrl * WAVES;
    const unsigned last_row32 = (unsigned)(a.row0 + a.nrows - 1);
    int npend = 0;                                            // wave-uniform: pairs pending in `list`

    int blk = __builtin_amdgcn_readfirstlane(blk0 + rl * WAVES + wave);
    if (blk < nblk) {
        uint4 Ta[KS], Tb[KS];                                 // this block's rows / the next block's (in flight for a whole step)
        // rows are loaded as buffer loads: scalar resource (rebased per block, so 32-bit offsets suffice for any copy size),
        // ONE lane-offset register, scalar k-step offsets - 64-bit per-lane addresses cost 26 registers here
        const unsigned lane16 = (unsigned)lane * 16u;
        auto load_block = [&](int b_, uint4 (&dst)[KS]) {
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<signed char*>(a.dbc) + (size_t)b_ * (32 * 512), 0, 32 * 512, 0x00020000);
#pragma unroll
            for (int s_ = 0; s_ < KS; ++s_)
                dst[s_] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, s_ * 1024, (CLIPMI_NT_MASK & 2) ? 2 : 0));
        };
        load_block(blk, Ta);
        // the block's (scale, largest error norm): fetched with the rows, one block ahead (a load at the head of a block
        // would be waited for behind that block's own 16 KB prefetch: vmcnt counts in order)
        float2 bm = a.bmeta[blk], bmn = bm;
        // One 32-row block against the tile's 32-query groups. The next block's 16 KB are requested during group `pf_g`
        // and land under the remaining MFMAs: group 0 in the tile-0 workgroup of a row lane, halfway through the step in
        // the other tiles' workgroups of the same row lane (same rows, same order, same XCD; a measured wash - 41 % of the
        // L2 requests hit either way - kept because it spreads the four workgroups' HBM bursts).
        // The compare of group G - 1 is issued between the MFMAs of group G (two accumulators): alone it left the matrix
        // pipe idle for the accumulator drain, a dependent VALU chain and a branch per group (19 % of the launch, by ablation).
        // Returns false after the wave's last block.
        const int ng = 2 * nsets;
        const int pf_g = (qt == 0 || a.pf_mode == 1) ? 0 : a.pf_mode == 2 ? ng - 2 : a.pf_mode == 3 ? qt * ng / a.nqt : ng / 2;
#ifndef CLIPMI_WIDE_BR
#define CLIPMI_WIDE_BR 4
#endif
        constexpr int BR = CLIPMI_WIDE_BR;                     // ring of query fragments: LDS reads run BR MFMAs ahead
        auto run_block = [&](uint4 (&T)[KS], uint4 (&TN)[KS]) -> bool {
            const int nxt = __builtin_amdgcn_readfirstlane(blk + tw);
            const bool has_next = nxt < nblk;
            const float inv_s = 1.0f / bm.x;
            uint4 B[BR];                                       // ring of query fragments, continuous over the groups
            {
                const uint4* img0 = qimg + lane;
#pragma unroll
                for (int p_ = 0; p_ < BR; ++p_) B[p_] = img0[p_ * 64];
            }
            // The exact test of a (row, query) pair is  D s + a_r Y_q >= T_q  (T_q = (tau - margin) / t_q, Y_q = 1.001 ||y|| / t_q;
            // s the block's scale). Integer pre-test, one per lane (= query): no row of the block can pass unless
            // D >= (T_q - amax_block Y_q) / s, rounded down with slack for the f32 roundings on both sides (|D| < 2^23 is exact
            // in f32; each side rounds a few times at ~1e-7 relative) - so the lane's largest D decides, and the per-row test
            // runs only where it can succeed. +inf thresholds (padding queries: any Q that is not a multiple of 32) are "never" by
            // an explicit test - inf - inf below would be NaN = "always", and every block would then run the 16-ballot append
            // path for its padding lanes; an all-zero block (s = 0: inv_s = inf) clamps to +-1e9 by its sign;
            // -inf (no threshold yet) and NaN stay "always".
            auto pretest = [&](const i32x16& acc, float tq, float yq) -> bool {
                if (tq == INFINITY) return false;
                float xq = fmaf(-bm.y, yq, tq) * inv_s;
                xq = fminf(fmaxf(xq, -1.0e9f), 1.0e9f);                   // before the slack: no inf - inf; fmaxf(NaN, c) = c: NaN -> always
                xq = xq - 2.0f - fabsf(xq) * 2e-6f;
                const int dmin = (int)floorf(xq);
                auto max3 = [](int a_, int b_, int c_) { const int m_ = a_ > b_ ? a_ : b_; return m_ > c_ ? m_ : c_; };
                const int m0 = max3(acc[0], acc[1], acc[2]), m1 = max3(acc[3], acc[4], acc[5]), m2 = max3(acc[6], acc[7], acc[8]);
                const int m3 = max3(acc[9], acc[10], acc[11]), m4 = max3(acc[12], acc[13], acc[14]);
                const int ma = max3(m0, m1, m2), mb = max3(m3, m4, acc[15]);              // three levels, not a chain of eight
                return (ma > mb ? ma : mb) >= dmin;
            };
            // per-row test with the block's largest error norm in place of the row's own (a few per cent looser, a superset
            // all the same): no per-row meta loads here - they would queue behind the next block's 16 KB
            auto append = [&](const i32x16& acc, int G, float tq, float yq) {
                const unsigned rbase = (unsigned)blk * 32u + 4u * (unsigned)h;      // rows fit 32 bits (N < 2^32 - 1)
                const float ay = bm.y * yq;
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const unsigned row = rbase + (unsigned)((i & 3) + 8 * (i >> 2));
                    const float val = fmaf((float)acc[i], bm.x, ay);
                    const bool pass = (val >= tq) && (row <= last_row32);
                    const unsigned long long m = __ballot(pass);
                    if (m) {
                        const int pos = npend + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32),
                                                                          __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (pass) list[pos] = make_uint2((unsigned)(G * 32 + n), row);
                        npend += __builtin_popcountll(m);
                    }
                    if ((i & 3) == 3 && npend > WIDE_FLUSH) {              // <= 256 appends between two tests
                        wide_flush(list, npend, qbase, a.gcnt, a.cand, a.cap, a.overflow);
                        npend = 0;
                    }
                }
            };
            // group G's 16 MFMAs into `acc`; group G - 1's pre-test (accumulator `accp`) between them
            auto run_group = [&](int G, i32x16& acc, const i32x16& accp, float& tq, float& yq, float tqp, float yqp) {
                if (G == pf_g) {
                    if (ABL == 3) {
#pragma unroll
                        for (int s_ = 0; s_ < KS; ++s_) TN[s_] = T[s_];
                    } else {
                        load_block(has_next ? nxt : blk, TN);
                    }
                    bmn = a.bmeta[has_next ? nxt : blk];
                }
                const uint4* img = qimg + (size_t)G * (KS * 64) + lane;
                const uint4* imgn = qimg + (size_t)(G + 1 < ng ? G + 1 : G) * (KS * 64) + lane;   // the next group's head
                tq = ltau[G * 32 + n];                                     // read ahead of the fragments: no wait of their own
                yq = lyt[G * 32 + n];
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = 0;
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, T[s_]),
                                                                __builtin_bit_cast(i32x4, B[s_ % BR]), acc, 0, 0, 0);
                    if (ABL != 2) B[s_ % BR] = s_ + BR < KS ? img[(s_ + BR) * 64] : imgn[(s_ + BR - KS) * 64];
                }
                bool hitp = false;
                if (ABL != 1) hitp = pretest(accp, tqp, yqp);
#pragma unroll
                for (int s_ = 0; s_ < KS; ++s_) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                      // MFMA of k-step s
                    if (ABL != 2) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);        // DS read: the fragment BR k-steps on
                    if (ABL != 1) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);        // VALU: two of the pre-test's ~30
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ABL == 1) asm volatile("" :: "v"(accp), "v"(tqp), "v"(yqp));
                if (G > 0 && __ballot(hitp)) append(accp, G - 1, tqp, yqp);
            };
            i32x16 accA, accB;
            float tqA = 0.f, yqA = 0.f, tqB = INFINITY, yqB = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) accB[i] = 0;
            for (int G = 0; G < ng; G += 2) {
                run_group(G, accA, accB, tqA, yqA, tqB, yqB);
                run_group(G + 1, accB, accA, tqB, yqB, tqA, yqA);
            }
            if (ABL != 1 && __ballot(pretest(accB, tqB, yqB))) append(accB, ng - 1, tqB, yqB);
            blk = nxt;
            bm = bmn;
            return has_next;
        };
        while (run_block(Ta, Tb) && run_block(Tb, Ta)) {}
    }
    // final publication, aggregated per block (one returning global atomic per (block, query))
    __syncthreads();
    constexpr int QMAXC = 64 * WIDE_TILE_SETS;
    unsigned* hist = reinterpret_cast<unsigned*>(smem);              // the image is dead now
    unsigned* gbase = hist + QMAXC;
    unsigned* hoff = hist + 2 * QMAXC;
    for (int i = tid; i < QMAXC; i += WAVES * 64) { hist[i] = 0; hoff[i] = 0; }
    __syncthreads();
    for (int e = lane; e < npend; e += 64) atomicAdd(&hist[list[e].x], 1u);
    __syncthreads();
    for (int i = tid; i < QMAXC; i += WAVES * 64) gbase[i] = hist[i] ? atomicAdd(&a.gcnt[qbase + i], hist[i]) : 0u;
    __syncthreads();
    for (int e = lane; e < npend; e += 64) {
        const uint2 c = list[e];
        const unsigned pos = gbase[c.x] + atomicAdd(&hoff[c.x], 1u);
        if ((long long)pos < a.cap) a.cand[(size_t)(qbase + c.x) * a.cap + pos] = make_uint2(CAND_SLOT, c.y);
        else *a.overflow = 1u;
    }
}

struct Plan {
    int C, waves, QA, QG, grid, grid_sample, wave_bytes, stage;
    size_t lds_scan, lds_sel;
    long long NL, NL_sample, cap, sample_rows;
    bool sample;
};

// Shared by workspace sizing and launch so both always agree.
// lds_limit: what a workgroup of the exact scan or a select may take (the coarse path asks for COARSE_SIDE_LDS so its
// side kernels fit on a CU beside a coarse scan workgroup of another batch in flight)
bool make_plan(long long N, int E, int Q, int K, Plan& p, int lds_limit = LDS_LIMIT) {
    if (N < 0 || Q < 1 || K < 1 || (E != 512 && E != 768)) return false;
    p.C = (int)align_up((size_t)K + 16, 16);
    p.waves = 0;
    // prefer 4 waves per block and as many queries per pass as LDS allows (32, else 16, else fewer)
    for (int w = 4; w >= 1 && p.waves == 0; w >>= 1)
        for (int qg = (Q > 16 ? 2 : 1); qg >= 1; --qg) {
            const int qimg = qg * (E / 16) * 1024;
            const long long per_wave = (lds_limit - qimg) / w - 256;
            const long long fit = per_wave / ((long long)p.C * 8) - 1;     // queries that fit beside scratch
            const int qwant = Q < 16 * qg ? Q : 16 * qg;
            if (fit >= qwant || (qg == 1 && fit >= 1)) {
                p.waves = w; p.QG = qg; p.QA = (int)(fit < qwant ? fit : qwant);
                break;
            }
        }
    if (p.waves == 0) return false;
    if (SEL_FIXED + (long long)K * 8 > lds_limit) return false;
    p.stage = sel_stage_entries(K, lds_limit);
    p.lds_sel = SEL_FIXED + (size_t)K * 8 + (size_t)p.stage * 8;
    if (p.lds_sel > (size_t)lds_limit) return false;
    const int qimg = p.QG * (E / 16) * 1024;
    p.wave_bytes = (p.QA + 1) * p.C * 8 + 256;
    p.lds_scan = (size_t)qimg + (size_t)p.waves * p.wave_bytes;
    const long long ntiles = (N + 15) / 16;
    long long grid = (ntiles + p.waves - 1) / p.waves;
    if (grid > NUM_CU) grid = NUM_CU;            // one block per CU: the ring keeps ~100 KB in flight per CU
    if (grid < 1) grid = 1;
    p.grid = (int)grid;
    p.NL = grid * p.waves;
    // threshold pre-pass over the first S rows: S ~ sqrt(N*K) balances the two select passes
    p.sample = N >= SAMPLE_MIN_N;
    long long S = 4096;
    while (S < 65536 && S * S < N * (long long)K) S *= 2;
    p.sample_rows = S;
    p.grid_sample = (int)((S / 16 + p.waves - 1) / p.waves);
    p.NL_sample = (long long)p.grid_sample * p.waves;
    const long long nl = p.NL > p.NL_sample ? p.NL : p.NL_sample;
    p.cap = nl * p.C;
    return true;
}

int opt_in_lds(const void* fn, size_t bytes) {
    if (bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e != hipSuccess) return set_err(CLIPMI_EHIP, "hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
    }
    return 0;
}

template <int E, bool PREPASS, int QG>
int launch_scan_t(const ScanArgs& a, int grid, int waves, size_t lds, hipStream_t st, hipEvent_t* ev, int ny) {
    const void* fn = (const void*)scan_topk_f32_kernel<E, PREPASS, QG>;
    if (int rc = opt_in_lds(fn, lds)) return rc;
    // measurement: plain event records around the launch (for a millisecond-scale kernel they agree
    // with rocprofv3's dispatch time to <1 %; hipExtLaunchKernel's start/stop events read ~7 % long here)
    if (ev) (void)hipEventRecord(ev[0], st);
    hipLaunchKernelGGL((scan_topk_f32_kernel<E, PREPASS, QG>), dim3(grid, ny), dim3(waves * 64), lds, st, a);
    if (ev) (void)hipEventRecord(ev[1], st);
    CLIPMI_CHECK_LAUNCH("scan_topk_f32_kernel");
    return 0;
}

template <bool PREPASS>
int launch_scan(int E, int QG, const ScanArgs& a, int grid, int waves, size_t lds, hipStream_t st, hipEvent_t* ev = nullptr,
                int ny = 1) {
    if (E == 512) return QG == 2 ? launch_scan_t<512, PREPASS, 2>(a, grid, waves, lds, st, ev, ny)
                                 : launch_scan_t<512, PREPASS, 1>(a, grid, waves, lds, st, ev, ny);
    return QG == 2 ? launch_scan_t<768, PREPASS, 2>(a, grid, waves, lds, st, ev, ny)
                   : launch_scan_t<768, PREPASS, 1>(a, grid, waves, lds, st, ev, ny);
}

}  // namespace
}  // namespace clipmi

using namespace clipmi;

extern "C" size_t clipmi_topk_ip_workspace_bytes(int64_t N, int E, int Q, int K) {
    Plan p;
    if (!make_plan(N, E, Q, K, p)) {
        set_err(CLIPMI_EINVAL, "topk_ip: unsupported N=%lld E=%d Q=%d K=%d (E in {512,768}, 1 <= K <= ~8000)",
                (long long)N, E, Q, K);
        return 0;
    }
    return align_up((size_t)p.QA * p.cap * sizeof(uint2), 256) + 256 /*gcnt*/ + 256 /*thr*/ + 256;
}

static int topk_ip_impl(const void* db_dev, int db_dtype, int64_t N, int E, const float* q_dev, int Q, int K,
                        int64_t id_base, float* out_score_dev, int64_t* out_id_dev, void* ws_dev, size_t ws_bytes,
                        void* stream, hipEvent_t* scan_ev) {
    if (db_dtype != CLIPMI_F32) return set_err(CLIPMI_EUNSUPPORTED, "topk_ip: db_dtype %d (only CLIPMI_F32)", db_dtype);
    if (!q_dev || !out_score_dev || !out_id_dev) return set_err(CLIPMI_EINVAL, "topk_ip: NULL pointer");
    if (N >= (1ll << 32) - 1) return set_err(CLIPMI_EINVAL, "topk_ip: N=%lld exceeds 2^32-2 rows per shard", (long long)N);
    Plan p;
    if (!make_plan(N, E, Q, K, p))
        return set_err(CLIPMI_EINVAL, "topk_ip: unsupported N=%lld E=%d Q=%d K=%d", (long long)N, E, Q, K);
    hipStream_t st = as_stream(stream);
    if (N == 0) {
        const long long n = (long long)Q * K;
        hipLaunchKernelGGL(fill_empty_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, out_score_dev,
                           (long long*)out_id_dev, n);
        CLIPMI_CHECK_LAUNCH("fill_empty_kernel");
        return 0;
    }
    if (!db_dev || !ws_dev) return set_err(CLIPMI_EINVAL, "topk_ip: NULL db or workspace");
    const size_t need_ws = clipmi_topk_ip_workspace_bytes(N, E, Q, K);
    if (ws_bytes < need_ws) return set_err(CLIPMI_EWORKSPACE, "topk_ip: workspace %zu < %zu", ws_bytes, need_ws);
    Arena ar(ws_dev, ws_bytes);
    uint2* cand = ar.take<uint2>((size_t)p.QA * p.cap);
    unsigned* gcnt = ar.take<unsigned>(32);
    float* thr0 = ar.take<float>(32);

    if (int rc = opt_in_lds((const void*)select_topk_kernel, p.lds_sel)) return rc;

    // candidate counters: zeroed once here; every select_topk_kernel re-zeroes the ones it consumed
    if (hipMemsetAsync(gcnt, 0, 128, st) != hipSuccess) return set_err(CLIPMI_EHIP, "hipMemsetAsync");
    for (int q0 = 0; q0 < Q; q0 += p.QA) {
        const int qa = (Q - q0) < p.QA ? (Q - q0) : p.QA;
        ScanArgs a;
        a.db = static_cast<const float*>(db_dev);
        a.q = q_dev + (size_t)q0 * E;
        a.QA = qa;
        a.K = K;
        a.C = p.C;
        a.wave_bytes = p.wave_bytes;
        a.cand = cand;
        a.gcnt = gcnt;
        a.cap = p.cap;
        a.thr_in = nullptr;
        a.run_if = nullptr;
        a.q_total = 0;
        if (p.sample) {
            // pre-pass: exact K-th best score of the first S rows = a valid lower bound for the
            // K-th best of all rows; the main pass then only buffers scores >= that bound
            a.nrows = p.sample_rows;
            int rc = launch_scan<true>(E, p.QG, a, p.grid_sample, p.waves, p.lds_scan, st);
            if (rc) return rc;
            hipLaunchKernelGGL(select_topk_kernel, dim3(qa), dim3(SEL_THREADS), p.lds_sel, st, cand, gcnt, p.cap, K,
                               (long long)0, (float*)nullptr, (long long*)nullptr, thr0, (const unsigned*)nullptr);
            CLIPMI_CHECK_LAUNCH("select_topk_kernel(sample)");
            a.thr_in = thr0;
        }
        a.nrows = N;
        int rc = launch_scan<false>(E, p.QG, a, p.grid, p.waves, p.lds_scan, st, scan_ev);
        if (rc) return rc;
        hipLaunchKernelGGL(select_topk_kernel, dim3(qa), dim3(SEL_THREADS), p.lds_sel, st, cand, gcnt, p.cap, K,
                           (long long)id_base, out_score_dev + (size_t)q0 * K, (long long*)out_id_dev + (size_t)q0 * K,
                           (float*)nullptr, (const unsigned*)nullptr);
        CLIPMI_CHECK_LAUNCH("select_topk_kernel");
    }
    return 0;
}

extern "C" int clipmi_topk_ip(const void* db_dev, int db_dtype, int64_t N, int E, const float* q_dev, int Q, int K,
                              int64_t id_base, float* out_score_dev, int64_t* out_id_dev, void* ws_dev,
                              size_t ws_bytes, void* stream) {
    return topk_ip_impl(db_dev, db_dtype, N, E, q_dev, Q, K, id_base, out_score_dev, out_id_dev, ws_dev, ws_bytes,
                        stream, nullptr);
}

namespace clipmi {
namespace {
constexpr long long COARSE_CAP = 1ll << 18;      // candidate slots per query of the coarse pass
constexpr int COARSE_SIDE_LDS = 100 * 1024;      // LDS of every kernel of the coarse path other than the scans: a 64-query
                                                 // int8 scan workgroup of an early segment takes 56 KiB, so one of these
                                                 // (from another batch in flight) fits beside it
constexpr int COARSE_MAIN_LDS = 104 * 1024;      // LDS the last segment's scan reserves, see launch_coarse
constexpr int COARSE_Q = 64;                     // queries per coarse pass. (128 per pass was tried: its 128-KiB query image
                                                 // leaves LDS for only 2 waves per CU = 2 of 4 SIMDs, and the pass turns
                                                 // MFMA-bound: 5.29 ms per 128 queries vs 2 x 2.21 ms)

template <int QG, bool PREPASS, bool I8, bool Q2 = false>
int launch_coarse(const CoarseArgs& a, long long nsteps, hipStream_t st, hipEvent_t* ev) {
    static_assert(QG <= 4, "64 queries per pass at most");
    constexpr int WAVES = 4;
    // the final publication reuses the head of the query image for 3 * 16 QG counters: keep >= 1 KiB
    size_t lds = (size_t)(Q2 ? 2 : 1) * QG * (512 / (I8 ? 64 : 32)) * 1024 + WAVES * COARSE_WAVE_BYTES;
    // The LAST segment's scan (3/4 of the rows) reserves COARSE_MAIN_LDS although it uses 56-88 KiB: with two batches in
    // flight on two streams, two such scans then cannot share a CU. Sharing halves each one's bandwidth, both finish
    // together and both batches run their latency-bound side kernels at the same time with HBM idle (1.06-1.08 ms per 64
    // queries at 10 M rows); kept apart the streams settle half a batch out of phase - one scans while the other runs
    // its pre-pass, early segments, re-scoring and selects, which DO fit beside the early segments' unpadded scans - and a
    // batch takes 0.97-1.00 ms. (Padding every scan, i.e. the 114 KiB all of them used to need: 1.02-1.04 ms.)
    // development: CLIPMI_COARSE_WGS=2 - two workgroups (8 waves, 128 KB of rows in flight) per CU, no LDS padding
    static const int wgs = [] { const int v = (int)dev_knob("CLIPMI_COARSE_WGS", 1); return v == 2 ? 2 : 1; }();
    if (!PREPASS && wgs == 1 && lds < (size_t)COARSE_MAIN_LDS) lds = COARSE_MAIN_LDS;
    if (lds > (size_t)LDS_LIMIT) return set_err(CLIPMI_EUNSUPPORTED, "scan_coarse: %zu B of LDS", lds);
    if (int rc = opt_in_lds((const void*)scan_coarse_kernel<512, QG, PREPASS, I8, Q2>, lds)) return rc;
    long long g_ = (nsteps + WAVES - 1) / WAVES;
    const int grid = (int)(g_ < (long long)NUM_CU * wgs ? g_ : (long long)NUM_CU * wgs);
    if (ev) (void)hipEventRecord(ev[0], st);
    hipLaunchKernelGGL((scan_coarse_kernel<512, QG, PREPASS, I8, Q2>), dim3(grid), dim3(WAVES * 64), lds, st, a);
    if (ev) (void)hipEventRecord(ev[1], st);
    CLIPMI_CHECK_LAUNCH("scan_coarse_kernel");
    return 0;
}

constexpr int COARSE_CTL = 2 * COARSE_Q + 32 + COARSE_Q * LIVE_NB;     // fallback counters | coarse counters | overflow flag + 28 live-scan statistics | ladder
struct CoarseWs {
    uint2* cand_e; uint2* cand_c; unsigned* gcnt_e; unsigned* gcnt_c; float* thr0; float* tauc; unsigned* flag;
    unsigned* last_m; float* qmeta; uint4* qimage;
    unsigned* hist; unsigned* live_keys; float* live_edges;       // live-threshold scan (scan_coarse_live_kernel)
};

// the plan of the coarse path's side kernels (sample select, fallback scan + select): within COARSE_SIDE_LDS when K allows
bool coarse_plan(long long N, int E, int Q, int K, Plan& p) {
    const int q = Q > 32 ? 32 : Q;
    return make_plan(N, E, q, K, p, COARSE_SIDE_LDS) || make_plan(N, E, q, K, p);
}

size_t carve_coarse(const Plan& p, void* base, size_t cap, CoarseWs* w) {
    Arena ar(base ? base : reinterpret_cast<void*>(256), cap);
    CoarseWs x;
    x.cand_e = ar.take<uint2>((size_t)COARSE_Q * p.cap);   // the fallback's lists, all query groups side by side
    x.cand_c = ar.take<uint2>((size_t)COARSE_Q * COARSE_CAP);
    x.gcnt_e = ar.take<unsigned>(COARSE_CTL);          // one control block, cleared by coarse_prep_kernel
    x.gcnt_c = x.gcnt_e + COARSE_Q;
    x.flag = x.gcnt_e + 2 * COARSE_Q;
    x.hist = x.flag + 32;
    x.live_keys = ar.take<unsigned>(2 * COARSE_Q);
    x.live_edges = ar.take<float>(2 * COARSE_Q);
    x.thr0 = ar.take<float>(COARSE_Q);
    x.tauc = ar.take<float>(COARSE_Q);
    x.last_m = ar.take<unsigned>(COARSE_Q);
    x.qmeta = ar.take<float>(4 * 64);
    x.qimage = ar.take<uint4>(4 * 16 * 64);            // 64 KiB: the bf16 image of 64 queries (int8: half of it)
    if (w) *w = x;
    return ar.off + 256;
}

inline int opt_in_rescore() {
    if (int rc = opt_in_lds((const void*)rescore_pairs_kernel<512>, RESCORE_LDS)) return rc;
    return opt_in_lds((const void*)rescore_pairs16_kernel<512>, RESCORE16_LDS);
}

// wide: the lists of a wide pass (up to 1 024 queries x ~500 pairs) go to the 16-pair form, the 64-query passes' to the 64-pair one
inline void launch_rescore(bool wide, dim3 grid, hipStream_t st, const float* db, const float* q, uint2* cand, const unsigned* gcnt,
                           long long cap, const unsigned* slot_rows) {
#ifdef CLIPMI_DEV
    static const int form = (int)dev_knob("CLIPMI_RESCORE", 0);            // 16 / 64: force one form (A/B)
    if (form == 16) wide = true;
    if (form == 64) wide = false;
#endif
    if (wide) hipLaunchKernelGGL(rescore_pairs16_kernel<512>, grid, dim3(256), RESCORE16_LDS, st, db, q, cand, gcnt, cap, slot_rows);
    else hipLaunchKernelGGL(rescore_pairs_kernel<512>, grid, dim3(256), RESCORE_LDS, st, db, q, cand, gcnt, cap, slot_rows);
}

// i8 = false: dbh_dev is the bf16 copy (rmeta, amax unused); i8 = true: dbh_dev is the int8 copy, rmeta its per-row
// (scale, error norm) pairs padded to a multiple of 32 rows, amax >= every error norm
int topk_ip_coarse_impl(const void* db_dev, const void* dbh_dev, bool i8, const float2* rmeta, float amax, int64_t N, int E,
                        float rmax, const float* q_dev, int Q,
                        int K, int64_t id_base, float* out_score_dev, int64_t* out_id_dev, void* ws_dev, size_t ws_bytes,
                        void* stream, hipEvent_t* scan_ev) {
    if (!db_dev || !dbh_dev || !q_dev || !out_score_dev || !out_id_dev || !ws_dev || (i8 && !rmeta))
        return set_err(CLIPMI_EINVAL, "topk_ip_coarse: NULL pointer");
    if (i8 && !(amax >= 0.f)) return set_err(CLIPMI_EINVAL, "topk_ip_coarse_i8: amax=%g", amax);
    if (E != 512 || N < SAMPLE_MIN_N) return set_err(CLIPMI_EUNSUPPORTED, "topk_ip_coarse: needs E = 512 and N >= %d", SAMPLE_MIN_N);
    if (!(rmax > 0.f) || N >= (1ll << 32) - 1) return set_err(CLIPMI_EINVAL, "topk_ip_coarse: rmax=%g N=%lld", rmax, (long long)N);
    Plan p;
    if (!coarse_plan(N, E, Q, K, p)) return set_err(CLIPMI_EINVAL, "topk_ip_coarse: unsupported Q=%d K=%d", Q, K);
    if (ws_bytes < carve_coarse(p, nullptr, ~(size_t)0, nullptr))
        return set_err(CLIPMI_EWORKSPACE, "topk_ip_coarse: workspace %zu too small", ws_bytes);
    CoarseWs w;
    carve_coarse(p, ws_dev, ws_bytes, &w);
    hipStream_t st = as_stream(stream);
    if (int rc = opt_in_lds((const void*)select_topk_kernel, p.lds_sel)) return rc;
    if (int rc = opt_in_rescore()) return rc;
    // int8: the rows the copy's slots hold, behind the block meta (quantize_rows_i8_kernel); bf16 copy: rows in order
    const unsigned* slot_rows = i8 ? i8_slot_rows(rmeta, N) : nullptr;
    // int8 copy, shards below 2^26 rows: the live-threshold scan - EXPERIMENTAL, off unless CLIPMI_LIVE=1. Bit-exact, ONE scan
    // launch, half the re-scored pairs, but 2.3 ms per 10 M-row scan against 0.9 ms for the three segmented scans (r03): its
    // 138 KB of LDS leave one workgroup = FOUR scanner waves per CU with one 16 KB step in flight each, a quarter of the bytes
    // in flight the segmented kernel's 16 waves per CU keep, and the scan is HBM-latency bound.
#ifdef CLIPMI_DEV
    static const bool live_on = dev_knob("CLIPMI_LIVE", 0) == 1;
    const bool live = i8 && live_on && N < (1ll << LIVE_ROW_BITS);
    // CLIPMI_COARSE_Q2=1: the query as TWO int8 digits (coarse_prep_kernel q2): -42 % coarse survivors, but the scan's second
    // MFMA per fragment costs more than the re-scoring it saves - measured, development library only
    static const bool q2_on = dev_knob("CLIPMI_COARSE_Q2", 0) != 0;
    const bool q2 = i8 && q2_on && !live;
#else
    constexpr bool live = false;
    constexpr bool q2 = false;
#endif

    for (int q0 = 0; q0 < Q; q0 += COARSE_Q) {
        const int qa = (Q - q0) < COARSE_Q ? (Q - q0) : COARSE_Q;
        const float* qg = q_dev + (size_t)q0 * E;
        // per-query constants, the query image the scans copy, and the cleared counters + overflow flag (selects re-zero
        // what they consume): one launch per 64-query group
        {
            const int nq = (qa + 15) / 16 * 16;
            if (i8)
                hipLaunchKernelGGL(coarse_prep_kernel<true>, dim3(nq / 4), dim3(256), 0, st, qg, E, rmax, amax, qa, w.qmeta, w.qimage,
                                   w.gcnt_e, COARSE_CTL, COARSE_QS, 0, q2 ? 1 : 0);
            else
                hipLaunchKernelGGL(coarse_prep_kernel<false>, dim3(nq / 4), dim3(256), 0, st, qg, E, rmax, amax, qa, w.qmeta, w.qimage,
                                   w.gcnt_e, COARSE_CTL, COARSE_QS);
            CLIPMI_CHECK_LAUNCH("coarse_prep_kernel");
        }
        ScanArgs a;
        a.db = static_cast<const float*>(db_dev);
        a.K = K; a.C = p.C; a.wave_bytes = p.wave_bytes;
        a.cand = w.cand_e; a.gcnt = w.gcnt_e; a.cap = p.cap;
        // 1. exact two-level pre-pass: thr0[q] = exact K-th best score of the first S2 rows (S2 ~ N*K/2048,
        //    so that only ~2-3 k rows per query survive the coarse pass; every survivor costs a 2-KB row
        //    read in the re-scoring pass). Level 1 (S1 rows, no threshold) only feeds level 2's filter.
        // (not rounded up to a power of two: at 12.5 M rows that made the first segment 524 k rows instead of 311 k and its
        //  re-scoring, filtered only by level 1's weak bound, the largest of the three)
        long long S2 = (N * (long long)K / 2048 + 31) & ~31ll;
        if (S2 > ((N / 8) & ~31ll)) S2 = (N / 8) & ~31ll;
        if (S2 < 32768) S2 = 32768;
        // level 1 only has to thin level 2's candidates (S2*K/S1 per query): 8 k rows are enough, and its
        // select then ranks 8 k entries per query instead of 32 k
        // (12288 = what the select keeps in LDS; N >= 65536 here)
        long long S1 = p.stage < 4096 ? 4096 : (p.stage / 16) * 16;
        if (S1 > 12288) S1 = 12288;
        if (S2 < S1) S2 = S1;
        // (one unfiltered scan of S2 rows was tried for small shards: its 32 k-entry selects cost more
        //  than the level-1 scan + select they replace)
        const bool two_level = S2 > S1;
        CoarseArgs c;
        c.dbc = dbh_dev; c.rmeta = rmeta; c.qmeta = w.qmeta; c.qimage = w.qimage; c.q = qg; c.QA = qa; c.tauc = w.tauc; c.row0 = 0;
        c.cand = w.cand_c; c.gcnt = w.gcnt_c; c.cap = COARSE_CAP; c.overflow = w.flag;
        c.abl = (int)dev_knob("CLIPMI_COARSE_ABL", 0);
        // rows [r0, r1) of the copy through the coarse machinery: scan -> exact re-scoring of the survivors -> select.
        // keep & 1: the K best so far stay at the head of the candidate lists (gcnt = K) and the next segment appends behind
        // them, so no row is scanned twice; `pre` only picks the kernel NAME profilers average under.
        auto coarse_pass = [&](long long r0, long long r1, bool pre, float* thr_out, float* os, long long* oi, long long idb,
                               hipEvent_t* ev, unsigned* m_out, int keep) -> int {
            c.row0 = r0;
            c.nrows = r1 - r0;
            const long long nsteps = (c.nrows + 31) / 32;
            int rc_;
#ifdef CLIPMI_DEV
            if (i8 && q2)
                rc_ = qa <= 16 ? (pre ? launch_coarse<1, true, true, true>(c, nsteps, st, ev) : launch_coarse<1, false, true, true>(c, nsteps, st, ev))
                    : qa <= 32 ? (pre ? launch_coarse<2, true, true, true>(c, nsteps, st, ev) : launch_coarse<2, false, true, true>(c, nsteps, st, ev))
                               : (pre ? launch_coarse<4, true, true, true>(c, nsteps, st, ev) : launch_coarse<4, false, true, true>(c, nsteps, st, ev));
            else
#endif
            if (i8)
                rc_ = qa <= 16 ? (pre ? launch_coarse<1, true, true>(c, nsteps, st, ev) : launch_coarse<1, false, true>(c, nsteps, st, ev))
                    : qa <= 32 ? (pre ? launch_coarse<2, true, true>(c, nsteps, st, ev) : launch_coarse<2, false, true>(c, nsteps, st, ev))
                               : (pre ? launch_coarse<4, true, true>(c, nsteps, st, ev) : launch_coarse<4, false, true>(c, nsteps, st, ev));
            else
                rc_ = qa <= 16 ? (pre ? launch_coarse<1, true, false>(c, nsteps, st, ev) : launch_coarse<1, false, false>(c, nsteps, st, ev))
                    : qa <= 32 ? (pre ? launch_coarse<2, true, false>(c, nsteps, st, ev) : launch_coarse<2, false, false>(c, nsteps, st, ev))
                               : (pre ? launch_coarse<4, true, false>(c, nsteps, st, ev) : launch_coarse<4, false, false>(c, nsteps, st, ev));
            if (rc_) return rc_;
            // ~1-3 k survivors per query = ~11 blocks of 256 pairs; a larger grid only queues idle blocks
            launch_rescore(false, dim3(12, qa), st, static_cast<const float*>(db_dev), qg, w.cand_c, w.gcnt_c, COARSE_CAP, slot_rows);
            CLIPMI_CHECK_LAUNCH("rescore_pairs_kernel");
            // 4096 staged entries (32 KiB of LDS) cover these lists; a select block then fits on a CU even beside the
            // last segment's scan of ANOTHER batch in flight (104 KiB), which the 96-KiB staging of the sample select does not
            const int scap = p.stage < 4096 ? p.stage : 4096;
            hipLaunchKernelGGL(select_topk_kernel, dim3(qa), dim3(SEL_THREADS), SEL_FIXED + (size_t)K * 8 + (size_t)scap * 8, st,
                               w.cand_c, w.gcnt_c, COARSE_CAP, K, idb, os, oi, thr_out, (const unsigned*)nullptr, m_out, keep, scap,
                               thr_out ? w.qmeta : (float*)nullptr);
            CLIPMI_CHECK_LAUNCH("select_topk_kernel(coarse)");
            return 0;
        };
        // level 1 (exact, unfiltered, all <= 64 queries in one launch): scores of the first rows -> K-th best per query
        {
            const long long rows1 = two_level ? S1 : S2;
            const size_t lds1 = (size_t)(512 / 16) * 1024;
            if (int rc = opt_in_lds((const void*)sample_scores_kernel<512>, lds1)) return rc;
            long long gs = ((rows1 + 15) / 16 + 3) / 4;
            if (gs > NUM_CU) gs = NUM_CU;
            hipLaunchKernelGGL(sample_scores_kernel<512>, dim3((unsigned)gs, (unsigned)((qa + 15) / 16)), dim3(256), lds1, st, static_cast<const float*>(db_dev),
                               rows1, qg, qa, w.cand_c, (long long)COARSE_CAP, w.gcnt_c);
            CLIPMI_CHECK_LAUNCH("sample_scores_kernel");
            // the sample's lists are dense (entry e IS row e): only the 4 score bytes are staged, as the wide pass's sample
            // select does (48 KiB instead of 96 per block; round 5)
            hipLaunchKernelGGL(select_topk_kernel, dim3(qa), dim3(SEL_THREADS), SEL_FIXED + (size_t)K * 8 + (size_t)rows1 * 4 + 16, st,
                               w.cand_c, w.gcnt_c, COARSE_CAP, K,
                               (long long)0, (float*)nullptr, (long long*)nullptr, w.thr0, (const unsigned*)nullptr,
                               (unsigned*)nullptr, 0, (int)rows1, w.qmeta, COARSE_QS, live ? w.live_keys : (unsigned*)nullptr,
                               live ? w.live_edges : (float*)nullptr, 1, w.flag);
            CLIPMI_CHECK_LAUNCH("select_topk_kernel(sample 1)");
        }
        // Segments of the copy, each scanned ONCE: [0, S2) (level 2 of the pre-pass, threshold from level 1), then
        // [S2, N1) and [N1, N) with N1 ~ N/4. After every segment the exact K-th best of all rows seen so far is the
        // next segment's threshold: the last 75 % of the rows are filtered by the K-th best of the first 25 % (about 100
        // survivors per query at 10 M rows instead of 6 k with the level-2 threshold), which is what the exact re-scoring
        // pass - a 2-KB row read per survivor - is paid for. One more segment would save ~25 us of re-scoring and cost
        // ~55 us of launches.
        float* os_final = out_score_dev + (size_t)q0 * K;
        long long* oi_final = (long long*)out_id_dev + (size_t)q0 * K;
        long long r_done = 0;
#ifdef CLIPMI_DEV
        if (live) {
            // ONE launch over all rows: scanners + re-scoring waves, thresholds rising through the ladder (scan_coarse_live_kernel)
            LiveArgs la;
            la.dbc = static_cast<const signed char*>(dbh_dev); la.rmeta = rmeta; la.db = static_cast<const float*>(db_dev);
            la.slot_rows = slot_rows;
            la.q = qg; la.qmeta = w.qmeta; la.qimage = w.qimage; la.nrows = N; la.QA = qa; la.K = K;
            la.tau_key = w.live_keys; la.tex_key = w.live_keys + COARSE_Q; la.edge0 = w.live_edges; la.delta = w.live_edges + COARSE_Q;
            la.hist = w.hist; la.cand = w.cand_c; la.gcnt = w.gcnt_c; la.cap = COARSE_CAP; la.overflow = w.flag; la.chunk_ctr = w.flag + 1;
            static const int live_wb = [] { const int v = (int)dev_knob("CLIPMI_LIVE_WB", LIVE_WB); return v < 8 ? 8 : v; }();
            la.wb = live_wb;
            static const bool live_stats = dev_knob_set("CLIPMI_LIVE_STATS");
            la.stats = live_stats ? w.flag + 4 : nullptr;
            static const int live_abl = (int)dev_knob("CLIPMI_LIVE_ABL", 0);
            la.abl = live_abl;
            const int QGl = qa <= 16 ? 1 : qa <= 32 ? 2 : 4;
            size_t lds = (size_t)QGl * 8 * 1024 + (size_t)LIVE_QN * 8 + 64 + 512 + 1024 + 64;
            if (lds < (size_t)COARSE_MAIN_LDS) lds = COARSE_MAIN_LDS;           // as the segmented form's last segment
            const void* fn = QGl == 1 ? (const void*)scan_coarse_live_kernel<1> : QGl == 2 ? (const void*)scan_coarse_live_kernel<2>
                                                                                           : (const void*)scan_coarse_live_kernel<4>;
            if (int rc = opt_in_lds(fn, lds)) return rc;
            long long g_ = ((N + 31) / 32 + 3) / 4;
            const int grid = (int)(g_ < NUM_CU ? g_ : NUM_CU);
            static const int live_nscan = (int)dev_knob("CLIPMI_LIVE_NSCAN", 6);
            if (scan_ev) (void)hipEventRecord(scan_ev[0], st);
            if (QGl == 4 && live_nscan == 5) {
                if (int rc = opt_in_lds((const void*)scan_coarse_live_kernel<4, 5>, lds)) return rc;
                hipLaunchKernelGGL((scan_coarse_live_kernel<4, 5>), dim3(grid), dim3(512), lds, st, la);
            } else if (QGl == 4 && live_nscan == 6) {
                if (int rc = opt_in_lds((const void*)scan_coarse_live_kernel<4, 6>, lds)) return rc;
                hipLaunchKernelGGL((scan_coarse_live_kernel<4, 6>), dim3(grid), dim3(512), lds, st, la);
            } else
            if (QGl == 1) hipLaunchKernelGGL(scan_coarse_live_kernel<1>, dim3(grid), dim3(512), lds, st, la);
            else if (QGl == 2) hipLaunchKernelGGL(scan_coarse_live_kernel<2>, dim3(grid), dim3(512), lds, st, la);
            else hipLaunchKernelGGL(scan_coarse_live_kernel<4>, dim3(grid), dim3(512), lds, st, la);
            if (scan_ev) (void)hipEventRecord(scan_ev[1], st);
            CLIPMI_CHECK_LAUNCH("scan_coarse_live_kernel");
            const int scap = p.stage < 4096 ? p.stage : 4096;
            hipLaunchKernelGGL(select_topk_kernel, dim3(qa), dim3(SEL_THREADS), SEL_FIXED + (size_t)K * 8 + (size_t)scap * 8, st,
                               w.cand_c, w.gcnt_c, COARSE_CAP, K, (long long)id_base, os_final, oi_final, (float*)nullptr,
                               (const unsigned*)nullptr, w.last_m, 0, scap, (float*)nullptr);
            CLIPMI_CHECK_LAUNCH("select_topk_kernel(live)");
        }
#endif
        // development knob: CLIPMI_COARSE_SEGS=n (>= 4): n geometric segments from 64 k rows, as the wide pass plans them
        static const int nseg_env = (int)dev_knob("CLIPMI_COARSE_SEGS", 0);
        if (live) {
        } else if (two_level && nseg_env >= 4 && N >= (1 << 20)) {
            long long b = 65536;
            const double ratio = pow((double)N / 65536.0, 1.0 / (nseg_env - 1));
            for (int sgi = 0; sgi + 1 < nseg_env; ++sgi) {
                const long long r1 = b & ~31ll;
                if (int rc = coarse_pass(r_done, r1, true, w.thr0, nullptr, nullptr, 0, nullptr, w.last_m, sgi ? 3 : 1)) return rc;
                r_done = r1;
                b = (long long)(b * ratio);
            }
        } else
        if (two_level) {
            if (int rc = coarse_pass(0, S2, true, w.thr0, nullptr, nullptr, 0, scan_ev ? scan_ev + 2 : nullptr, w.last_m, 1)) return rc;
            r_done = S2;
            long long N1 = (N / 4) & ~31ll;
            if (N1 < 4 * S2) N1 = 4 * S2;
            if (N1 + 65536 <= N) {
                if (int rc = coarse_pass(r_done, N1, true, w.thr0, nullptr, nullptr, 0, scan_ev ? scan_ev + 4 : nullptr, w.last_m, 3)) return rc;
                r_done = N1;
            }
        }
        // last segment (its threshold was written by the previous select): scan, exact re-scoring, select into the result
        if (!live)
            if (int rc = coarse_pass(r_done, N, false, nullptr, os_final, oi_final, (long long)id_base, scan_ev, w.last_m, two_level ? 2 : 0))
                return rc;
        // 6. fallback: exact scan + select, exiting at once unless a coarse list overflowed
        //    (two launches: the exact scan takes its groups of p.QA queries as blockIdx.y, each with its own lists)
        {
            const int ny = (qa + p.QA - 1) / p.QA;
            a.q = qg; a.QA = qa < p.QA ? qa : p.QA; a.q_total = qa; a.nrows = N; a.thr_in = w.thr0; a.run_if = w.flag;
            if (int rc2 = launch_scan<false>(E, p.QG, a, p.grid, p.waves, p.lds_scan, st, nullptr, ny)) return rc2;
            hipLaunchKernelGGL(select_topk_kernel, dim3(qa), dim3(SEL_THREADS), p.lds_sel, st, w.cand_e, w.gcnt_e, p.cap, K,
                               (long long)id_base, os_final, oi_final, (float*)nullptr, (const unsigned*)w.flag,
                               (unsigned*)nullptr, 0, p.stage, (float*)nullptr);
            CLIPMI_CHECK_LAUNCH("select_topk_kernel(fallback)");
        }
    }
    return 0;
}

// ---- wide pass: host side -----------------------------------------------------------------------------------------
struct WideWs {
    uint2* cand_e; uint2* cand_c; unsigned* gcnt_e; unsigned* gcnt_c; unsigned* flag; float* thr0; unsigned* last_m;
    float* qmeta; uint4* qimage;
};

inline int wide_qs(int Q) { return (Q + 63) / 64 * 64; }

size_t carve_wide(const Plan& p, int Qc, void* base, size_t cap, WideWs* w) {
    const size_t qs = (size_t)wide_qs(Qc);
    Arena ar(base ? base : reinterpret_cast<void*>(256), cap);
    WideWs x;
    x.cand_e = ar.take<uint2>(qs * p.cap);            // the exact fallback's lists
    x.cand_c = ar.take<uint2>(qs * WIDE_CAP);
    x.gcnt_e = ar.take<unsigned>(2 * qs + 4);         // one control block, cleared by coarse_prep_kernel
    x.gcnt_c = x.gcnt_e + qs;
    x.flag = x.gcnt_e + 2 * qs;
    x.thr0 = ar.take<float>(qs);
    x.last_m = ar.take<unsigned>(qs);
    x.qmeta = ar.take<float>(4 * qs);
    x.qimage = ar.take<uint4>(qs / 64 * 2048);        // 32 KiB per 64-query set
    if (w) *w = x;
    return ar.off + 256;
}

bool wide_disabled() {          // CLIPMI_WIDE=0: searches of more than 64 queries as 64-query passes (A/B aid)
    static const bool off = dev_knob("CLIPMI_WIDE", 1) == 0;
    return off;
}

int wide_waves() {
    static const int w = dev_knob("CLIPMI_WIDE_WAVES", 8) == 4 ? 4 : 8;
    return w;
}

// =================================================================================================
// Wide pass, second form (more than 256 queries): QUERIES IN REGISTERS, ROWS THROUGH AN LDS RING.
//
// scan_coarse_wide_kernel keeps a 256-query tile's image in LDS and the rows in registers: one KiB of LDS per MFMA, the rows'
// fragments twice over (this block + the next) in 128 registers, 256 registers with spills - no room for a deeper fragment ring,
// and four workgroups fetch every block. Here a wave owns NG (two, or one) 32-query groups for the whole launch (their B
// fragments: 64 NG registers), a workgroup's 8 waves = a tile of 256 NG queries, and the 32-row blocks arrive by LDS-DMA in a
// ring of NB slots (the copy's block layout IS the MFMA register image, so the DMA deposits fragments as they are read): with
// two groups every row fragment read from LDS feeds two MFMAs (half a KiB per MFMA) and two workgroups fetch a block instead of
// four. All 8 waves walk the same blocks; ONE barrier per block publishes the slot that landed and frees the one that was
// read, waves 4-7 take it half a block later than waves 0-3, and the fragment ring in registers never drains.
// The compare is scan_coarse_wide_kernel's integer pre-test per lane; lanes that pass queue their sums and the per-row test
// runs on 64 queued hits at a time (see the kernel). One call of 1 024 queries at 10 M rows: 6.26 -> 5.46 ms.
// =================================================================================================
constexpr int W2_MIN_Q = 192;                     // query counts below this take scan_coarse_wide_kernel (rows in registers, queries in
                                                  // LDS: 1.17 ms against 1.23 at 128 queries, 1.65 against 1.59 at 200 - 10 M rows,
                                                  // profiles/r05_wide_balanced_tiles.txt)
constexpr int W2_SLOT = 16384 + 256;              // a ring slot: 16 KiB of fragments + the block's meta

template <int N_>
__device__ __forceinline__ void w2_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory"); }

template <int NB, int NG>      // ring slots; 32-query groups per wave (2: tiles of 512 queries, 1: tiles of 256)
__global__ void __launch_bounds__(512) scan_coarse_wide2_kernel(WideArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int KS = 16;
    constexpr int TILE_Q = 8 * NG * 32;                       // queries of a workgroup
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 31, h = lane >> 5;
    // workgroup -> (query tile, row lane): blocks b, b + 8, ... share an XCD; the tiles of a row lane are such neighbours.
    // Round 5 - BALANCED tiles (640 queries used to be a full tile + a quarter-full one at the full tile's
    // price): the call's 32-query groups are dealt evenly over the tiles, and inside a tile evenly over the waves - a wave owns
    // cnt = 0, 1 or .. NG consecutive groups (waves 0 .. rem-1 one more than the rest; SIMD partners are waves w, w + 4, so a
    // tile of 10 groups loads its SIMDs 3 / 3 / 2 / 2 instead of 4 / 4 / 2 / 0) and SKIPS the MFMAs of the groups it does not have.
    const int ngroups = (a.Q + 31) >> 5;
    const int ntile = (ngroups + 8 * NG - 1) / (8 * NG);
    const int x = blockIdx.x & 7, j = blockIdx.x >> 3, jb = gridDim.x >> 3;
    const int rlx = jb / ntile;
    if (j >= rlx * ntile) return;
    const int qt = j % ntile, nrl = __builtin_amdgcn_readfirstlane(rlx * 8);
    const int rl = (j / ntile) * 8 + x;
    const int gpt = (ngroups + ntile - 1) / ntile;                                   // groups per tile (the last may hold fewer)
    const int tg0 = qt * gpt, tgn = (ngroups - tg0) < gpt ? (ngroups - tg0) : gpt;   // this tile's groups [tg0, tg0 + tgn)
    const int gbase_ = tgn >> 3, grem_ = tgn & 7;
    const int cntw = __builtin_amdgcn_readfirstlane(gbase_ + (wave < grem_ ? 1 : 0));              // this wave's groups
    const int wg0 = __builtin_amdgcn_readfirstlane(wave * gbase_ + (wave < grem_ ? wave : grem_)); // its first group, tile-local
    const int qbase = tg0 * 32;
    char* ring = smem;                                            // NB slots of 16 KiB of fragments + 256 B of block meta
    uint2* list = reinterpret_cast<uint2*>(smem + (size_t)NB * W2_SLOT + (size_t)wave * WIDE_WAVE_BYTES);

    // this wave's NG query groups: fragments (the wide image of coarse_prep_kernel: entry [(G 16 + s) 64 + lane]) and per-lane
    // thresholds - constant over the launch
    uint4 Bq[NG][KS];
    float tq[NG], yq[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const int G = tg0 + wg0 + g;
        const bool has_img = g < cntw && G * 32 < a.qs;             // the image covers qs (a multiple of 64) queries
#pragma unroll
        for (int s_ = 0; s_ < KS; ++s_)
            Bq[g][s_] = has_img ? a.qimage[((size_t)G * KS + s_) * 64 + lane] : make_uint4(0u, 0u, 0u, 0u);
        const int qi = G * 32 + n;
        const bool act = g < cntw && qi < a.Q;
        tq[g] = act ? a.qmeta[2 * a.qs + qi] : INFINITY;
        yq[g] = act ? a.qmeta[a.qs + qi] : 0.f;
    }

    const int blk0 = (int)(a.row0 >> 5);
    const int nblk = blk0 + (int)((a.nrows + 31) >> 5);
    const unsigned last_row32 = (unsigned)(a.row0 + a.nrows - 1);
    const int first = __builtin_amdgcn_readfirstlane(blk0 + rl);
    const int nk = first < nblk ? (nblk - first + nrl - 1) / nrl : 0;       // blocks of this row lane (the same for all 8 waves)
    int npend = 0;
    if (nk > 0) {
        // block k of the row lane -> ring slot k % NB; this wave moves k-steps 2 wave, 2 wave + 1 of it (2 x 1 KiB). Past the
        // last block the last one is fetched again (into a slot nobody reads): the counted waits below stay valid to the end.
        // (buffer loads: scalar resource rebased per block, ONE long-lived lane-offset register - per-lane 64-bit addresses were
        //  recycled as fragment registers and the compiler then waited vmcnt(0) for the DMA before the first LDS read)
        const unsigned lane16 = (unsigned)lane * 16u, lane4 = (unsigned)lane * 4u;
        auto issue = [&](int k) {
            const int b_ = __builtin_amdgcn_readfirstlane(first + (k < nk ? k : nk - 1) * nrl);
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<signed char*>(a.dbc) + (size_t)b_ * (32 * 512), 0, 32 * 512, 0x00020000);
            char* dst = ring + (size_t)(k & (NB - 1)) * W2_SLOT + (size_t)(2 * wave) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) char*)dst, 16, lane16, (2 * wave) * 1024, 0,
                                                     (CLIPMI_NT_MASK & 16) ? 2 : 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) char*)(dst + 1024), 16, lane16,
                                                     (2 * wave + 1) * 1024, 0, (CLIPMI_NT_MASK & 16) ? 2 : 0);
            // the block's (scale, largest error norm) travels with it: 8 bytes through a buffer of 8 (lanes 2.. read past it: zeros),
            // every wave writes the same words - a third DMA per block and wave keeps the counted waits uniform. (A scalar load
            // returns out of order on lgkmcnt, a vector load shares vmcnt with the DMA and the compiler waits vmcnt(0) for it.)
            const __amdgpu_buffer_rsrc_t rm = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float2*>(a.bmeta) + b_, 0, 8, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rm, (__attribute__((address_space(3))) char*)(ring + (size_t)(k & (NB - 1)) * W2_SLOT + 16384),
                                                     4, lane4, 0, 0, 0);
        };
#pragma unroll
        for (int k = 0; k < NB - 1; ++k) issue(k);
        const unsigned ring_a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring + lane16;
        const unsigned list_a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)reinterpret_cast<char*>(list);
        // The compare of scan_coarse_wide_kernel (see there): an integer pre-test per lane (= query), the per-row test and the
        // ballot appends only behind it. It follows the block's MFMAs; waves 4-7 run HALF A BLOCK behind waves 0-3 (their barrier
        // sits at the end of a block instead of its middle), so on every SIMD one wave's compare runs under its partner's MFMAs.
        auto pretest = [&](const i32x16& acc, float tq_, float yq_, float2 bm_) -> bool {
            float xq = fmaf(-bm_.y, yq_, tq_) * __builtin_amdgcn_rcpf(bm_.x);      // 1 ulp: far inside the slack below
            xq = fminf(fmaxf(xq, -1.0e9f), 1.0e9f);
            xq = xq - 2.0f - fabsf(xq) * 2e-6f;
            const int dmin = (int)floorf(xq);
            auto max3 = [](int a_, int b_, int c_) { const int m_ = a_ > b_ ? a_ : b_; return m_ > c_ ? m_ : c_; };
            const int m0 = max3(acc[0], acc[1], acc[2]), m1 = max3(acc[3], acc[4], acc[5]), m2 = max3(acc[6], acc[7], acc[8]);
            const int m3 = max3(acc[9], acc[10], acc[11]), m4 = max3(acc[12], acc[13], acc[14]);
            const int ma = max3(m0, m1, m2), mb = max3(m3, m4, acc[15]);
            return ((ma > mb ? ma : mb) >= dmin) & (tq_ != INFINITY);      // +inf (padding query): never; no branch in the MFMA stream
        };
        // A lane whose pre-test passes ("hit": one (query, block) pair with possibly a passing row) does NOT run the per-row test
        // there: that is 16 values x (convert, fma, compare, ballot, branch) for the whole wave per hit, the wave is coupled to the
        // other seven by the ring's barrier, and in the last segment one of the 16 (wave, group) pairs of a workgroup hits in four
        // blocks out of five. The hit lanes drop their 16 sums + (query, first row, scale, a_max Y_q, T_q) into the wave's queue in
        // LDS (96 bytes each, by ballot prefix); when 64 are queued every lane takes ONE entry and the 16 ballot rounds serve 64
        // hits at once. Same per-row test, same survivors (their order in the lists differs; the lists are unordered).
        // (All LDS traffic here is inline asm: a compiler-visible access waits vmcnt(0) for the DMA in flight.)
        int hcount = 0;                                            // wave-uniform: entries queued
        const unsigned hq_a = (unsigned)(size_t)(__attribute__((address_space(3))) char*)(smem + (size_t)NB * W2_SLOT + 8 * WIDE_WAVE_BYTES) +
                              (unsigned)wave * (64u * 96u);
        auto drain = [&]() {
            const bool mine = lane < hcount;
            const unsigned ea = hq_a + (unsigned)(mine ? lane : 0) * 96u;
            i32x4 mt0, mt1;
            asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b128 %0, %2 offset:64\n\tds_read_b128 %1, %2 offset:80\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(mt0), "=&v"(mt1) : "v"(ea) : "memory");
            const unsigned qidx = (unsigned)mt0[0], rbase = (unsigned)mt0[1];
            const float s_ = __int_as_float(mt0[2]), ay = __int_as_float(mt0[3]), tq_ = __int_as_float(mt1[0]);
#pragma unroll
            for (int i4 = 0; i4 < 4; ++i4) {
                i32x4 v;
                asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(ea + (unsigned)i4 * 16u) : "memory");
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const unsigned row = rbase + (unsigned)(c + 8 * i4);
                    const float val = fmaf((float)v[c], s_, ay);
                    const bool pass = mine && (val >= tq_) && (row <= last_row32);
                    const unsigned long long m = __ballot(pass);
                    if (m) {
                        const int pos = npend + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        if (pass) {
                            const uint2 ent = make_uint2(qidx, row);
                            asm volatile("ds_write_b64 %0, %1" ::"v"(list_a + (unsigned)pos * 8u), "v"(ent) : "memory");
                        }
                        npend += __builtin_popcountll(m);
                    }
                }
                if (npend > WIDE_FLUSH) {                          // <= 256 appends between two tests
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    wide_flush(list, npend, qbase, a.gcnt, a.cand, a.cap, a.overflow);
                    npend = 0;
                }
            }
            hcount = 0;
        };
        auto enqueue = [&](const i32x16& acc, int g, bool hit, int blk_, float2 bm_) {
            const unsigned long long m = __ballot(hit);
            if (!m) return;
            const int nh = __builtin_popcountll(m);
            if (hcount + nh > 64) drain();
            if (hit) {
                const int slot = hcount + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                const unsigned ea = hq_a + (unsigned)slot * 96u;
                const i32x4 v0 = {acc[0], acc[1], acc[2], acc[3]}, v1 = {acc[4], acc[5], acc[6], acc[7]};
                const i32x4 v2 = {acc[8], acc[9], acc[10], acc[11]}, v3 = {acc[12], acc[13], acc[14], acc[15]};
                const i32x4 m0 = {(wg0 + g) * 32 + n, (int)((unsigned)blk_ * 32u + 4u * (unsigned)h), __float_as_int(bm_.x),
                                  __float_as_int(bm_.y * yq[g])};
                const i32x4 m1 = {__float_as_int(tq[g]), 0, 0, 0};
                asm volatile("ds_write_b128 %0, %1\n\tds_write_b128 %0, %2 offset:16\n\tds_write_b128 %0, %3 offset:32\n\t"
                             "ds_write_b128 %0, %4 offset:48\n\tds_write_b128 %0, %5 offset:64\n\tds_write_b128 %0, %6 offset:80"
                             ::"v"(ea), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(m0), "v"(m1) : "memory");
            }
            hcount += nh;
        };
        // The fragment ring never drains: the reads behind k-steps 12-15 of block k fetch k-steps 0-3 of block k + 1. Barrier b
        // publishes block b + 1 (everyone's pieces have landed) and frees block b - 1's slot for block b + NB - 1; waves 0-3 reach
        // it in the MIDDLE of block b, waves 4-7 at the END of block b - 1 (before block 0 for b = 0): same count, half a block apart.
        // lgkmcnt by hand: LDS operations complete in order, 4 fragment reads are in flight before every k-step, so lgkmcnt(3)
        // retires the oldest; anything younger in between (the meta read, an append's list writes) only makes a wait retire more.
        const bool late = wave >= 4;
        i32x4 A[4];
        auto frag = [&](int k, int s_) -> unsigned { return ring_a + (unsigned)(k & (NB - 1)) * (unsigned)W2_SLOT + (unsigned)s_ * 1024u; };
        auto ring_barrier = [&](int b_) {
            // my pieces of block b + 1 have landed (NB - 3 younger blocks may be in flight)
            w2_wait_vmcnt<3 * (NB - 3)>();
            __builtin_amdgcn_s_barrier();
            issue(b_ + NB - 1);
        };
        // block 0: my pieces landed (NB - 2 younger blocks may be in flight), then everyone's
        w2_wait_vmcnt<3 * (NB - 2)>();
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int p_ = 0; p_ < 4; ++p_) asm volatile("ds_read_b128 %0, %1" : "=v"(A[p_]) : "v"(frag(0, p_)));
        if (late) ring_barrier(0);
        // The block loop exists once per number of groups the wave really has (NGA = 0 .. NG; wave-uniform, chosen ONCE): a
        // wave-uniform `if` around the MFMAs inside one loop was tried first and tripled the launch's time - any control flow
        // inside the hand-counted stream costs the schedule. All copies take the same barriers at the same places.
        // (acc has a fixed bound: with `acc[NG]` the operands of the refill's asm are type-dependent and hipcc drops the
        //  kernel's HOST stub without a diagnostic - the library then fails to load with an undefined __device_stub__ symbol)
#define W2_BLOCKS(NGA)                                                                                                 \
        for (int k = 0; k < nk; ++k) {                                                                                 \
            const int blk = __builtin_amdgcn_readfirstlane(first + k * nrl);                                           \
            i32x16 acc[2];                                                                                             \
            _Pragma("unroll") for (int g = 0; g < 2; ++g)                                                              \
                _Pragma("unroll") for (int i = 0; i < 16; ++i) acc[g][i] = 0;                                          \
            uint2 pm = make_uint2(0u, 0u);                    /* the block's meta, read from its slot behind k-step 0 */ \
            _Pragma("unroll") for (int s_ = 0; s_ < KS; ++s_) {                                                        \
                /* (k-step 4's wait also retires the meta read: it is older than the fragment of k-step 5) */          \
                if (s_ == 4) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(A[s_ & 3]), "+v"(pm));                         \
                else asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(A[s_ & 3]));                                           \
                _Pragma("unroll") for (int g = 0; g < (NGA); ++g)                                                      \
                    acc[g] = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[s_ & 3], __builtin_bit_cast(i32x4, Bq[g][s_]), acc[g], 0, 0, 0); \
                if (s_ == KS / 2 - 1 && !late) ring_barrier(k);                                                        \
                if (s_ == KS - 1 && late && k + 1 < nk) ring_barrier(k + 1);                                           \
                /* the refill may not overtake the MFMAs that read the register: tie it to their results */            \
                const unsigned fa = s_ + 4 < KS ? frag(k, s_ + 4) : frag(k + 1, s_ + 4 - KS);                          \
                asm volatile("ds_read_b128 %0, %1" : "=v"(A[s_ & 3]) : "v"(fa), "v"(acc[0]), "v"(acc[(NGA) > 1 ? (NGA) - 1 : 0])); \
                if (s_ == 0)                                                                                           \
                    asm volatile("ds_read_b64 %0, %1" : "=v"(pm) : "v"(ring_a - lane16 + (unsigned)(k & (NB - 1)) * (unsigned)W2_SLOT + 16384u)); \
            }                                                                                                          \
            const float2 bm = make_float2(__uint_as_float(pm.x), __uint_as_float(pm.y));                               \
            _Pragma("unroll") for (int g = 0; g < (NGA); ++g) enqueue(acc[g], g, pretest(acc[g], tq[g], yq[g], bm), blk, bm); \
        }
        if (cntw >= NG) { W2_BLOCKS(NG) }
        else if (NG > 1 && cntw == 1) { W2_BLOCKS(1) }
        else { W2_BLOCKS(0) }
#undef W2_BLOCKS
        if (hcount) drain();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        w2_wait_vmcnt<0>();                                   // the re-fetched tail blocks: nothing may land after the ring is reused
    }
    // final publication, aggregated per workgroup (one returning global atomic per (workgroup, query))
    __syncthreads();
    unsigned* hist = reinterpret_cast<unsigned*>(smem);              // the ring is dead now
    unsigned* gbase = hist + TILE_Q;
    unsigned* hoff = hist + 2 * TILE_Q;
    for (int i = tid; i < TILE_Q; i += 512) { hist[i] = 0; hoff[i] = 0; }
    __syncthreads();
    for (int e = lane; e < npend; e += 64) atomicAdd(&hist[list[e].x], 1u);
    __syncthreads();
    for (int i = tid; i < TILE_Q; i += 512) gbase[i] = hist[i] ? atomicAdd(&a.gcnt[qbase + i], hist[i]) : 0u;
    __syncthreads();
    for (int e = lane; e < npend; e += 64) {
        const uint2 c = list[e];
        const unsigned pos = gbase[c.x] + atomicAdd(&hoff[c.x], 1u);
        if ((long long)pos < a.cap) a.cand[(size_t)(qbase + c.x) * a.cap + pos] = make_uint2(CAND_SLOT, c.y);
        else *a.overflow = 1u;
    }
}

// groups per wave: always the two-group instantiation since round 5 - its tiles are balanced and a wave skips the groups it does
// not have, so a part-filled tile no longer costs a full one (round 4 picked tiles of 256 queries for 513-768 queries)
inline int wide2_groups(int Q) { (void)Q; return 2; }

template <int NB, int NG>
int launch_wide2_t(const WideArgs& a, hipStream_t st, hipEvent_t* ev) {
    const size_t lds = (size_t)NB * W2_SLOT + 8 * WIDE_WAVE_BYTES + 8 * 64 * 96;        // ring | pair lists | hit queues
    if (lds > (size_t)LDS_LIMIT) return set_err(CLIPMI_EUNSUPPORTED, "scan_coarse_wide2: %zu B of LDS", lds);
    if (int rc = opt_in_lds((const void*)scan_coarse_wide2_kernel<NB, NG>, lds)) return rc;
    if (ev) (void)hipEventRecord(ev[0], st);
    hipLaunchKernelGGL((scan_coarse_wide2_kernel<NB, NG>), dim3(NUM_CU), dim3(512), lds, st, a);
    if (ev) (void)hipEventRecord(ev[1], st);
    CLIPMI_CHECK_LAUNCH("scan_coarse_wide2_kernel");
    return 0;
}

template <int NB>
int launch_wide2(const WideArgs& a, hipStream_t st, hipEvent_t* ev, int ng) {
    return ng == 2 ? launch_wide2_t<NB, 2>(a, st, ev) : launch_wide2_t<NB, 1>(a, st, ev);
}

template <int WAVES, int ABL = 0>
int launch_wide_t(const WideArgs& a, hipStream_t st, hipEvent_t* ev) {
    const size_t lds = (size_t)a.spt * 32768 + 2048 + WAVES * WIDE_WAVE_BYTES;
    if (lds > (size_t)LDS_LIMIT) return set_err(CLIPMI_EUNSUPPORTED, "scan_coarse_wide: %zu B of LDS", lds);
    if (int rc = opt_in_lds((const void*)scan_coarse_wide_kernel<WAVES, ABL>, lds)) return rc;
    if (ev) (void)hipEventRecord(ev[0], st);
    hipLaunchKernelGGL((scan_coarse_wide_kernel<WAVES, ABL>), dim3(NUM_CU), dim3(WAVES * 64), lds, st, a);
    if (ev) (void)hipEventRecord(ev[1], st);
    CLIPMI_CHECK_LAUNCH("scan_coarse_wide_kernel");
    return 0;
}

// segment boundaries of a wide search: rows [0, b[0]), [b[0], b[1]), ... [b[n-2], N); b[n-1] = N. After every segment the
// exact K-th best of all rows seen so far filters the next one, so the boundaries grow geometrically: with ratio r the
// number of exactly re-scored rows per query is about (segments) x (r - 1) x K x e^{0.36 z} - 5 segments of ratio 4 from
// 64 k rows give ~3 k at 10 M rows where the 64-query pass's three (ratio ~10 from N K / 2048) give ~5 k; a wide pass
// re-scores for up to 1024 queries at once (2 KB of f32 row per pair), so its re-scoring bytes rival the scan's
// (measured at 10 M rows, Q = 1024, same box: ratio 4 158 k q/s, 5 150 k, 6 152 k, 8 149 k, 3 146 k).
constexpr int WIDE_MAX_SEGS = 8;
int wide_segments(long long N, int K, long long S1, long long* b) {
    static const long long first = dev_knob("CLIPMI_WIDE_SEG0", 0);
    static const int ratio = [] { const int r = (int)dev_knob("CLIPMI_WIDE_SEG_RATIO", 4); return r >= 2 ? r : 4; }();
    long long s = first > 0 ? first : 65536;
    if (s < S1) s = S1;
    s = (s + 31) & ~31ll;
    int n = 0;
    while (n < WIDE_MAX_SEGS - 1 && s * 2 <= N && s + 65536 <= N) {
        b[n++] = s;
        s = (s * ratio) & ~31ll;
    }
    b[n++] = N;
    (void)K;
    return n;
}

// int8 coarse-then-exact search of 64 < Q queries: chunks of <= WIDE_MAX_Q queries, each ONE pass of the copy in segments
int topk_wide_impl(const void* db_dev, const void* db8_dev, const float2* rmeta, float amax, int64_t N, int E, float rmax,
                   const float* q_dev, int Q, int K, int64_t id_base, float* out_score_dev, int64_t* out_id_dev, void* ws_dev,
                   size_t ws_bytes, void* stream, hipEvent_t* scan_ev, int max_ev, int* n_ev) {
    if (!db_dev || !db8_dev || !q_dev || !out_score_dev || !out_id_dev || !ws_dev || !rmeta)
        return set_err(CLIPMI_EINVAL, "topk_ip_coarse_i8: NULL pointer");
    if (!(amax >= 0.f)) return set_err(CLIPMI_EINVAL, "topk_ip_coarse_i8: amax=%g", amax);
    if (E != 512 || N < SAMPLE_MIN_N) return set_err(CLIPMI_EUNSUPPORTED, "topk_ip_coarse: needs E = 512 and N >= %d", SAMPLE_MIN_N);
    if (!(rmax > 0.f) || N >= (1ll << 32) - 1) return set_err(CLIPMI_EINVAL, "topk_ip_coarse: rmax=%g N=%lld", rmax, (long long)N);
    Plan p;
    if (!coarse_plan(N, E, Q, K, p)) return set_err(CLIPMI_EINVAL, "topk_ip_coarse: unsupported Q=%d K=%d", Q, K);
    const int qc_max = Q < WIDE_MAX_Q ? Q : WIDE_MAX_Q;
    if (ws_bytes < carve_wide(p, qc_max, nullptr, ~(size_t)0, nullptr))
        return set_err(CLIPMI_EWORKSPACE, "topk_ip_coarse_i8: workspace %zu too small", ws_bytes);
    WideWs w;
    carve_wide(p, qc_max, ws_dev, ws_bytes, &w);
    const int qs = wide_qs(qc_max);
    hipStream_t st = as_stream(stream);
    const unsigned* slot_rows = i8_slot_rows(rmeta, N);
    if (int rc = opt_in_lds((const void*)select_topk_kernel, p.lds_sel)) return rc;
    if (int rc = opt_in_rescore()) return rc;
    const size_t lds1 = (size_t)(512 / 16) * 1024;
    if (int rc = opt_in_lds((const void*)sample_scores_kernel<512>, lds1)) return rc;
    int ev_used = 0;

    for (int q0 = 0; q0 < Q; q0 += WIDE_MAX_Q) {
        const int qc = (Q - q0) < WIDE_MAX_Q ? (Q - q0) : WIDE_MAX_Q;
        const float* qg = q_dev + (size_t)q0 * E;
        const int nsets = (qc + 63) / 64;
        hipLaunchKernelGGL(coarse_prep_kernel<true>, dim3(nsets * 16), dim3(256), 0, st, qg, E, rmax, amax, qc, w.qmeta, w.qimage,
                           w.gcnt_e, 2 * qs + 4, qs, 1);
        CLIPMI_CHECK_LAUNCH("coarse_prep_kernel(wide)");
        long long S1 = p.stage < 4096 ? 4096 : (p.stage / 16) * 16;
        if (S1 > 12288) S1 = 12288;
        {
            // grid.y = 16-query groups (up to 64): blocks of the same x read the same rows, and every block first copies a
            // 32-KiB query image - so few, fat blocks per group (each wave 4+ row tiles) once the groups alone fill the chip
            long long gs = ((S1 + 15) / 16 + 3) / 4;
            if (gs > NUM_CU) gs = NUM_CU;
            static const int sdiv = [] { const int d = (int)dev_knob("CLIPMI_WIDE_SAMPLE_DIV", 8); return d > 0 ? d : 8; }();
            static const int sqg = (int)dev_knob("CLIPMI_WIDE_SAMPLE_QG", 2);
            const int ngroups = (qc + 16 * sqg - 1) / (16 * sqg);
            if (ngroups >= 8 / sqg && gs > sdiv) gs = (gs + sdiv - 1) / sdiv;
            if (sqg == 2) {
                if (int rc = opt_in_lds((const void*)sample_scores_kernel<512, 2>, 2 * lds1)) return rc;
                hipLaunchKernelGGL((sample_scores_kernel<512, 2>), dim3((unsigned)gs, (unsigned)ngroups), dim3(256), 2 * lds1, st,
                                   static_cast<const float*>(db_dev), S1, qg, qc, w.cand_c, (long long)WIDE_CAP, w.gcnt_c);
            } else
            hipLaunchKernelGGL(sample_scores_kernel<512>, dim3((unsigned)gs, (unsigned)ngroups), dim3(256), lds1, st,
                               static_cast<const float*>(db_dev), S1, qg, qc, w.cand_c, (long long)WIDE_CAP, w.gcnt_c);
            CLIPMI_CHECK_LAUNCH("sample_scores_kernel(wide)");
            // dense staging (4 bytes per row): S1 <= 12 288 rows always fit; LDS sized for them, not for the generic 8-byte stage
            const size_t lds_dense = SEL_FIXED + (size_t)K * 8 + (size_t)S1 * 4 + 16;
            hipLaunchKernelGGL(select_topk_kernel, dim3(qc), dim3(SEL_THREADS), lds_dense, st, w.cand_c, w.gcnt_c, WIDE_CAP, K,
                               (long long)0, (float*)nullptr, (long long*)nullptr, w.thr0, (const unsigned*)nullptr,
                               (unsigned*)nullptr, 0, (int)S1, w.qmeta, qs, (unsigned*)nullptr, (float*)nullptr, 1, w.flag);
            CLIPMI_CHECK_LAUNCH("select_topk_kernel(wide sample)");
        }
        WideArgs c;
        c.dbc = static_cast<const signed char*>(db8_dev); c.rmeta = rmeta; c.qmeta = w.qmeta; c.qimage = w.qimage;
        c.bmeta = rmeta + i8_row_meta_entries(N);
        c.Q = qc; c.qs = qs;
        c.nqt = (nsets + WIDE_TILE_SETS - 1) / WIDE_TILE_SETS;
        c.spt = (nsets + c.nqt - 1) / c.nqt;
        c.cand = w.cand_c; c.gcnt = w.gcnt_c; c.cap = WIDE_CAP; c.overflow = w.flag;
        {
            static const int map_env = (int)dev_knob("CLIPMI_WIDE_MAP", 0), pf_env = (int)dev_knob("CLIPMI_WIDE_PF", 0);
            c.map_mode = map_env; c.pf_mode = pf_env;
        }
        long long bnd[WIDE_MAX_SEGS];
        const int nseg = wide_segments(N, K, S1, bnd);
        float* os_final = out_score_dev + (size_t)q0 * K;
        long long* oi_final = (long long*)out_id_dev + (size_t)q0 * K;
        const int scap = p.stage < 4096 ? p.stage : 4096;
        long long r0 = 0;
        for (int sgi = 0; sgi < nseg; ++sgi) {
            const bool last = sgi + 1 == nseg;
            c.row0 = r0;
            c.nrows = bnd[sgi] - r0;
            hipEvent_t* ev = (scan_ev && ev_used + 2 <= max_ev) ? scan_ev + ev_used : nullptr;
#ifdef CLIPMI_DEV
            static const int abl = (int)dev_knob("CLIPMI_WIDE_ABL", 0);        // ablations + the 4-wave form: development build only
            // CLIPMI_WIDE2=0: never the second form (A/B); CLIPMI_WIDE2_MINQ: smallest query count that takes it
            static const int w2 = (int)dev_knob("CLIPMI_WIDE2", 1);
            static const int w2_minq = (int)dev_knob("CLIPMI_WIDE2_MINQ", W2_MIN_Q);
            static const int w2_ng = (int)dev_knob("CLIPMI_WIDE2_NG", 0);              // 1 / 2: force the groups per wave (A/B)
            if (w2 && qc >= w2_minq) {
                if (int rc = launch_wide2<4>(c, st, ev, w2_ng == 1 || w2_ng == 2 ? w2_ng : wide2_groups(qc))) return rc;
            } else
            if (int rc = abl == 1 ? launch_wide_t<8, 1>(c, st, ev) : abl == 2 ? launch_wide_t<8, 2>(c, st, ev)
                       : abl == 3 ? launch_wide_t<8, 3>(c, st, ev)
                       : wide_waves() == 4 ? launch_wide_t<4>(c, st, ev) : launch_wide_t<8>(c, st, ev)) return rc;
#else
            if (int rc = qc >= W2_MIN_Q ? launch_wide2<4>(c, st, ev, wide2_groups(qc)) : launch_wide_t<8>(c, st, ev)) return rc;
#endif
            if (ev) ev_used += 2;
            launch_rescore(true, dim3(12, qc), st, static_cast<const float*>(db_dev), qg, w.cand_c, w.gcnt_c, WIDE_CAP, slot_rows);
            CLIPMI_CHECK_LAUNCH("rescore_pairs16_kernel(wide)");
            // keep: bit 0 = the K best stay at the head of the list for the next segment; bit 1 = add to the survivor count
            const int keep = (last ? 0 : 1) | (sgi > 0 ? 2 : 0);
            hipLaunchKernelGGL(select_topk_kernel, dim3(qc), dim3(SEL_THREADS), SEL_FIXED + (size_t)K * 8 + (size_t)scap * 8, st,
                               w.cand_c, w.gcnt_c, WIDE_CAP, K, last ? (long long)id_base : 0ll, last ? os_final : (float*)nullptr,
                               last ? oi_final : (long long*)nullptr, last ? (float*)nullptr : w.thr0, (const unsigned*)nullptr,
                               w.last_m, keep, scap, last ? (float*)nullptr : w.qmeta, qs);
            CLIPMI_CHECK_LAUNCH("select_topk_kernel(wide)");
            r0 = bnd[sgi];
        }
        // fallback: exact scan + select, exiting at once unless a list overflowed
        {
            ScanArgs a;
            a.db = static_cast<const float*>(db_dev);
            a.K = K; a.C = p.C; a.wave_bytes = p.wave_bytes;
            a.cand = w.cand_e; a.gcnt = w.gcnt_e; a.cap = p.cap;
            const int ny = (qc + p.QA - 1) / p.QA;
            a.q = qg; a.QA = qc < p.QA ? qc : p.QA; a.q_total = qc; a.nrows = N; a.thr_in = w.thr0; a.run_if = w.flag;
            if (int rc2 = launch_scan<false>(E, p.QG, a, p.grid, p.waves, p.lds_scan, st, nullptr, ny)) return rc2;
            hipLaunchKernelGGL(select_topk_kernel, dim3(qc), dim3(SEL_THREADS), p.lds_sel, st, w.cand_e, w.gcnt_e, p.cap, K,
                               (long long)id_base, os_final, oi_final, (float*)nullptr, (const unsigned*)w.flag,
                               (unsigned*)nullptr, 0, p.stage, (float*)nullptr, qs);
            CLIPMI_CHECK_LAUNCH("select_topk_kernel(wide fallback)");
        }
    }
    if (n_ev) *n_ev = ev_used;
    return 0;
}
}  // namespace
}  // namespace clipmi

extern "C" size_t clipmi_topk_ip_coarse_workspace_bytes(int64_t N, int E, int Q, int K) {
    Plan p;
    if (E != 512 || N < SAMPLE_MIN_N || !coarse_plan(N, E, Q, K, p)) {
        set_err(CLIPMI_EUNSUPPORTED, "topk_ip_coarse: needs E = 512, N >= %d and a supported K", SAMPLE_MIN_N);
        return 0;
    }
    size_t need = carve_coarse(p, nullptr, ~(size_t)0, nullptr);
    if (Q > COARSE_Q) {                       // the int8 path takes a search of more than 64 queries as wide passes
        const size_t wide = carve_wide(p, Q < WIDE_MAX_Q ? Q : WIDE_MAX_Q, nullptr, ~(size_t)0, nullptr);
        need = wide > need ? wide : need;
    }
    return need;
}

extern "C" int clipmi_topk_ip_coarse(const void* db_dev, const void* db_bf16_dev, int64_t N, int E, float rmax,
                                     const float* q_dev, int Q, int K, int64_t id_base, float* out_score_dev,
                                     int64_t* out_id_dev, void* ws_dev, size_t ws_bytes, void* stream) {
    return topk_ip_coarse_impl(db_dev, db_bf16_dev, false, nullptr, 0.f, N, E, rmax, q_dev, Q, K, id_base, out_score_dev,
                               out_id_dev, ws_dev, ws_bytes, stream, nullptr);
}

extern "C" size_t clipmi_i8_copy_bytes(int64_t N, int E) {
    return N < 0 || E < 32 ? 0 : (size_t)((N + 31) / 32 * 32) * (size_t)E;
}

extern "C" size_t clipmi_i8_meta_bytes(int64_t N) {
    return N < 0 ? 0 : (i8_row_meta_entries(N) + i8_block_meta_entries(N)) * sizeof(float2) + (size_t)((N + 31) / 32 * 32 + 32) * sizeof(unsigned);
}

extern "C" int clipmi_quantize_rows_i8(const float* db_dev, int64_t N, int E, const uint32_t* perm_dev, void* out_i8_dev,
                                       size_t out_i8_bytes, float* meta_dev, size_t meta_bytes, void* stream) {
    if (!db_dev || !out_i8_dev || !meta_dev || N < 1 || E < 32 || E % 32 != 0)
        return set_err(CLIPMI_EINVAL, "quantize_rows_i8: bad arguments (N=%lld E=%d; E must be a multiple of 32)", (long long)N, E);
    if (out_i8_bytes < clipmi_i8_copy_bytes(N, E) || meta_bytes < clipmi_i8_meta_bytes(N))
        return set_err(CLIPMI_EINVAL, "quantize_rows_i8: copy %zu B / meta %zu B, need %zu / %zu (clipmi_i8_copy_bytes, clipmi_i8_meta_bytes)",
                       out_i8_bytes, meta_bytes, clipmi_i8_copy_bytes(N, E), clipmi_i8_meta_bytes(N));
    if (N >= (1ll << 32) - 1) return set_err(CLIPMI_EINVAL, "quantize_rows_i8: N=%lld (slots are 32-bit)", (long long)N);
    float2* meta = reinterpret_cast<float2*>(meta_dev);
    hipLaunchKernelGGL(quantize_rows_i8_kernel, dim3((unsigned)((N + 31) / 32)), dim3(256), 0, as_stream(stream), db_dev,
                       (long long)N, E, perm_dev, static_cast<signed char*>(out_i8_dev), meta, meta + i8_row_meta_entries(N),
                       const_cast<unsigned*>(i8_slot_rows(meta, N)));
    CLIPMI_CHECK_LAUNCH("quantize_rows_i8_kernel");
    return 0;
}

// ---- build-side helpers of the coarse copies (index load time, reference build-index.py:75-89 / query-index.py:60-75:
// the vectors are read once and kept): the bounds the coarse passes need and the bf16 copy, computed on the device by
// the library itself (r02 did these with framework tensor ops).
// stats2[0] = largest row norm, accumulated in f64 and rounded UP to f32 (a valid `rmax`); stats2[1] = largest a_r of
// the int8 copy's row meta (0 when meta is null). Non-negative floats order as their bit patterns: atomicMax on uint.
__global__ void __launch_bounds__(256) rows_stats_kernel(const float* __restrict__ db, long long N, int E, const float2* __restrict__ meta,
                                                         unsigned* __restrict__ stats2) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    float best = 0.f, abest = 0.f;
    for (long long r = wave0; r < N; r += nw) {
        const float* row = db + (size_t)r * E;
        double acc = 0.0;
        for (int k = lane * 4; k < E; k += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(row + k);
            acc += (double)v.x * v.x + (double)v.y * v.y + (double)v.z * v.z + (double)v.w * v.w;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
        const double nd = sqrt(acc);
        float nf = (float)nd;
        if ((double)nf < nd) nf = nextafterf(nf, INFINITY);
        // a NaN row (or NaN error norm) must surface as a non-finite maximum: the index then keeps the search on the exact
        // path (index.py: isfinite(rmax)); `nf > best` alone would silently drop it
        best = !(nf == nf) ? INFINITY : (nf > best ? nf : best);
        if (meta && lane == 0) {
            const float a_ = meta[r].y;
            abest = !(a_ == a_) ? INFINITY : (a_ > abest ? a_ : abest);
        }
    }
    if (lane == 0) {
        atomicMax(stats2, __float_as_uint(best));
        if (meta) atomicMax(stats2 + 1, __float_as_uint(abest));
    }
}

__global__ void __launch_bounds__(256) rows_absmax_kernel(const float* __restrict__ db, long long N, int E, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    for (long long r = wave0; r < N; r += nw) {
        float mx = 0.f;
        for (int k = lane * 4; k < E; k += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(db + (size_t)r * E + k);
            mx = fmaxf(fmaxf(mx, fabsf(v.x)), fmaxf(fabsf(v.y), fmaxf(fabsf(v.z), fabsf(v.w))));
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        if (lane == 0) out[r] = mx;
    }
}

__global__ void __launch_bounds__(256) rows_to_bf16_kernel(const float* __restrict__ db, long long n4, uint2* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(db)[i];
        out[i] = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
    }
}

extern "C" int clipmi_rows_stats(const float* db_dev, int64_t N, int E, const float* meta_dev, float* stats2_dev, void* stream) {
    if (!db_dev || !stats2_dev || N < 1 || E < 4 || E % 4 != 0)
        return set_err(CLIPMI_EINVAL, "rows_stats: bad arguments (N=%lld E=%d; E must be a multiple of 4)", (long long)N, E);
    if (hipMemsetAsync(stats2_dev, 0, 2 * sizeof(float), as_stream(stream)) != hipSuccess) return set_err(CLIPMI_EHIP, "rows_stats: memset");
    const long long want = (N + 3) / 4;
    const unsigned grid = (unsigned)(want < 8 * NUM_CU ? want : 8 * NUM_CU);
    hipLaunchKernelGGL(rows_stats_kernel, dim3(grid), dim3(256), 0, as_stream(stream), db_dev, (long long)N, E,
                       reinterpret_cast<const float2*>(meta_dev), reinterpret_cast<unsigned*>(stats2_dev));
    CLIPMI_CHECK_LAUNCH("rows_stats_kernel");
    return 0;
}

extern "C" int clipmi_rows_absmax(const float* db_dev, int64_t N, int E, float* out_dev, void* stream) {
    if (!db_dev || !out_dev || N < 1 || E < 4 || E % 4 != 0) return set_err(CLIPMI_EINVAL, "rows_absmax: bad arguments (N=%lld E=%d)", (long long)N, E);
    const long long want = (N + 3) / 4;
    const unsigned grid = (unsigned)(want < 16 * NUM_CU ? want : 16 * NUM_CU);
    hipLaunchKernelGGL(rows_absmax_kernel, dim3(grid), dim3(256), 0, as_stream(stream), db_dev, (long long)N, E, out_dev);
    CLIPMI_CHECK_LAUNCH("rows_absmax_kernel");
    return 0;
}

// ---- clipmi_rows_order_by_absmax: the row order of the int8 copy (slot t holds row perm[t]), built by the
// library itself so that a C-ABI caller needs no framework sort: perm = the rows ordered by their largest |component|
// ascending, equal maxima in row order (a stable sort). Keys are the f32 bits of the maxima (non-negative floats order as
// their bit patterns); a least-significant-digit radix sort, four passes of 8 bits, each pass = per-block digit histogram ->
// one exclusive scan over (digit, block) -> stable scatter (a block keeps the order of its own items: ranks by wave ballots).
constexpr int ORD_THREADS = 256, ORD_ROUNDS = 8, ORD_TILE = ORD_THREADS * ORD_ROUNDS;

__global__ void __launch_bounds__(ORD_THREADS) order_hist_kernel(const unsigned* __restrict__ keys, long long N, int shift,
                                                                 unsigned* __restrict__ hist, unsigned nblocks) {
    __shared__ unsigned h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * ORD_TILE;
    for (int r = 0; r < ORD_ROUNDS; ++r) {
        const long long i = base + r * ORD_THREADS + threadIdx.x;
        if (i < N) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(size_t)threadIdx.x * nblocks + blockIdx.x] = h[threadIdx.x];      // digit-major: one scan gives every (digit, block) base
}

// exclusive scan of n counters in place, ONE workgroup of 1024 threads (n = 256 x blocks: 0.6 M entries at 10 M rows): chunks of
// 4096 entries, four consecutive ones per thread (16-byte accesses), a wave scan + the 16 wave totals per chunk, a running carry
__global__ void __launch_bounds__(1024) order_scan_kernel(unsigned* __restrict__ v, long long n) {
    __shared__ unsigned wtot[16];
    __shared__ unsigned carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (long long base = 0; base < n; base += 4096) {
        const long long i0 = base + 4ll * tid;
        unsigned x[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) x[j] = i0 + j < n ? v[i0 + j] : 0u;
        const unsigned mine = x[0] + x[1] + x[2] + x[3];
        unsigned inc = mine;                               // inclusive scan over the wave
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned t = __shfl_up(inc, o);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        unsigned before = carry_s;
        for (int w = 0; w < wave; ++w) before += wtot[w];
        unsigned run = before + inc - mine;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (i0 + j < n) v[i0 + j] = run;
            run += x[j];
        }
        __syncthreads();
        if (tid == 1023) carry_s = before + inc;            // the chunk's total joins the carry
        __syncthreads();
    }
}

__global__ void __launch_bounds__(ORD_THREADS) order_scatter_kernel(const unsigned* __restrict__ keys_in, const unsigned* __restrict__ idx_in,
                                                                    long long N, int shift, const unsigned* __restrict__ offs,
                                                                    unsigned nblocks, unsigned* __restrict__ keys_out,
                                                                    unsigned* __restrict__ idx_out) {
    __shared__ unsigned run[256];              // items of this block already placed, per digit
    __shared__ unsigned wcnt[4][256];          // this round's count per (wave, digit)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    run[tid] = 0;
    const unsigned goff = offs[(size_t)tid * nblocks + blockIdx.x];        // thread t keeps digit t's global base
    __shared__ unsigned gbase[256];
    gbase[tid] = goff;
    const long long base = (long long)blockIdx.x * ORD_TILE;
    for (int r = 0; r < ORD_ROUNDS; ++r) {
#pragma unroll
        for (int w = 0; w < 4; ++w) wcnt[w][tid] = 0;
        __syncthreads();
        const long long i = base + r * ORD_THREADS + tid;
        const bool live = i < N;
        const unsigned key = live ? keys_in[i] : 0u;
        const unsigned d = (key >> shift) & 255u;
        // lanes of this wave with the same digit (dead lanes match nobody): eight ballots
        unsigned long long same = __ballot(live);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long bal = __ballot(live && ((d >> b) & 1u));
            same &= ((d >> b) & 1u) ? bal : ~bal;
        }
        const unsigned below = (unsigned)__popcll(same & ((1ull << lane) - 1ull));
        if (live && below == 0) wcnt[wave][d] = (unsigned)__popcll(same);
        __syncthreads();
        if (live) {
            unsigned pos = gbase[d] + run[d] + below;
            for (int w = 0; w < wave; ++w) pos += wcnt[w][d];
            keys_out[pos] = key;
            idx_out[pos] = idx_in ? idx_in[i] : (unsigned)i;
        }
        __syncthreads();
        run[tid] += wcnt[0][tid] + wcnt[1][tid] + wcnt[2][tid] + wcnt[3][tid];
        // (the next round's clear of wcnt is ordered behind this read by the barrier that follows the clear's own round start:
        //  every thread clears only its own column tid, which it alone has just read)
    }
}

static inline size_t order_blocks(int64_t N) { return (size_t)((N + ORD_TILE - 1) / ORD_TILE); }

extern "C" size_t clipmi_rows_order_workspace_bytes(int64_t N) {
    if (N < 1 || N >= (1ll << 32) - 1) return 0;
    // keys A | keys B | ids B | histogram [256][blocks]      (ids A is the caller's perm buffer)
    return (3 * (size_t)N + 256 * order_blocks(N)) * sizeof(unsigned) + 64;
}

extern "C" int clipmi_rows_order_by_absmax(const float* db_dev, int64_t N, int E, uint32_t* perm_dev, void* ws_dev, size_t ws_bytes,
                                           void* stream) {
    if (!db_dev || !perm_dev || !ws_dev || N < 1 || E < 4 || E % 4 != 0)
        return set_err(CLIPMI_EINVAL, "rows_order_by_absmax: bad arguments (N=%lld E=%d)", (long long)N, E);
    const size_t need = clipmi_rows_order_workspace_bytes(N);
    if (need == 0 || ws_bytes < need)
        return set_err(CLIPMI_EINVAL, "rows_order_by_absmax: workspace %zu B, need %zu (clipmi_rows_order_workspace_bytes; N < 2^32 - 1)", ws_bytes, need);
    unsigned* ka = static_cast<unsigned*>(ws_dev);
    unsigned* kb = ka + N;
    unsigned* ib = kb + N;
    unsigned* hist = ib + N;
    const unsigned nb = (unsigned)order_blocks(N);
    int rc = clipmi_rows_absmax(db_dev, N, E, reinterpret_cast<float*>(ka), stream);
    if (rc) return rc;
    // passes 0, 2 write (kb, ib), passes 1, 3 write (ka, perm): the last pass lands in the caller's buffer
    for (int pass = 0; pass < 4; ++pass) {
        const unsigned* kin = pass & 1 ? kb : ka;
        unsigned* kout = pass & 1 ? ka : kb;
        const unsigned* iin = pass == 0 ? nullptr : (pass & 1 ? ib : perm_dev);
        unsigned* iout = pass & 1 ? perm_dev : ib;
        hipLaunchKernelGGL(order_hist_kernel, dim3(nb), dim3(ORD_THREADS), 0, as_stream(stream), kin, (long long)N, 8 * pass, hist, nb);
        hipLaunchKernelGGL(order_scan_kernel, dim3(1), dim3(1024), 0, as_stream(stream), hist, 256ll * nb);
        hipLaunchKernelGGL(order_scatter_kernel, dim3(nb), dim3(ORD_THREADS), 0, as_stream(stream), kin, iin, (long long)N, 8 * pass,
                           hist, nb, kout, iout);
    }
    CLIPMI_CHECK_LAUNCH("order_scatter_kernel");
    return 0;
}

extern "C" int clipmi_rows_to_bf16(const float* db_dev, int64_t N, int E, void* out_bf16_dev, void* stream) {
    if (!db_dev || !out_bf16_dev || N < 1 || E < 4 || E % 4 != 0)
        return set_err(CLIPMI_EINVAL, "rows_to_bf16: bad arguments (N=%lld E=%d; E must be a multiple of 4)", (long long)N, E);
    const long long n4 = (long long)N * E / 4, want = (n4 + 255) / 256;
    const unsigned grid = (unsigned)(want < 16 * NUM_CU ? want : 16 * NUM_CU);
    hipLaunchKernelGGL(rows_to_bf16_kernel, dim3(grid), dim3(256), 0, as_stream(stream), db_dev, n4, static_cast<uint2*>(out_bf16_dev));
    CLIPMI_CHECK_LAUNCH("rows_to_bf16_kernel");
    return 0;
}

extern "C" int clipmi_topk_ip_coarse_i8(const void* db_dev, const void* db_i8_dev, const float* meta_dev, float amax, int64_t N,
                                        int E, float rmax, const float* q_dev, int Q, int K, int64_t id_base,
                                        float* out_score_dev, int64_t* out_id_dev, void* ws_dev, size_t ws_bytes, void* stream) {
    if (Q > COARSE_Q && !wide_disabled())
        return topk_wide_impl(db_dev, db_i8_dev, reinterpret_cast<const float2*>(meta_dev), amax, N, E, rmax, q_dev, Q, K, id_base,
                              out_score_dev, out_id_dev, ws_dev, ws_bytes, stream, nullptr, 0, nullptr);
    return topk_ip_coarse_impl(db_dev, db_i8_dev, true, reinterpret_cast<const float2*>(meta_dev), amax, N, E, rmax, q_dev, Q,
                               K, id_base, out_score_dev, out_id_dev, ws_dev, ws_bytes, stream, nullptr);
}

// Measurement hook: clipmi_topk_ip_coarse `reps` times with events around the bf16 scan kernel (Q <= 64).
static int dbg_coarse_scan_ms(const void* db_dev, const void* db_bf16_dev, bool i8, const float2* rmeta, float amax, int64_t N,
                              int E, float rmax, const float* q_dev, int Q, int K, float* out_score_dev,
                              int64_t* out_id_dev, void* ws_dev, size_t ws_bytes, void* stream, int reps, float* scan_ms,
                              long long* survivors);

extern "C" int clipmi_dbg_topk_coarse_scan_ms(const void* db_dev, const void* db_bf16_dev, int64_t N, int E, float rmax,
                                              const float* q_dev, int Q, int K, float* out_score_dev, int64_t* out_id_dev,
                                              void* ws_dev, size_t ws_bytes, void* stream, int reps, float* scan_ms,
                                              long long* survivors) {
    return dbg_coarse_scan_ms(db_dev, db_bf16_dev, false, nullptr, 0.f, N, E, rmax, q_dev, Q, K, out_score_dev, out_id_dev,
                              ws_dev, ws_bytes, stream, reps, scan_ms, survivors);
}

extern "C" int clipmi_dbg_topk_coarse_i8_scan_ms(const void* db_dev, const void* db_i8_dev, const float* meta_dev, float amax,
                                                 int64_t N, int E, float rmax, const float* q_dev, int Q, int K,
                                                 float* out_score_dev, int64_t* out_id_dev, void* ws_dev, size_t ws_bytes,
                                                 void* stream, int reps, float* scan_ms, long long* survivors) {
    return dbg_coarse_scan_ms(db_dev, db_i8_dev, true, reinterpret_cast<const float2*>(meta_dev), amax, N, E, rmax, q_dev, Q, K,
                              out_score_dev, out_id_dev, ws_dev, ws_bytes, stream, reps, scan_ms, survivors);
}

static int dbg_coarse_scan_ms(const void* db_dev, const void* db_bf16_dev, bool i8, const float2* rmeta, float amax, int64_t N,
                              int E, float rmax, const float* q_dev, int Q, int K, float* out_score_dev,
                              int64_t* out_id_dev, void* ws_dev, size_t ws_bytes, void* stream, int reps, float* scan_ms,
                              long long* survivors) {
    if (!scan_ms || reps < 1 || Q > COARSE_Q) return set_err(CLIPMI_EINVAL, "dbg_topk_coarse_scan_ms: bad arguments");
    // three event pairs: [0,1] the last segment, [2,3] rows [0, S2), [4,5] rows [S2, N1) of the coarse copy (see
    // topk_ip_coarse_impl): together the launches that stream the copy ONCE; *scan_ms = their summed duration
    hipEvent_t ev[6];
    for (int i = 0; i < 6; ++i)
        if (hipEventCreate(&ev[i]) != hipSuccess) return set_err(CLIPMI_EHIP, "hipEventCreate");
    double total = 0.0;
    int rc = 0;
    for (int i = 0; i < reps && rc == 0; ++i) {
        for (int j = 0; j < 6; ++j) (void)hipEventRecord(ev[j], as_stream(stream));     // a skipped segment reads ~0
        rc = topk_ip_coarse_impl(db_dev, db_bf16_dev, i8, rmeta, amax, N, E, rmax, q_dev, Q, K, 0, out_score_dev, out_id_dev,
                                 ws_dev, ws_bytes, stream, ev);
        if (rc) break;
        if (hipStreamSynchronize(as_stream(stream)) != hipSuccess) { rc = set_err(CLIPMI_EHIP, "hipStreamSynchronize"); break; }
        for (int j = 0; j < 3; ++j) {
            float ms = 0.f;
            (void)hipEventElapsedTime(&ms, ev[2 * j], ev[2 * j + 1]);
            total += ms;
        }
    }
    for (int i = 0; i < 6; ++i) (void)hipEventDestroy(ev[i]);
    if (rc == 0) *scan_ms = (float)(total / reps);
    if (rc == 0 && dev_knob_set("CLIPMI_LIVE_STATS")) {
        Plan p; CoarseWs w; coarse_plan(N, E, Q, K, p); carve_coarse(p, ws_dev, ws_bytes, &w);
        unsigned h[28];
        if (hipMemcpy(h, w.flag + 4, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess) {
            fprintf(stderr, "scanner wave time, mean us: matrix loop (incl. waiting for rows) %.1f | bounds + compare %.1f | push %.1f\n",
                    h[12] * 16 * 0.01 / (h[15] ? h[15] : 1), h[13] * 16 * 0.01 / (h[15] ? h[15] : 1), h[14] * 16 * 0.01 / (h[15] ? h[15] : 1));
            fprintf(stderr, "live scan (last call): pushed %u (push spins %u) popped %u stale %u re-scored %u inserted %u rounds %u (phases %u) ladder rises %u | last scanner wave done at %.1f us (mean %.1f), last re-scoring wave at %.1f us\n",
                    h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[11], h[7], h[8] * 0.01, h[10] * 1024.0 * 0.01 / (h[15] ? h[15] : 1), h[9] * 0.01);
        }
    }
    if (rc == 0 && survivors) {       // rows that survived the coarse pass, summed over the Q queries of the last call
        Plan p;
        CoarseWs w;
        coarse_plan(N, E, Q, K, p);
        carve_coarse(p, ws_dev, ws_bytes, &w);
        unsigned host[COARSE_Q];
        if (hipStreamSynchronize(as_stream(stream)) != hipSuccess ||
            hipMemcpy(host, w.last_m, sizeof(host), hipMemcpyDeviceToHost) != hipSuccess)
            return set_err(CLIPMI_EHIP, "hipMemcpy(gcnt)");
        long long tot = 0;
        for (int i = 0; i < Q; ++i) tot += host[i] < COARSE_CAP ? host[i] : COARSE_CAP;
        *survivors = tot;
    }
    return rc;
}

// Measurement hook (bench.py roofline): the same call sequence as clipmi_topk_ip, `reps` times,
// with HIP events recorded on `stream` around the MAIN scan kernel only; synchronises, and
// returns the average scan-kernel duration in milliseconds through *scan_ms. Q <= 32 (one pass).
extern "C" int clipmi_dbg_topk_scan_ms(const void* db_dev, int64_t N, int E, const float* q_dev, int Q, int K,
                                       float* out_score_dev, int64_t* out_id_dev, void* ws_dev, size_t ws_bytes,
                                       void* stream, int reps, float* scan_ms) {
    if (!scan_ms || reps < 1 || Q > 32) return set_err(CLIPMI_EINVAL, "dbg_topk_scan_ms: bad arguments");
    hipEvent_t ev[2];
    if (hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess)
        return set_err(CLIPMI_EHIP, "hipEventCreate");
    double total = 0.0;
    int rc = 0;
    for (int i = 0; i < reps && rc == 0; ++i) {
        rc = topk_ip_impl(db_dev, CLIPMI_F32, N, E, q_dev, Q, K, 0, out_score_dev, out_id_dev, ws_dev, ws_bytes, stream, ev);
        if (rc) break;
        if (hipEventSynchronize(ev[1]) != hipSuccess) { rc = set_err(CLIPMI_EHIP, "hipEventSynchronize"); break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ev[0], ev[1]);
        total += ms;
    }
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    if (rc == 0) *scan_ms = (float)(total / reps);
    return rc;
}

extern "C" size_t clipmi_merge_topk_workspace_bytes(int R, int Q, int K) {
    (void)R; (void)Q; (void)K;
    return 256;   // the merge works out of LDS; a non-zero size keeps callers' allocation paths uniform
}

static int merge_impl(const float* scores_dev, const int64_t* ids_dev, long long rs_s, long long rs_i, int R, int Q,
                      int K, float* out_score_dev, int64_t* out_id_dev, void* stream) {
    if (!scores_dev || !ids_dev || !out_score_dev || !out_id_dev) return set_err(CLIPMI_EINVAL, "merge_topk: NULL pointer");
    if (R < 1 || Q < 1 || K < 1) return set_err(CLIPMI_EINVAL, "merge_topk: R=%d Q=%d K=%d", R, Q, K);
    const size_t lds = (size_t)R * K * 12 + 16;
    if (lds > (size_t)LDS_LIMIT) return set_err(CLIPMI_EUNSUPPORTED, "merge_topk: R*K=%d too large for one LDS pass", R * K);
    if (int rc = opt_in_lds((const void*)merge_lists_i64_kernel, lds)) return rc;
    hipLaunchKernelGGL(merge_lists_i64_kernel, dim3(Q), dim3(256), lds, as_stream(stream), scores_dev,
                       (const long long*)ids_dev, rs_s, rs_i, R, Q, K, out_score_dev, (long long*)out_id_dev);
    CLIPMI_CHECK_LAUNCH("merge_lists_i64_kernel");
    return 0;
}

extern "C" int clipmi_merge_topk(const float* scores_dev, const int64_t* ids_dev, int R, int Q, int K,
                                 float* out_score_dev, int64_t* out_id_dev, void* ws_dev, size_t ws_bytes,
                                 void* stream) {
    (void)ws_dev; (void)ws_bytes;
    return merge_impl(scores_dev, ids_dev, (long long)Q * K, (long long)Q * K, R, Q, K, out_score_dev, out_id_dev, stream);
}

// Same merge over the buffer ONE all-gather produces when every rank contributes a packed record
// [scores f32 Q*K | pad to 8 B | ids i64 Q*K]: rank r's record starts at gathered_dev + r*record_bytes.
extern "C" int clipmi_merge_topk_packed(const void* gathered_dev, size_t record_bytes, int R, int Q, int K,
                                        float* out_score_dev, int64_t* out_id_dev, void* stream) {
    const size_t ids_off = align_up((size_t)Q * K * 4, 8);
    if (!gathered_dev || record_bytes < ids_off + (size_t)Q * K * 8 || record_bytes % 8 != 0)
        return set_err(CLIPMI_EINVAL, "merge_topk_packed: record_bytes %zu for Q=%d K=%d", record_bytes, Q, K);
    const char* base = static_cast<const char*>(gathered_dev);
    return merge_impl(reinterpret_cast<const float*>(base), reinterpret_cast<const int64_t*>(base + ids_off),
                      (long long)(record_bytes / 4), (long long)(record_bytes / 8), R, Q, K, out_score_dev, out_id_dev,
                      stream);
}
