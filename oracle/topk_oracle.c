/*
 * topk_oracle.c — TEST INFRASTRUCTURE, not product code.
 *
 * CPU restatement of the exact flat inner-product top-K that stands in for
 * `D, I = index.search(features, k + offset + 1)` (reference query-index.py:111) over the packed
 * matrix of build-index.py:68-107. The reference delegates this call to faiss (un-vendored,
 * unpinned git HEAD: reference setup.sh:12-18) and holds no test or golden vector for it, so
 * PARITY IS UNPINNED against faiss itself; what this oracle pins is the contract BASELINE.json
 * states: "returned top-k indices are bit-exact against a CPU brute-force over the same vectors".
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call this.
 *
 * Score order (must equal cli-p_amd/csrc/topk.hip): E = 16*NT,
 *     acc = 0; for t in [0,NT) for c in [0,4) for g in [0,4): k = 16t + 4g + c;
 *         acc = fmaf(db[r][k], q[k], acc)
 * Ordering: score descending, then id ascending; NaN scores are never returned; slots beyond
 * the number of valid rows hold score -FLT_MAX and id -1 (faiss's IndexFlat pads with id -1).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

__attribute__((target_clones("fma", "default")))
void topk_oracle_scores(const float* db, int64_t n, int E, const float* q, float* out) {
    const int NT = E / 16;
    for (int64_t r = 0; r < n; ++r) {
        const float* row = db + r * (int64_t)E;
        float acc = 0.0f;
        for (int t = 0; t < NT; ++t)
            for (int c = 0; c < 4; ++c)
                for (int g = 0; g < 4; ++g) {
                    const int k = 16 * t + 4 * g + c;
                    acc = fmaf(row[k], q[k], acc);
                }
        out[r] = acc;
    }
}

typedef struct { float s; int64_t id; } cand_t;

static int cand_cmp(const void* pa, const void* pb) {
    const cand_t* a = (const cand_t*)pa;
    const cand_t* b = (const cand_t*)pb;
    if (a->s > b->s) return -1;
    if (a->s < b->s) return 1;
    return (a->id > b->id) - (a->id < b->id);
}

/* one query: every finite score, sorted by the rule, the first K kept (pads: -FLT_MAX / -1) */
static void topk_one(const float* db, int64_t N, int E, const float* q, int K, int64_t id_base, float* sc, cand_t* cd,
                     float* out_s, int64_t* out_i) {
    topk_oracle_scores(db, N, E, q, sc);
    int64_t m = 0;
    for (int64_t r = 0; r < N; ++r)
        if (sc[r] == sc[r]) { cd[m].s = sc[r]; cd[m].id = id_base + r; ++m; }
    qsort(cd, (size_t)m, sizeof(cand_t), cand_cmp);
    for (int k = 0; k < K; ++k) {
        if (k < m) { out_s[k] = cd[k].s; out_i[k] = cd[k].id; }
        else { out_s[k] = -FLT_MAX; out_i[k] = -1; }
    }
}

/* returns 0 on success */
int topk_oracle(const float* db, int64_t N, int E, const float* q, int Q, int K, int64_t id_base,
                float* out_s, int64_t* out_i) {
    if (E % 16 != 0 || N < 0 || Q < 1 || K < 1) return 1;
    float* sc = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    cand_t* cd = (cand_t*)malloc(sizeof(cand_t) * (size_t)(N > 0 ? N : 1));
    if (!sc || !cd) { free(sc); free(cd); return 2; }
    for (int qi = 0; qi < Q; ++qi)
        topk_one(db, N, E, q + (size_t)qi * E, K, id_base, sc, cd, out_s + (size_t)qi * K, out_i + (size_t)qi * K);
    free(sc); free(cd);
    return 0;
}

/* the same, the queries dealt over `threads` threads (queries are independent: identical results; the test suites use it so
 * that checking 1 024 queries against 200 k rows does not take a core's ten seconds; bench.py's cpu_baseline keeps topk_oracle) */
int topk_oracle_mt(const float* db, int64_t N, int E, const float* q, int Q, int K, int64_t id_base,
                   float* out_s, int64_t* out_i, int threads) {
    if (E % 16 != 0 || N < 0 || Q < 1 || K < 1) return 1;
    if (threads < 1) threads = 1;
    int rc = 0;
#pragma omp parallel num_threads(threads)
    {
        float* sc = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
        cand_t* cd = (cand_t*)malloc(sizeof(cand_t) * (size_t)(N > 0 ? N : 1));
        if (!sc || !cd) {
#pragma omp atomic write
            rc = 2;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int qi = 0; qi < Q; ++qi)
                topk_one(db, N, E, q + (size_t)qi * E, K, id_base, sc, cd, out_s + (size_t)qi * K, out_i + (size_t)qi * K);
        }
        free(sc); free(cd);
    }
    return rc;
}

/* merge of R per-shard lists, same rule (restates clipmi_merge_topk) */
int topk_oracle_merge(const float* scores, const int64_t* ids, int R, int Q, int K, float* out_s, int64_t* out_i) {
    cand_t* cd = (cand_t*)malloc(sizeof(cand_t) * (size_t)R * K);
    if (!cd) return 2;
    for (int qi = 0; qi < Q; ++qi) {
        int m = 0;
        for (int r = 0; r < R; ++r)
            for (int k = 0; k < K; ++k) {
                const size_t src = ((size_t)r * Q + qi) * K + k;
                if (ids[src] >= 0 && scores[src] == scores[src]) { cd[m].s = scores[src]; cd[m].id = ids[src]; ++m; }
            }
        qsort(cd, (size_t)m, sizeof(cand_t), cand_cmp);
        for (int k = 0; k < K; ++k) {
            if (k < m) { out_s[(size_t)qi * K + k] = cd[k].s; out_i[(size_t)qi * K + k] = cd[k].id; }
            else { out_s[(size_t)qi * K + k] = -FLT_MAX; out_i[(size_t)qi * K + k] = -1; }
        }
    }
    free(cd);
    return 0;
}
