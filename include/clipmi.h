/*
 * clipmi.h — C ABI of libclipmi.so, the MI355X (gfx950) CLIP index-and-search hot path.
 *
 * The reference (ps-auxw/CLI-P) has no FFI of its own: its "interface" for this path is the
 * set of Python calls its two scripts make on third-party objects. Each entry
 * point below names the reference call site it stands in for, as `file:line` under the
 * reference tree. The Python mirror of those call shapes lives in `cli-p_amd/`; the binding a
 * maintainer would add to the reference scripts is shown in INTEGRATION.md.
 *
 * Contract for every entry point:
 *   - plain C types only; every `*_dev` pointer is a DEVICE (HBM) pointer owned by the caller;
 *   - the library never allocates, frees or synchronises: all work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream) and is graph-capturable;
 *   - scratch memory is a caller-owned workspace whose size the matching *_workspace_bytes()
 *     function returns (0 from that function = unsupported arguments, see clipmi_last_error);
 *   - return value 0 = enqueued; non-zero = a CLIPMI_E* code, nothing was enqueued (or, for
 *     CLIPMI_EHIP, a launch failed); the message is in clipmi_last_error() (thread-local);
 *   - re-entrant: no global mutable state apart from that thread-local error string.
 */
#ifndef CLIPMI_H
#define CLIPMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLIPMI_ABI_VERSION 6

enum {
    CLIPMI_OK = 0,
    CLIPMI_EINVAL = 1,   /* bad argument (shape, dtype, NULL pointer, K out of range ...) */
    CLIPMI_EWORKSPACE = 2, /* workspace too small */
    CLIPMI_EHIP = 3,     /* a HIP call failed */
    CLIPMI_EUNSUPPORTED = 4
};

/* element types of buffers crossing the ABI */
enum {
    CLIPMI_F32 = 0,
    CLIPMI_BF16 = 1,
    CLIPMI_U8 = 2
};

/*
 * One transformer tower (vision or text) as a POD view into ONE packed device blob.
 * Produced on the host by the packer (cli-p_amd/weights.py) from tensors with OpenAI CLIP
 * state-dict names (SURVEY.md §8b "weight-file contract"); replaces the model object that
 * `clip.load("ViT-B/32", device=device, jit=False)` returns (build-index.py:18,
 * query-index.py:21). All offsets are bytes from the blob base, 256-byte aligned.
 * GEMM weights are bf16, row-major [out_features][in_features] (PyTorch Linear layout, so the
 * contraction index is contiguous for both operands); LayerNorm parameters, biases and
 * embeddings are f32.
 */
typedef struct clipmi_tower {
    int32_t abi_version;   /* CLIPMI_ABI_VERSION */
    int32_t kind;          /* 0 = vision, 1 = text */
    int32_t width;         /* W: 768 (ViT-B/32 vision), 512 (text) */
    int32_t layers;
    int32_t heads;         /* W / 64 */
    int32_t mlp;           /* 4 W */
    int32_t embed;         /* E: 512 */
    int32_t tokens;        /* L: 50 for ViT-B/32 (49 patches + CLS); 77 for text */
    /* vision only */
    int32_t patch;         /* P: 32 */
    int32_t res;           /* R: 224 */
    int32_t patch_k;       /* 3*P*P rounded up to a multiple of 64 (zero-padded columns) */
    /* text only */
    int32_t vocab;         /* 49408 */

    uint64_t blob_bytes;

    /* vision: conv1.weight as bf16 [W][patch_k]; class_embedding f32 [W];
       positional_embedding f32 [L][W]; ln_pre f32 */
    uint64_t off_patch_w, off_cls, off_pos, off_ln_pre_w, off_ln_pre_b;
    /* text: token_embedding.weight f32 [vocab][W]; positional_embedding uses off_pos */
    uint64_t off_tok_emb;

    /* per layer l: base = off_layers + l * layer_stride, then the intra-layer offsets */
    uint64_t off_layers, layer_stride;
    uint64_t lo_ln1_w, lo_ln1_b;      /* f32 [W] */
    uint64_t lo_qkv_w, lo_qkv_b;      /* attn.in_proj_weight bf16 [3W][W], in_proj_bias f32 [3W] */
    uint64_t lo_out_w, lo_out_b;      /* attn.out_proj bf16 [W][W], f32 [W] */
    uint64_t lo_ln2_w, lo_ln2_b;
    uint64_t lo_fc_w, lo_fc_b;        /* mlp.c_fc bf16 [4W][W], f32 [4W] */
    uint64_t lo_proj_w, lo_proj_b;    /* mlp.c_proj bf16 [W][4W], f32 [W] */

    /* vision: ln_post + visual.proj stored TRANSPOSED as bf16 [E][W];
       text: ln_final + text_projection stored TRANSPOSED as bf16 [E][W] */
    uint64_t off_ln_post_w, off_ln_post_b, off_out_proj;

    /* ABI 2. weight_format 0: the four linear layers of every block are bf16 (above). weight_format 1
       (BASELINE.json configs[4], FP8 matrix cores): lo_qkv_w / lo_out_w / lo_fc_w / lo_proj_w point at OCP e4m3
       bytes [out_features][in_features] and lo_*_s at f32 [out_features] per-output-channel scales
       (weight = scale * e4m3 value). Activations travel as e4m3 with MX block scales (one e8m0 scale 2^(e-7) per 32
       consecutive values of a row, written by the producing kernel: attention, the QuickGELU GEMM) or, where a producer
       sees whole rows (LayerNorm), with one f32 scale per row. Needs width % 256 == 0. */
    int32_t weight_format, ln_fold;
    uint64_t lo_qkv_s, lo_out_s, lo_fc_s, lo_proj_s;

    /* ABI 3. ln_fold = 1 (weight_format 0, width % 256 == 0): ln_1 / ln_2 are folded into the GEMMs that consume
       them. With mean / rstd the row statistics of the residual row x,
           LayerNorm(x; g, b) W^T + bias = rstd * (x Wg^T - mean * colsum) + cb,
       Wg = W diag(g) (lo_qkv_w / lo_fc_w then hold Wg, rounded to bf16 once), colsum[n] = sum_k Wg[n][k],
       cb[n] = sum_k b[k] W[n][k] + bias[n] (W = the bf16-rounded plain weights; sums in float64, stored f32 [3W] / [4W]).
       The residual stream is kept split x = hi + lo (two bf16 arrays); hi is the GEMM operand, the residual GEMM's
       store pass updates hi / lo and the statistics, and no stand-alone LayerNorm pass over the stream remains. */
    uint64_t lo_qkv_colsum, lo_qkv_cb, lo_fc_colsum, lo_fc_cb;
} clipmi_tower;

/* ---- a3/a4: model.encode_image(image) and the row L2-normalise that follows it ----------
 * Replaces build-index.py:49 (`model.encode_image(image)`) and, with normalize != 0,
 * build-index.py:50 (`image_features / image_features.norm(dim=-1, keepdim=True)`).
 *   pixels_dev : [B][3][R][R], NCHW. CLIPMI_F32 / CLIPMI_BF16 = already normalised (the output
 *                of the reference `transform`, build-index.py:48); CLIPMI_U8 = raw 0..255 RGB,
 *                the /255, -mean, /std of CLIP's transform is fused into the patch kernel.
 *   out_dev    : f32 [B][E] (the layout build-index.py:51 serialises: 2048 B per row for E=512)
 */
size_t clipmi_encode_image_workspace_bytes(const clipmi_tower* t, int B);
int clipmi_encode_image(const clipmi_tower* t, const void* blob_dev,
                        const void* pixels_dev, int pix_dtype, int B,
                        float* out_dev, int normalize,
                        void* ws_dev, size_t ws_bytes, void* stream);

/* ---- a9/a10: model.encode_text(texts) + normalize() -------------------------------------
 * Replaces query-index.py:108 (`model.encode_text(texts)`; `normalize`, query-index.py:13-17,
 * when normalize != 0 — per ROW here; identical to the reference for its Q = 1).
 *   ids_dev : int32 [Q][L] token ids as produced by clip.tokenize (query-index.py:107);
 *             the pooled row is the first argmax of each row (the EOT token).
 *   out_dev : f32 [Q][E]
 */
size_t clipmi_encode_text_workspace_bytes(const clipmi_tower* t, int Q);
int clipmi_encode_text(const clipmi_tower* t, const void* blob_dev,
                       const int32_t* ids_dev, int Q,
                       float* out_dev, int normalize,
                       void* ws_dev, size_t ws_bytes, void* stream);

/* ---- a12: index.search(features, k + offset + 1) ----------------------------------------
 * Replaces query-index.py:111 (`D, I = index.search(features, K)`), as EXACT flat inner-product
 * search over this rank's shard of the packed matrix that build-index.py:68-107 assembles.
 *   db_dev     : [N][E] row-major, CLIPMI_F32 (E in {512, 768})
 *   q_dev      : f32 [Q][E]
 *   out_score  : f32 [Q][K] descending; out_id: int64 [Q][K] = id_base + row; ties broken by
 *                ascending id; slots beyond min(K, N) hold score -FLT_MAX and id -1
 *   Scores are bit-exact f32: for each (row, query) one fmaf chain in the fixed order
 *   documented in DESIGN.md ("score order") and restated in oracle/topk_oracle.c.
 */
size_t clipmi_topk_ip_workspace_bytes(int64_t N, int E, int Q, int K);
int clipmi_topk_ip(const void* db_dev, int db_dtype, int64_t N, int E,
                   const float* q_dev, int Q, int K, int64_t id_base,
                   float* out_score_dev, int64_t* out_id_dev,
                   void* ws_dev, size_t ws_bytes, void* stream);

/* ---- a12, coarse-then-exact variant: the same exact result (bit-exact scores, same ordering rule) from a
 * bf16 coarse scan of `db_bf16_dev` (the matrix rounded to bf16, [N][E]) that keeps a provable superset,
 * followed by exact f32 re-scoring of the survivors from `db_dev` (DESIGN.md "coarse path"). Half the HBM
 * bytes per pass and 64 queries per pass. `rmax` = an upper bound of the largest row L2 norm of db_dev.
 * E = 512, N >= 65536; if a candidate list overflows, the exact scan runs as a device-side fallback. */
size_t clipmi_topk_ip_coarse_workspace_bytes(int64_t N, int E, int Q, int K);
int clipmi_topk_ip_coarse(const void* db_dev, const void* db_bf16_dev, int64_t N, int E, float rmax,
                          const float* q_dev, int Q, int K, int64_t id_base,
                          float* out_score_dev, int64_t* out_id_dev,
                          void* ws_dev, size_t ws_bytes, void* stream);

/* ---- a12, int8 coarse copy: the same contract with a quarter of the f32 bytes per pass.
 * clipmi_quantize_rows_i8 builds the copy (E a multiple of 32). Rows are taken in blocks of 32 that share one scale
 * s = (largest |x| of the block) / 127; q_rk = rint(x_rk / s). `out_i8_dev` (clipmi_i8_copy_bytes) holds the blocks as
 * [E / 32][64][16 bytes] - entry (k-step s, lane l) = bytes 32 s + 16 (l >> 5) .. + 15 of row (l & 31) of the block, the
 * register image of v_mfma_i32_32x32x32_i8's row operand, so a scanning wave reads whole contiguous KiB; rows past N in
 * the last block are zero. `meta_dev` (clipmi_i8_meta_bytes) receives meta[r] = (s, a_r) with a_r >= ||x_r - s q_r||_2
 * for N rounded up to 32 rows (+32), followed by one (s, largest a_r) pair per block and one u32 per slot: the row it holds.
 * clipmi_topk_ip_coarse_i8 scans the copy with integer MFMA (exact integer dot products) and keeps every row whose exact
 * score could reach the running K-th best, by |x.y - s t_q D| <= a_r ||y|| + (rmax + amax) ||y - t_q p_q||; survivors are
 * re-scored in exact f32 as above. `amax` >= every a_r, `rmax` >= every row norm. Up to 64 queries are one pass of the
 * copy (v_mfma_i32_16x16x64_i8, HBM-bound); MORE than 64 queries (query-index.py:111 is one call whatever Q) are taken
 * in chunks of <= 1024 as ONE pass each (v_mfma_i32_32x32x32_i8, query tiles of 256 resident in LDS, matrix-bound).
 * Workspace: clipmi_topk_ip_coarse_workspace_bytes. */
/* Build-side helpers of the coarse copies (index load: query-index.py:60-75 reads every vector once and keeps the
 * matrix). clipmi_rows_stats writes stats2_dev[0] = the largest row norm of db (f64 accumulation, rounded up: a valid
 * `rmax`) and stats2_dev[1] = the largest a_r of an int8 copy's meta (`meta_dev` may be NULL: 0). clipmi_rows_to_bf16
 * writes the bf16 copy (round to nearest even) clipmi_topk_ip_coarse scans. E a multiple of 4. Asynchronous on `stream`. */
int clipmi_rows_stats(const float* db_dev, int64_t N, int E, const float* meta_dev, float* stats2_dev, void* stream);
int clipmi_rows_to_bf16(const float* db_dev, int64_t N, int E, void* out_bf16_dev, void* stream);
size_t clipmi_i8_copy_bytes(int64_t N, int E);
size_t clipmi_i8_meta_bytes(int64_t N);
/* (ABI 4: the two buffer sizes are arguments - a copy or meta buffer smaller than clipmi_i8_copy_bytes / clipmi_i8_meta_bytes
 *  is refused with CLIPMI_EINVAL instead of written past.
 *  ABI 4, `perm_dev`: NULL, or a permutation of 0 .. N-1 (u32 [N]): slot t of the copy then holds row perm[t]. Order the rows by
 *  their largest |component| (clipmi_rows_absmax + any sort) and the 32 rows of a block share a scale that is nearly each
 *  row's own: error norms and with them the re-scored rows per query drop ~12 %. The search is exact for ANY permutation; it
 *  reports row ids - the meta buffer carries the slot -> row table behind the block meta.
 *  `perm_dev` is NOT validated (a check would need a pass and memory of its own): entries >= N become empty slots (memory-safe),
 *  duplicate or missing rows silently drop rows from every coarse search. Build it with clipmi_rows_order_by_absmax below, which
 *  returns a permutation by construction.) */
int clipmi_quantize_rows_i8(const float* db_dev, int64_t N, int E, const uint32_t* perm_dev, void* out_i8_dev, size_t out_i8_bytes,
                            float* meta_dev, size_t meta_bytes, void* stream);
/* out_dev[r] = largest |x_rk| of row r (f32 [N]); E a multiple of 4. Asynchronous on `stream`. */
int clipmi_rows_absmax(const float* db_dev, int64_t N, int E, float* out_dev, void* stream);
/* perm_dev[t] (u32 [N]) = the row that belongs in slot t of the int8 copy: the rows ordered by their largest |component|,
 * ascending, equal maxima in row order (a stable radix sort of clipmi_rows_absmax's values on the device; no framework sort is
 * needed to build the sorted copy). `ws_dev`: clipmi_rows_order_workspace_bytes(N) bytes. Asynchronous on `stream`; pass the
 * result to clipmi_quantize_rows_i8 as `perm_dev`. Replaces nothing in the reference (its IVF lists have no such order); the
 * index-load site is query-index.py:29. */
size_t clipmi_rows_order_workspace_bytes(int64_t N);
int clipmi_rows_order_by_absmax(const float* db_dev, int64_t N, int E, uint32_t* perm_dev, void* ws_dev, size_t ws_bytes, void* stream);
int clipmi_topk_ip_coarse_i8(const void* db_dev, const void* db_i8_dev, const float* meta_dev, float amax,
                             int64_t N, int E, float rmax, const float* q_dev, int Q, int K, int64_t id_base,
                             float* out_score_dev, int64_t* out_id_dev,
                             void* ws_dev, size_t ws_bytes, void* stream);

/* ---- multi-GPU merge of per-shard partial results (no reference counterpart: the reference
 * is single-process). Inputs are R lists per query as gathered by one
 * all-gather: scores f32 [R][Q][K], ids int64 [R][Q][K] (id -1 = empty slot). Same ordering
 * rule (score desc, id asc), so the result equals single-GPU exact top-K.
 */
size_t clipmi_merge_topk_workspace_bytes(int R, int Q, int K);
int clipmi_merge_topk(const float* scores_dev, const int64_t* ids_dev, int R, int Q, int K,
                      float* out_score_dev, int64_t* out_id_dev,
                      void* ws_dev, size_t ws_bytes, void* stream);

/* The same merge over the buffer ONE all-gather yields when every rank contributes a packed record
 * [scores f32 Q*K | pad to 8 B | ids int64 Q*K] of record_bytes (a multiple of 8): rank r's record is at
 * gathered_dev + r * record_bytes. Lets the N>1 search path be: top-k -> one collective -> merge. */
int clipmi_merge_topk_packed(const void* gathered_dev, size_t record_bytes, int R, int Q, int K,
                             float* out_score_dev, int64_t* out_id_dev, void* stream);

/* ---- a4/a10 stand-alone: rows of x[n][E] scaled to unit L2 norm in place (rows with
 * norm < 1e-9 are left unchanged, as query-index.py:13-17 does). */
int clipmi_l2_normalize_rows(float* x_dev, int64_t n, int E, void* stream);

/* ---- a2 on the device: Pillow's bicubic resize (shorter side -> n_px) + centre crop of 8-bit RGB images that were
 * shipped at full size, bit for bit what `transform(image)` computes before its float tail (build-index.py:48; Pillow's
 * two-pass 8-bit resampling with 22-bit integer coefficients). The host supplies, per image, a job record and the integer
 * coefficient blocks of the n_px outputs per axis that survive the crop ([n_px] first tap | [n_px] tap count | [n_px][k]
 * coefficients, int32): cli-p_amd/decode_worker.py computes them with Pillow's own float64 formula.
 * raw_dev: the images, HWC bytes at job.src_off; out_dev: uint8 [.][3][n_px][n_px] (block job.out_index);
 * scratch_dev: the 8-bit intermediate rows, job.tmp_off + nrows*n_px*3 bytes each; max_rows = the largest job.nrows. */
typedef struct clipmi_resize_job {
    int64_t src_off;              /* bytes from raw_dev to the image's first pixel (rows of w*3 bytes) */
    int32_t w, h;                 /* source size */
    int32_t r0, nrows;            /* source rows [r0, r0 + nrows) the output window needs */
    int32_t out_index;
    int32_t need_h, need_v;       /* 0: that axis is not resampled (source size == target size), only cropped */
    int32_t left, top;            /* window origin on an axis that is not resampled */
    int32_t hk, vk;               /* taps per output of the horizontal / vertical coefficient block */
    int64_t hcoef_off, vcoef_off; /* int32 offsets into coef_dev */
    int64_t tmp_off;              /* bytes into scratch_dev */
} clipmi_resize_job;
int clipmi_resize_crop_rgb8(const void* raw_dev, const void* jobs_dev, int njobs, int max_rows, const int32_t* coef_dev,
                            int n_px, void* out_dev, void* scratch_dev, void* stream);

/* ---- `Image.open(tfn)` on the device for baseline JPEG files (build-index.py:47, the decode in front of `transform` at :48;
 * SURVEY.md §8(f) next-1): the bytes Pillow (libjpeg-turbo: Huffman decode, jpeg_idct_islow, fancy upsampling, 16-bit
 * fixed-point YCbCr -> RGB) hands to the transform, bit for bit. The HOST walks the markers (cli-p_amd/jpeg.py) and lets through
 * 8-bit baseline / extended-sequential Huffman files with one interleaved scan, 1 (grey) or 3 (YCbCr) components, luma sampling
 * 1x1, 2x1 or 2x2 with 1x1 chroma, with or without restart intervals; everything else stays with Pillow. Per image one record; the
 * entropy-coded segments travel with the 0xFF00 stuffing removed, each 4-byte aligned and followed by at least 16 zero bytes.
 * tables_dev: the batch's distinct Huffman tables, 288 bytes each (DHT's 16 counts + up to 256 symbols, zero padded to 272,
 * then the table class - 0 DC, 1 AC - and 15 zero bytes).
 * out_dev: per image height rows of width*3 RGB bytes at out_off (grey files replicated, as Image.convert("RGB") does) - the
 * layout clipmi_resize_crop_rgb8 takes. status_dev[i]: 0 decoded; 1 invalid Huffman code; 2 the data ended early or ran
 * over; 3 a marker inside a segment handed over with its stuffing - such a file goes back to Pillow, whose error handling is the reference's. total_blocks = sum of the images' 8x8 blocks
 * (coef_off counts in blocks), max_blocks / max_pixels = the largest image's. */
typedef struct clipmi_jpeg_image {
    int64_t stream_off;           /* bytes from streams_dev */
    int64_t coef_off;             /* the image's first block in the workspace's coefficient / sample buffers */
    int64_t out_off;              /* bytes from out_dev */
    int64_t intervals_off;        /* restart intervals: bytes from streams_dev to n_intervals uint32 byte offsets into the segment */
    int32_t stream_bytes;
    int32_t width, height;
    int32_t ncomp;                /* 1 or 3 */
    int32_t hs, vs;               /* luma sampling factors (1,1) (2,1) (2,2) */
    int32_t dc_tbl[3], ac_tbl[3]; /* per component: index into tables_dev */
    int32_t restart_interval;     /* MCUs per restart interval (DRI), 0 = none */
    int32_t n_intervals;          /* ceil(MCUs / restart_interval); the RSTn markers themselves are removed from the segment */
    int32_t stuffed;              /* 1: the segment still holds the 0xFF00 byte stuffing (no restart intervals then) - the device
                                     removes it IN PLACE and rewrites stream_bytes and this field (0 done, 2 marker inside the scan) */
    int32_t reserved;
    uint8_t quant[3][64];         /* per component: quantisation steps, natural (row-major) order */
} clipmi_jpeg_image;
int64_t clipmi_jpeg_workspace_bytes(int64_t total_blocks, int ntables);
int clipmi_jpeg_decode_rgb8(void* streams_dev, void* images_dev, int n, const void* tables_dev, int ntables,
                            int64_t total_blocks, int64_t max_blocks, int64_t max_pixels, void* out_dev, int32_t* status_dev,
                            void* ws_dev, int64_t ws_bytes, void* stream);

/* thread-local message of the last failing call on this thread ("" if none) */
const char* clipmi_last_error(void);
int clipmi_abi_version(void);

/* ---- test/bench hooks: the individual kernels behind encode_*, exported so that parity
 * tests and bench.py can launch and time them one at a time. Not part of the drop-in surface.
 */
/* C[M][N] = A[M][K](bf16) . W[N][K]^T(bf16) with fused epilogue `epi`:
 *   0: out bf16 = acc + bias        1: out bf16 = quickgelu(acc + bias)
 *   2: out f32 += acc + bias (residual, in place)    3: out f32 = acc (bias may be NULL)  */
int clipmi_dbg_gemm_bf16(const void* a_dev, const void* w_dev, const float* bias_dev,
                         void* out_dev, int M, int N, int K, int epi, void* stream);
/* rows of f32 x[M][W] -> LayerNorm(eps 1e-5) -> bf16 (out_bf16 != 0) or f32 */
int clipmi_dbg_layernorm(const float* x_dev, const float* w_dev, const float* b_dev,
                         void* out_dev, int M, int W, int out_bf16, void* stream);
/* qkv bf16 [B*L][3W] -> softmax(q k^T / 8 (+causal)) v -> bf16 [B*L][W], head dim 64 */
int clipmi_dbg_attention(const void* qkv_dev, void* out_dev, int B, int L, int heads,
                         int causal, void* stream);

/* clipmi_topk_ip's launch sequence `reps` times with HIP events around the MAIN scan kernel on
 * `stream`; synchronises; *scan_ms = average duration of that kernel (bench.py roofline). Q <= 16 */
int clipmi_dbg_topk_scan_ms(const void* db_dev, int64_t N, int E, const float* q_dev, int Q, int K,
                            float* out_score_dev, int64_t* out_id_dev, void* ws_dev, size_t ws_bytes,
                            void* stream, int reps, float* scan_ms);

/* clipmi_topk_ip_coarse `reps` times with HIP events around the bf16 scan kernel (Q <= 64) */
int clipmi_dbg_topk_coarse_scan_ms(const void* db_dev, const void* db_bf16_dev, int64_t N, int E, float rmax,
                                   const float* q_dev, int Q, int K, float* out_score_dev, int64_t* out_id_dev,
                                   void* ws_dev, size_t ws_bytes, void* stream, int reps, float* scan_ms,
                                   long long* survivors /* optional: rows kept by the coarse pass, summed over Q */);

int clipmi_dbg_topk_coarse_i8_scan_ms(const void* db_dev, const void* db_i8_dev, const float* meta_dev, float amax,
                                      int64_t N, int E, float rmax, const float* q_dev, int Q, int K,
                                      float* out_score_dev, int64_t* out_id_dev, void* ws_dev, size_t ws_bytes,
                                      void* stream, int reps, float* scan_ms, long long* survivors);

/* the FP8 path's two kernels alone: bf16 [M][K] -> e4m3 [M][K] + f32 row scales; C = a_scale w_scale (A8 W8^T) + epilogue
 * `epi` (0 bias -> bf16, 1 bias + QuickGELU -> bf16, 2 bias + residual into f32 out, 3 f32) */
int clipmi_dbg_quantize_rows_fp8(const void* in_bf16_dev, void* out_fp8_dev, float* scale_dev, int M, int K, void* stream);
int clipmi_dbg_gemm_fp8(const void* a8_dev, const void* w8_dev, const float* a_scale_dev, const float* w_scale_dev,
                        const float* bias_dev, void* out_dev, int M, int N, int K, int epi, void* stream);

/* LN-folded linear layers (tower ABI 3, csrc/gemm.hpp), kernel by kernel:
 * The residual stream is kept SPLIT (3 bytes per element since round 5 / ABI 5): a row of width W is W bf16 values hi = bf16(x)
 * followed by W biased 8-bit remainders u, bits(x') = (bits(hi) << 16) + (u << 8) - 0x8000 - x to 15 mantissa bits; `x3` buffers are [M][3 W] bytes.
 *   split_stats   rows = add ? x + x3 : x (f32 [M][W]) -> x3 = the split rows, part [M][W/256][2]
 *                 = per-256-column (sum, sum of squares)                                           (W % 256 == 0, <= 1024)
 *   gemm_ln       out bf16 = [quick_gelu] (rstd * (hi(x3) wg^T - mean * colsum) + cb), (mean, rstd) from part; epi 5 | 6,
 *                 bits 8-9 force a kernel
 *   gemm_resid_ln x3 += a w^T + bias, part of the new rows; tmp = f32 [M][N] scratch; algo 3 = the persistent
 *                 kernel's fused store pass, else GEMM into tmp + split_stats(add) */
int clipmi_dbg_split_stats(const float* x_dev, int add, void* x3_dev, float* part_dev, int M, int W, void* stream);
int clipmi_dbg_gemm_ln(const void* x3_dev, const void* wg_dev, const float* cb_dev, const float* colsum_dev,
                       const float* part_dev, void* out_dev, int M, int N, int K, int epi, void* stream);
/* MX block scales (round 3): e4m3 activations with one e8m0 scale per 32 consecutive values of a row, 2^(e - 7) with
 * e = floor(log2(largest magnitude of the block)). clipmi_dbg_quantize_rows_fp8mx: bf16 [M][K] -> e4m3 [M][K] + scale
 * bytes [M][K / 32] (K a multiple of 32). clipmi_dbg_gemm_fp8_bsa: out = w_scale[n] * sum_k (scaled A8)[m][k] W8[n][k]
 * + bias (+ out for epi 2; epi 3 = plain f32) on the block-scaled matrix-core instruction; a_bscale_dev holds
 * ceil(M / 256) * 256 rows of K / 32 bytes. */
int clipmi_dbg_quantize_rows_fp8mx(const void* in_bf16_dev, void* out_fp8_dev, void* bscale_dev, int M, int K, void* stream);
int clipmi_dbg_gemm_fp8_bsa(const void* a8_dev, const void* w8_dev, const void* a_bscale_dev, const float* w_scale_dev,
                            const float* bias_dev, void* out_dev, int M, int N, int K, int epi, void* stream);
/* One prompt / one image (M <= 128 rows, csrc/gemm_skinny.hpp, round 5): the residual GEMM updates the split rows in place and writes
 * the statistics LEAVES leaf_dev [M][N / 4][2] ((sum, sum of squares) of every 4-column group) instead of running a split /
 * statistics pass; the LN-folded consumer behind it reads them and runs the canonical reduction tree itself. Bit for bit
 * clipmi_dbg_gemm_resid_ln followed by clipmi_dbg_gemm_ln. */
int clipmi_dbg_gemm_resid_ln_leaf(const void* a_dev, const void* w_dev, const float* bias_dev, void* x3_dev, float* leaf_dev,
                                  int M, int N, int K, void* stream);
int clipmi_dbg_gemm_ln_leaf(const void* x3_dev, const void* wg_dev, const float* cb_dev, const float* colsum_dev,
                            const float* leaf_dev, void* out_dev, int M, int N, int K, int epi, void* stream);
int clipmi_dbg_gemm_resid_ln(const void* a_dev, const void* w_dev, const float* bias_dev, void* x3_dev, float* part_dev,
                             float* tmp_dev, int M, int N, int K, int algo, void* stream);

/* clipmi_encode_image `reps` times with HIP events around every launch of the GEMM whose
 * epilogue is `probe_epi` (1 = MLP c_fc + QuickGELU), on `stream`; synchronises;
 * *kernel_ms = average duration of the timed launches (bench.py roofline) */
int clipmi_dbg_encode_image_probe_ms(const clipmi_tower* t, const void* blob_dev, const void* pixels_dev,
                                     int pix_dtype, int B, float* out_dev, void* ws_dev, size_t ws_bytes,
                                     void* stream, int probe_epi, int reps, float* kernel_ms, int* launches);

/* The same with three estimators of the kernel's in-situ duration, ms3[0..2] (csrc/encode.hip encode_probe): begin -> end
 * events of the launch itself (reads long in a back-to-back stream), plain event in front -> end event, and end event of the
 * GEMM directly in front -> end event of this one (completion to completion; 0 when not applicable). *kernel_kind: 0 / 1 / 2 =
 * gemm_bf16_nt_kernel / gemm256_bf16_nt_kernel / gemm256p_bf16_nt_kernel; *epi_ran: its EPI template argument.
 * probe_epi bits 8 and up, when non-zero, restrict the probe to launches with that K (attn.out_proj and mlp.c_proj share the
 * residual epilogue: 2 | (3072 << 8) times c_proj alone). */
int clipmi_dbg_encode_image_probe3_ms(const clipmi_tower* t, const void* blob_dev, const void* pixels_dev,
                                      int pix_dtype, int B, float* out_dev, void* ws_dev, size_t ws_bytes,
                                      void* stream, int probe_epi, int reps, float* ms3, int* launches,
                                      int* kernel_kind, int* epi_ran);

#ifdef __cplusplus
}
#endif
#endif /* CLIPMI_H */
