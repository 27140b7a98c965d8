// encode.hip — clipmi_encode_image / clipmi_encode_text: the two towers as kernel sequences on
// one stream. Replaces model.encode_image (reference build-index.py:49-50) and
// model.encode_text + normalize (query-index.py:108, 13-17). Op order follows SURVEY.md §2.1.
//
// Activation layout in HBM (all row-major, token row t = b*L + l):
//   ln_fold towers (bf16 weights, W % 256 == 0: every real CLIP tower; gemm.hpp "LN-folded linear layers"):
//     x3   [B*L][3 W bytes]    the residual stream, split (gemm.hpp split_make): a row is W bf16 hi = bf16(x) then W u8
//                              remainders (3 bytes per element since round 5); the hi halves are the A operand of the
//                              LN-folded qkv / c_fc GEMMs (row pitch 3 W), 12-24 residual adds keep ~2^-17 relative
//     ln_part f32 [B*L][W/256][2]  row statistics of x as per-256-column (sum, sum of squares)
//     x    f32  [B*L][W]       embedding-stage output only (and scratch of non-persistent residual GEMMs)
//   other towers (FP8 weights, toy widths):
//     x    f32  [B*L][W]       residual stream;  h also receives the LayerNorm outputs
//   h    bf16 [B*L][W]     attention output (GEMM A operand)
//   big  bf16 [B*L][4W]    qkv ([..][3W]) and, later in the layer, the MLP hidden ([..][4W])
//   patches bf16 [B*np][patch_k] (vision), pooled bf16 [B][W]
#include "vit_kernels.hpp"

namespace clipmi {
namespace {

struct Ws {
    unsigned short* patches;
    float* x;
    unsigned short* h;
    unsigned short* big;
    unsigned short* pooled;
    int* rowidx;
    unsigned char* a8;        // weight_format 1: the current GEMM's A operand as e4m3 [B*L][<= 4W]
    float* a_scale;           //                  and its row scales [B*L]
    unsigned char* a_bs;      //                  or (A produced by attention / QuickGELU) its MX block scales, rows padded to 256
    unsigned char* x8;        // weight_format 1 + ln_fold: the residual rows as e4m3 [B*L][W] (A operand of the LN-folded qkv / c_fc GEMMs)
    unsigned char* x8_bs;     //                            and their MX block scales, rows padded to 256
    void* x3;                 // ln_fold: the residual stream kept split, rows of [W bf16 hi | W u8 lo]; the hi halves are
                              //          also the A operand of the LN-folded qkv / c_fc GEMMs
    float* ln_part;           //          row statistics of x as per-256-column (sum, sum of squares) [B*L][W/256][2]
    float* ln_leaf;           //          <= 128 rows (one prompt, one image): the statistics as per-4-column leaves [B*L][W/4][2]
};

size_t carve(const clipmi_tower* t, int B, void* base, size_t cap, Ws* out) {
    Arena ar(base ? base : reinterpret_cast<void*>(256), cap);
    const size_t rows = (size_t)B * t->tokens;
    const int W = t->width;
    Ws w{};
    if (t->kind == 0) w.patches = ar.take<unsigned short>((size_t)B * (t->tokens - 1) * t->patch_k);
    w.x = ar.take<float>(rows * W);
    w.h = ar.take<unsigned short>(rows * W);
    w.big = ar.take<unsigned short>(rows * 4 * W);
    w.pooled = ar.take<unsigned short>((size_t)B * W);
    w.rowidx = ar.take<int>((size_t)B);
    if (t->weight_format == 1) {
        w.a8 = ar.take<unsigned char>(rows * 4 * W);
        w.a_scale = ar.take<float>(rows);
        w.a_bs = ar.take<unsigned char>((rows + 255) / 256 * 256 * (size_t)(4 * W / 32));
    }
    if (t->ln_fold && t->weight_format == 1) {
        // FP8 tower with folded LayerNorms: the residual stays f32 (w.x); its e4m3 image + statistics travel beside it
        w.x8 = ar.take<unsigned char>(rows * W);
        w.x8_bs = ar.take<unsigned char>((rows + 255) / 256 * 256 * (size_t)(W / 32));
        w.ln_part = ar.take<float>(rows * 2 * (W / 256));
    } else if (t->ln_fold) {
        w.x3 = ar.take<unsigned char>(rows * resid_row_bytes(W));
        w.ln_part = ar.take<float>(rows * 2 * (W / 256));
        if (rows <= (size_t)SKINNY_MAX_M) w.ln_leaf = ar.take<float>(rows * 2 * (W / 4));
    }
    if (out) *out = w;
    return ar.off + 256;
}

int check_tower(const clipmi_tower* t, int kind, const char* who) {
    if (!t) return set_err(CLIPMI_EINVAL, "%s: NULL tower", who);
    if (t->abi_version != CLIPMI_ABI_VERSION)
        return set_err(CLIPMI_EINVAL, "%s: tower ABI %d != %d", who, t->abi_version, CLIPMI_ABI_VERSION);
    if (t->kind != kind) return set_err(CLIPMI_EINVAL, "%s: tower kind %d", who, t->kind);
    if (t->width % 128 != 0 || t->heads * 64 != t->width || t->mlp != 4 * t->width || t->embed % 128 != 0 ||
        t->width > 1024 || t->layers < 1)
        return set_err(CLIPMI_EUNSUPPORTED,
                       "%s: width=%d heads=%d mlp=%d embed=%d (need width %% 128 == 0 <= 1024, head dim 64, embed %% 128 == 0)",
                       who, t->width, t->heads, t->mlp, t->embed);
    if (t->weight_format != 0 && t->weight_format != 1)
        return set_err(CLIPMI_EINVAL, "%s: weight_format %d", who, t->weight_format);
    if (t->weight_format == 1 && t->width % 256 != 0)
        return set_err(CLIPMI_EUNSUPPORTED, "%s: fp8 weights need width %% 256 == 0 (width %d)", who, t->width);
    if (t->ln_fold != 0 && (t->ln_fold != 1 || t->width % 256 != 0))
        return set_err(CLIPMI_EINVAL, "%s: ln_fold %d needs width %% 256 == 0 (width %d)", who, t->ln_fold, t->width);
#ifndef CLIPMI_DEV
    if (t->ln_fold != 0 && t->weight_format != 0)
        return set_err(CLIPMI_EUNSUPPORTED, "%s: folded LayerNorms on FP8 weights exist in the development library only", who);
#endif
    if (t->tokens > 80 && kind == 1)
        return set_err(CLIPMI_EUNSUPPORTED, "%s: %d tokens (causal attention covers <= 80)", who, t->tokens);
    return 0;
}

template <typename T>
const T* at(const void* blob, uint64_t off) {
    return reinterpret_cast<const T*>(static_cast<const char*>(blob) + off);
}

// the 12 (or 24) residual attention blocks shared by both towers. weight_format 1 (BASELINE.json configs[4]): the
// four linear layers of a block run on the FP8 matrix cores - weights are e4m3 with one scale per output channel
// (packed by weights.py), the bf16 activation rows are quantised to e4m3 with one scale per row right before each
// GEMM (quantize_rows_fp8_kernel), gemm256f8 applies both scales in its epilogue. LayerNorm, attention, the
// residual stream, patch embedding and the final projection are unchanged.
int run_layers(const clipmi_tower* t, const void* blob, const Ws& w, int B, int causal, hipStream_t st, GemmProbe* probe) {
    const int W = t->width, L = t->tokens, M = B * L;
    const bool fp8 = t->weight_format == 1;
    auto linear = [&](const unsigned short* A, int K, uint64_t w_off, uint64_t b_off, void* out, int N, int epi) -> int {
        GemmArgs g{};
        g.bias = at<float>(blob, b_off); g.out = out; g.M = M; g.N = N; g.K = K;
        g.A = A; g.W = at<unsigned short>(blob, w_off);
        return launch_gemm_algo(g, epi, 0, st, probe);
    };
#ifdef CLIPMI_DEV
    if (t->ln_fold && fp8) {
        // FP8 blocks with folded LayerNorms (round 4; VERDICT r03 "What's missing" #1; development library: parity-green and
        // measured SLOWER than the LayerNorm-pass tower below - DESIGN 4.4c - while its GEMMs run on the non-persistent kernel). The residual stream stays f32 in w.x;
        // every producer of residual rows (the embedding stage through rows_mx_stats_kernel, then the two residual GEMMs'
        // store passes: EPI_BIAS_RESID_LN8) also leaves them as e4m3 with MX block scales (w.x8 / w.x8_bs) with their
        // row-statistics partials (w.ln_part), and qkv / c_fc take THOSE as their A operand with the LN-folded epilogue
        // rstd (acc w_scale - mean colsum) + cb on weights e4m3(W diag(gamma)). No stand-alone LayerNorm pass is left
        // (round 3: two per layer, 26 x 30 us of a 7.37 ms step).
        auto lnfold8 = [&](uint64_t w_off, uint64_t s_off, uint64_t cb_off, uint64_t cs_off, void* out, unsigned char* out_bs, int N,
                           int epi) -> int {
            GemmArgs g{};
            g.A = reinterpret_cast<const unsigned short*>(w.x8); g.a_bscale = w.x8_bs;
            g.W = at<unsigned short>(blob, w_off); g.w_scale = at<float>(blob, s_off);
            g.bias = at<float>(blob, cb_off); g.colsum = at<float>(blob, cs_off); g.ln_part_in = w.ln_part;
            g.out = out; g.out_bscale = out_bs; g.M = M; g.N = N; g.K = W;
            return launch_gemm_fp8(g, epi, st);
        };
        auto resid8 = [&](const unsigned char* A8, int K, uint64_t w_off, uint64_t s_off, uint64_t b_off) -> int {
            GemmArgs g{};
            g.A = reinterpret_cast<const unsigned short*>(A8); g.a_bscale = w.a_bs;
            g.W = at<unsigned short>(blob, w_off); g.w_scale = at<float>(blob, s_off); g.bias = at<float>(blob, b_off);
            g.out = w.x; g.x8 = w.x8; g.x8_bs = w.x8_bs; g.ln_part = w.ln_part; g.M = M; g.N = W; g.K = K;
            return launch_gemm_fp8(g, EPI_BIAS_RESID_LN8, st);
        };
        unsigned char* const big8 = reinterpret_cast<unsigned char*>(w.big);
        static const bool fuse_off = dev_knob("CLIPMI_FP8_FUSE", 1) == 0;   // A/B aid (development build): stand-alone MX passes
        for (int l = 0; l < t->layers; ++l) {
            const uint64_t lb = t->off_layers + (uint64_t)l * t->layer_stride;
            if (int rc = lnfold8(lb + t->lo_qkv_w, lb + t->lo_qkv_s, lb + t->lo_qkv_cb, lb + t->lo_qkv_colsum, w.big, nullptr, 3 * W, EPI_LN_BIAS_BF16)) return rc;
            bool fused = false;
            if (int rc = launch_attention(w.big, w.h, B, L, t->heads, causal, 1, st, fuse_off ? nullptr : w.a8, w.a_bs, &fused)) return rc;
            if (!fused)
                if (int rc = launch_quantize_rows_fp8mx(w.h, w.a8, w.a_bs, M, W, st)) return rc;
            if (int rc = resid8(w.a8, W, lb + t->lo_out_w, lb + t->lo_out_s, lb + t->lo_out_b)) return rc;
            if (!fuse_off) {
                // the QuickGELU rows leave c_fc as e4m3 + block scales, into the bf16 buffer's bytes
                if (int rc = lnfold8(lb + t->lo_fc_w, lb + t->lo_fc_s, lb + t->lo_fc_cb, lb + t->lo_fc_colsum, big8, w.a_bs, 4 * W, EPI_LN_BIAS_QGELU_BF16)) return rc;
                if (int rc = resid8(big8, 4 * W, lb + t->lo_proj_w, lb + t->lo_proj_s, lb + t->lo_proj_b)) return rc;
            } else {
                if (int rc = lnfold8(lb + t->lo_fc_w, lb + t->lo_fc_s, lb + t->lo_fc_cb, lb + t->lo_fc_colsum, w.big, nullptr, 4 * W, EPI_LN_BIAS_QGELU_BF16)) return rc;
                if (int rc = launch_quantize_rows_fp8mx(w.big, w.a8, w.a_bs, M, 4 * W, st)) return rc;
                if (int rc = resid8(w.a8, 4 * W, lb + t->lo_proj_w, lb + t->lo_proj_s, lb + t->lo_proj_b)) return rc;
            }
        }
        return 0;
    }
#endif
    if (t->ln_fold) {
        // LN-folded blocks (gemm.hpp): ln_1 / ln_2 never run as passes of their own and the residual stream lives split
        // in w.x3 with its row statistics in w.ln_part; w.x (f32) is only the embedding stage's output and the
        // scratch of residual GEMMs that run on a non-persistent kernel. lo_qkv_w / lo_fc_w hold W * diag(ln weight).
        auto lb_of = [&](int l) { return t->off_layers + (uint64_t)l * t->layer_stride; };
        // One prompt / one image (M <= 128: the skinny kernels, ~90 dependent launches of ~4 us): the residual GEMMs update the
        // split rows themselves and hand the statistics on as leaves (GemmArgs.ln_leaf; round 5: 24 split / statistics launches
        // fewer per tower). `leaves`: w.ln_leaf, not w.ln_part, describes the rows in w.x3.
        const bool leaf_mode = w.ln_leaf != nullptr && gemm_resid_writes_leaves(M, W, W) && gemm_resid_writes_leaves(M, W, 4 * W);
        bool leaves = false;
        auto ln_linear = [&](uint64_t w_off, uint64_t cb_off, uint64_t cs_off, int N, int epi) -> int {
            GemmArgs g{};
            g.A = static_cast<const unsigned short*>(w.x3); g.lda_bytes = (unsigned)resid_row_bytes(W);
            g.W = at<unsigned short>(blob, w_off); g.bias = at<float>(blob, cb_off);
            g.colsum = at<float>(blob, cs_off); g.ln_part_in = w.ln_part;
            if (leaves) g.ln_leaf_in = w.ln_leaf;
            g.out = w.big; g.M = M; g.N = N; g.K = W;
            return launch_gemm_algo(g, epi, 0, st, probe);
        };
        auto resid_linear = [&](const unsigned short* A, int K, uint64_t w_off, uint64_t b_off) -> int {
            GemmArgs g{};
            g.A = A; g.W = at<unsigned short>(blob, w_off); g.bias = at<float>(blob, b_off);
            g.M = M; g.N = W; g.K = K;
            g.x3 = w.x3; g.ln_part = w.ln_part; g.tmp_f32 = w.x;
            if (leaf_mode) { g.ln_leaf = w.ln_leaf; leaves = true; }
            return launch_gemm_algo(g, EPI_BIAS_RESID_LN_F32, 0, st, probe);
        };
        // (the caller left the embedded rows split: ln_pre writes hi / lo / partials itself; the text tower, which has no
        //  LayerNorm in front of the blocks, runs split_stats_kernel over its embedded rows)
        for (int l = 0; l < t->layers; ++l) {
            const uint64_t lb = lb_of(l);
            if (int rc = ln_linear(lb + t->lo_qkv_w, lb + t->lo_qkv_cb, lb + t->lo_qkv_colsum, 3 * W, EPI_LN_BIAS_BF16)) return rc;
            // (bench probe, mode 2: attention's own end stamp, so that out_proj gets a completion-to-completion time too)
            hipEvent_t* aev = nullptr;
            if (probe && probe->mode == 2 && probe->n < GemmProbe::MAX) {
                const int pi = probe->begin(GemmProbe::EPI_ATTENTION, 2, st, 0);
                aev = &probe->ev[2 * pi];
            }
            if (int rc = launch_attention(w.big, w.h, B, L, t->heads, causal, 1, st, nullptr, nullptr, nullptr, aev)) return rc;
            if (int rc = resid_linear(w.h, W, lb + t->lo_out_w, lb + t->lo_out_b)) return rc;
            if (int rc = ln_linear(lb + t->lo_fc_w, lb + t->lo_fc_cb, lb + t->lo_fc_colsum, 4 * W, EPI_LN_BIAS_QGELU_BF16)) return rc;
            if (int rc = resid_linear(w.big, 4 * W, lb + t->lo_proj_w, lb + t->lo_proj_b)) return rc;
        }
        return 0;
    }
    if (fp8) {
        // FP8 blocks. LayerNorm writes its rows as e4m3 with a row scale (it sees whole rows); the two activations that a
        // GEMM consumes straight from a producing kernel - attention's output and the QuickGELU rows - travel as e4m3 with
        // MX block scales written by the producer itself where it has the fused form (attention52x4; the persistent
        // QuickGELU GEMM), else by quantize_rows_fp8mx_kernel over the producer's bf16 rows: the same bytes either way.
        auto gemm8 = [&](const unsigned char* A8, const float* a_scale, const unsigned char* a_bs, int K, uint64_t w_off,
                         uint64_t s_off, uint64_t b_off, void* out, unsigned char* out_bs, int N, int epi) -> int {
            GemmArgs g{};
            g.bias = at<float>(blob, b_off); g.out = out; g.M = M; g.N = N; g.K = K;
            g.A = reinterpret_cast<const unsigned short*>(A8);
            g.W = at<unsigned short>(blob, w_off);
            g.a_scale = a_scale; g.a_bscale = a_bs; g.out_bscale = out_bs;
            g.w_scale = at<float>(blob, s_off);
            return launch_gemm_fp8(g, epi, st);
        };
        static const bool fuse_off = dev_knob("CLIPMI_FP8_FUSE", 1) == 0;   // A/B aid (development build)
        const bool fc_mx = !fuse_off && gemm_fp8_emits_mx(M, 4 * W, W);
        unsigned char* const big8 = reinterpret_cast<unsigned char*>(w.big);
        for (int l = 0; l < t->layers; ++l) {
            const uint64_t lb = t->off_layers + (uint64_t)l * t->layer_stride;
            LnArgs ln{w.x, at<float>(blob, lb + t->lo_ln1_w), at<float>(blob, lb + t->lo_ln1_b), w.h, nullptr, 1, M, W, 1};
            ln.out8 = w.a8; ln.scale8 = w.a_scale;
            if (int rc = launch_layernorm(ln, st)) return rc;
            if (int rc = gemm8(w.a8, w.a_scale, nullptr, W, lb + t->lo_qkv_w, lb + t->lo_qkv_s, lb + t->lo_qkv_b, w.big, nullptr, 3 * W, EPI_BIAS_BF16)) return rc;
            bool fused = false;
            if (int rc = launch_attention(w.big, w.h, B, L, t->heads, causal, 1, st, fuse_off ? nullptr : w.a8, w.a_bs, &fused)) return rc;
            if (!fused)
                if (int rc = launch_quantize_rows_fp8mx(w.h, w.a8, w.a_bs, M, W, st)) return rc;
            if (int rc = gemm8(w.a8, nullptr, w.a_bs, W, lb + t->lo_out_w, lb + t->lo_out_s, lb + t->lo_out_b, w.x, nullptr, W, EPI_BIAS_RESID_F32)) return rc;
            ln.w = at<float>(blob, lb + t->lo_ln2_w); ln.b = at<float>(blob, lb + t->lo_ln2_b);
            if (int rc = launch_layernorm(ln, st)) return rc;
            const unsigned char* h8 = w.a8;
            if (fc_mx) {
                // the QuickGELU rows leave the GEMM as e4m3 + block scales, into the bf16 buffer's bytes
                if (int rc = gemm8(w.a8, w.a_scale, nullptr, W, lb + t->lo_fc_w, lb + t->lo_fc_s, lb + t->lo_fc_b, big8, w.a_bs, 4 * W, EPI_BIAS_QGELU_BF16)) return rc;
                h8 = big8;
            } else {
                if (int rc = gemm8(w.a8, w.a_scale, nullptr, W, lb + t->lo_fc_w, lb + t->lo_fc_s, lb + t->lo_fc_b, w.big, nullptr, 4 * W, EPI_BIAS_QGELU_BF16)) return rc;
                if (int rc = launch_quantize_rows_fp8mx(w.big, w.a8, w.a_bs, M, 4 * W, st)) return rc;
            }
            if (int rc = gemm8(h8, nullptr, w.a_bs, 4 * W, lb + t->lo_proj_w, lb + t->lo_proj_s, lb + t->lo_proj_b, w.x, nullptr, W, EPI_BIAS_RESID_F32)) return rc;
        }
        return 0;
    }
    for (int l = 0; l < t->layers; ++l) {
        const uint64_t lb = t->off_layers + (uint64_t)l * t->layer_stride;
        LnArgs ln{w.x, at<float>(blob, lb + t->lo_ln1_w), at<float>(blob, lb + t->lo_ln1_b), w.h, nullptr, 1, M, W, 1};
        if (int rc = launch_layernorm(ln, st)) return rc;
        if (int rc = linear(w.h, W, lb + t->lo_qkv_w, lb + t->lo_qkv_b, w.big, 3 * W, EPI_BIAS_BF16)) return rc;
        if (int rc = launch_attention(w.big, w.h, B, L, t->heads, causal, 1, st)) return rc;
        if (int rc = linear(w.h, W, lb + t->lo_out_w, lb + t->lo_out_b, w.x, W, EPI_BIAS_RESID_F32)) return rc;
        ln.w = at<float>(blob, lb + t->lo_ln2_w); ln.b = at<float>(blob, lb + t->lo_ln2_b);
        if (int rc = launch_layernorm(ln, st)) return rc;
        if (int rc = linear(w.h, W, lb + t->lo_fc_w, lb + t->lo_fc_b, w.big, 4 * W, EPI_BIAS_QGELU_BF16)) return rc;
        if (int rc = linear(w.big, 4 * W, lb + t->lo_proj_w, lb + t->lo_proj_b, w.x, W, EPI_BIAS_RESID_F32)) return rc;
    }
    return 0;
}

// pooled rows -> final LayerNorm -> projection [E][W] -> f32 [B][E] (-> optional L2 normalise)
int run_head(const clipmi_tower* t, const void* blob, const Ws& w, int B, const int* rowidx, long long row_step,
             float* out, int normalize, hipStream_t st) {
    const bool split = t->ln_fold && t->weight_format == 0;      // bf16 LN-folded towers keep the residual as (hi, lo)
    LnArgs ln{split ? nullptr : w.x, at<float>(blob, t->off_ln_post_w), at<float>(blob, t->off_ln_post_b), w.pooled, rowidx,
              row_step, B, t->width, 1};
    ln.x3 = w.x3;                          // split: the residual stream is rows of (hi | lo)
    if (int rc = launch_layernorm(ln, st)) return rc;
    GemmArgs g{};
    g.A = w.pooled; g.W = at<unsigned short>(blob, t->off_out_proj); g.bias = nullptr; g.out = out;
    g.M = B; g.N = t->embed; g.K = t->width;
    if (int rc = launch_gemm_algo(g, EPI_F32, 0, st)) return rc;
    if (normalize) return clipmi_l2_normalize_rows(out, B, t->embed, st);
    return 0;
}

}  // namespace
}  // namespace clipmi

using namespace clipmi;

extern "C" size_t clipmi_encode_image_workspace_bytes(const clipmi_tower* t, int B) {
    if (check_tower(t, 0, "encode_image")) return 0;
    if (B < 1) { set_err(CLIPMI_EINVAL, "encode_image: B=%d", B); return 0; }
    return carve(t, B, nullptr, ~(size_t)0, nullptr);
}

static int encode_image_impl(const clipmi_tower* t, const void* blob_dev, const void* pixels_dev, int pix_dtype,
                             int B, float* out_dev, int normalize, void* ws_dev, size_t ws_bytes, void* stream, GemmProbe* probe) {
    if (int rc = check_tower(t, 0, "encode_image")) return rc;
    if (!blob_dev || !pixels_dev || !out_dev || !ws_dev) return set_err(CLIPMI_EINVAL, "encode_image: NULL pointer");
    if (B < 1) return set_err(CLIPMI_EINVAL, "encode_image: B=%d", B);
    if (pix_dtype != CLIPMI_F32 && pix_dtype != CLIPMI_BF16 && pix_dtype != CLIPMI_U8)
        return set_err(CLIPMI_EINVAL, "encode_image: pix_dtype %d", pix_dtype);
    const int grid = t->res / t->patch, np = grid * grid;
    if (np + 1 != t->tokens || t->patch_k % 64 != 0 || t->patch_k < 3 * t->patch * t->patch)
        return set_err(CLIPMI_EINVAL, "encode_image: inconsistent tower (res %d patch %d tokens %d patch_k %d)", t->res,
                       t->patch, t->tokens, t->patch_k);
    const size_t need = clipmi_encode_image_workspace_bytes(t, B);
    if (ws_bytes < need) return set_err(CLIPMI_EWORKSPACE, "encode_image: workspace %zu < %zu", ws_bytes, need);
    Ws w;
    carve(t, B, ws_dev, ws_bytes, &w);
    hipStream_t st = as_stream(stream);
    const int W = t->width, L = t->tokens;

    PatchArgs pa{pixels_dev, w.patches, pix_dtype, B, t->res, t->patch, grid, np, t->patch_k};
    if (int rc = launch_patchify(pa, st)) return rc;
    GemmArgs g{};
    g.A = w.patches; g.W = at<unsigned short>(blob_dev, t->off_patch_w); g.bias = nullptr; g.out = w.x;
    g.M = B * np; g.N = W; g.K = t->patch_k; g.pos = at<float>(blob_dev, t->off_pos); g.np = np; g.L = L;
    if (int rc = launch_gemm_algo(g, EPI_PATCH_F32, 0, st)) return rc;
    hipLaunchKernelGGL(cls_rows_kernel, dim3((unsigned)(((long long)B * W + 255) / 256)), dim3(256), 0, st, w.x,
                       at<float>(blob_dev, t->off_cls), at<float>(blob_dev, t->off_pos), B, L, W);
    CLIPMI_CHECK_LAUNCH("cls_rows_kernel");
    LnArgs ln{w.x, at<float>(blob_dev, t->off_ln_pre_w), at<float>(blob_dev, t->off_ln_pre_b), w.x, nullptr, 1, B * L, W, 0};
    const bool fold8 = t->ln_fold && t->weight_format == 1;
    if (t->ln_fold && !fold8) { ln.out_x3 = w.x3; ln.out_part = w.ln_part; }     // straight into the split residual
    if (int rc = launch_layernorm(ln, st)) return rc;       // ln_pre (ln_fold 0 / FP8: in place, each wave owns its row)
#ifdef CLIPMI_DEV
    if (fold8)     // FP8 tower with folded LayerNorms: the embedded rows' e4m3 image + statistics for the first qkv GEMM
        if (int rc = launch_rows_mx_stats(w.x, w.x8, w.x8_bs, w.ln_part, B * L, W, st)) return rc;
#endif
    if (int rc = run_layers(t, blob_dev, w, B, 0, st, probe)) return rc;
    return run_head(t, blob_dev, w, B, nullptr, L, out_dev, normalize, st);
}

extern "C" int clipmi_encode_image(const clipmi_tower* t, const void* blob_dev, const void* pixels_dev, int pix_dtype,
                                   int B, float* out_dev, int normalize, void* ws_dev, size_t ws_bytes, void* stream) {
    return encode_image_impl(t, blob_dev, pixels_dev, pix_dtype, B, out_dev, normalize, ws_dev, ws_bytes, stream, nullptr);
}

extern "C" size_t clipmi_encode_text_workspace_bytes(const clipmi_tower* t, int Q) {
    if (check_tower(t, 1, "encode_text")) return 0;
    if (Q < 1) { set_err(CLIPMI_EINVAL, "encode_text: Q=%d", Q); return 0; }
    return carve(t, Q, nullptr, ~(size_t)0, nullptr);
}

extern "C" int clipmi_encode_text(const clipmi_tower* t, const void* blob_dev, const int32_t* ids_dev, int Q,
                                  float* out_dev, int normalize, void* ws_dev, size_t ws_bytes, void* stream) {
    if (int rc = check_tower(t, 1, "encode_text")) return rc;
    if (!blob_dev || !ids_dev || !out_dev || !ws_dev) return set_err(CLIPMI_EINVAL, "encode_text: NULL pointer");
    if (Q < 1) return set_err(CLIPMI_EINVAL, "encode_text: Q=%d", Q);
    const size_t need = clipmi_encode_text_workspace_bytes(t, Q);
    if (ws_bytes < need) return set_err(CLIPMI_EWORKSPACE, "encode_text: workspace %zu < %zu", ws_bytes, need);
    Ws w;
    carve(t, Q, ws_dev, ws_bytes, &w);
    hipStream_t st = as_stream(stream);
    const int W = t->width, L = t->tokens;
    const long long n4 = (long long)Q * L * (W / 4);
    const bool fold8_ = t->ln_fold && t->weight_format == 1;
    if (t->ln_fold && !fold8_ && W % 256 == 0 && W <= 1024) {
        // LN-folded tower: embedding, split rows + statistics and the EOT rows in one launch (the bits of the three kernels below)
        hipLaunchKernelGGL(text_embed_split_kernel, dim3((unsigned)((Q * L + 3) / 4 + 1)), dim3(256), 0, st, ids_dev,
                           at<float>(blob_dev, t->off_tok_emb), at<float>(blob_dev, t->off_pos), w.x3, w.ln_part, w.rowidx, Q, L, W,
                           t->vocab);
        CLIPMI_CHECK_LAUNCH("text_embed_split_kernel");
        if (int rc = run_layers(t, blob_dev, w, Q, 1, st, nullptr)) return rc;
        return run_head(t, blob_dev, w, Q, w.rowidx, 1, out_dev, normalize, st);
    }
    hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, w.x, ids_dev,
                       at<float>(blob_dev, t->off_tok_emb), at<float>(blob_dev, t->off_pos), Q, L, W, t->vocab);
    CLIPMI_CHECK_LAUNCH("text_embed_kernel");
    hipLaunchKernelGGL(eot_rows_kernel, dim3((Q + 63) / 64), dim3(64), 0, st, ids_dev, w.rowidx, Q, L);
    CLIPMI_CHECK_LAUNCH("eot_rows_kernel");
#ifdef CLIPMI_DEV
    if (t->ln_fold && t->weight_format == 1) {
        if (int rc = launch_rows_mx_stats(w.x, w.x8, w.x8_bs, w.ln_part, Q * L, W, st)) return rc;
    } else
#endif
    if (t->ln_fold) {
        if (int rc = launch_split_stats(w.x, false, w.x3, w.ln_part, Q * L, W, st)) return rc;
    }
    if (int rc = run_layers(t, blob_dev, w, Q, 1, st, nullptr)) return rc;
    return run_head(t, blob_dev, w, Q, w.rowidx, 1, out_dev, normalize, st);
}

// Measurement hook (bench.py roofline): clipmi_encode_image `reps` times per estimator with HIP events around the launches
// of the GEMM whose epilogue is `probe_epi` (1 = MLP c_fc + QuickGELU) on `stream`; synchronises. Three estimators of that
// kernel's in-situ duration, each from its own `reps` forward passes (ms[0..2], averages over the timed launches):
//   ms[0] begin -> end events of hipExtLaunchKernel. The begin event is stamped when the dispatch packet is PROCESSED, which in
//         a back-to-back stream is before the previous kernel has drained: it reads long (14 % for the 200-us c_fc launch).
//   ms[1] a plain event record in front of the launch (completes with everything before it) -> the launch's end event:
//         kernel + the boundary to the previous kernel + the event packet.
//   ms[2] end event of the previous GEMM launch -> end event of this one, used where that previous launch is the kernel
//         directly in front in the stream (LN-folded tower: out_proj -> c_fc): completion to completion = the kernel + ONE
//         kernel boundary (~1.5 us), the closest to the profiler's dispatch duration; 0 when not applicable.
// kernel_kind / epi_ran: which kernel and epilogue the timed launches really were (0 gemm_bf16_nt_kernel, 1 gemm256_..., 2
// gemm256p_...; the EPI_* number).
static int encode_probe(const clipmi_tower* t, const void* blob_dev, const void* pixels_dev, int pix_dtype, int B, float* out_dev,
                        void* ws_dev, size_t ws_bytes, void* stream, int probe_epi, int reps, int nmodes, float* ms, int* launches,
                        int* kernel_kind, int* epi_ran) {
    if (!ms || reps < 1) return set_err(CLIPMI_EINVAL, "dbg_encode_image_probe_ms: bad arguments");
    // probe_epi bits 8..: only launches with this K (out_proj and c_proj share the residual epilogue and differ in K); 0 = any
    const int k_filter = probe_epi >> 8;
    probe_epi &= 0xff;
    GemmProbe p;             // lives on this call's stack: the library keeps no mutable state (clipmi.h)
    for (int i = 0; i < 2 * GemmProbe::MAX; ++i)
        if (hipEventCreate(&p.ev[i]) != hipSuccess) return set_err(CLIPMI_EHIP, "hipEventCreate");
    for (int i = 0; i < GemmProbe::MAX; ++i)
        if (hipEventCreate(&p.pre[i]) != hipSuccess) return set_err(CLIPMI_EHIP, "hipEventCreate");
    int rc = 0;
    for (int mode = 0; mode < nmodes && rc == 0; ++mode) {
        double total = 0.0;
        int count = 0;
        for (int r = 0; r < reps && rc == 0; ++r) {
            p.epi = probe_epi; p.n = 0; p.mode = mode;
            rc = encode_image_impl(t, blob_dev, pixels_dev, pix_dtype, B, out_dev, 1, ws_dev, ws_bytes, stream, &p);
            if (rc) break;
            if (hipStreamSynchronize(as_stream(stream)) != hipSuccess) { rc = set_err(CLIPMI_EHIP, "hipStreamSynchronize"); break; }
            for (int i = 0; i < p.n; ++i) {
                if (GemmProbe::base_of(p.epi_of[i]) != probe_epi || (k_filter && p.k_of[i] != k_filter)) continue;
                float v = 0.f;
                if (mode == 0) (void)hipEventElapsedTime(&v, p.ev[2 * i], p.ev[2 * i + 1]);
                else if (mode == 1) (void)hipEventElapsedTime(&v, p.pre[i], p.ev[2 * i + 1]);
                else {
                    // the launch in front must carry its own end stamp with NO kernel between them: the residual producer in front
                    // of qkv / c_fc, c_fc in front of c_proj, attention (probed in this mode) in front of out_proj
                    if (i == 0 || p.kernel_of[i - 1] != 2 ||
                        (p.epi_of[i - 1] != EPI_BIAS_RESID_LN_F32 && p.epi_of[i - 1] != EPI_LN_BIAS_QGELU_BF16 &&
                         p.epi_of[i - 1] != GemmProbe::EPI_ATTENTION)) continue;
                    (void)hipEventElapsedTime(&v, p.ev[2 * (i - 1) + 1], p.ev[2 * i + 1]);
                }
                total += v;
                ++count;
                if (kernel_kind) *kernel_kind = p.kernel_of[i];
                if (epi_ran) *epi_ran = p.epi_of[i];
            }
        }
        if (rc == 0 && count == 0 && mode == 0) rc = set_err(CLIPMI_EINVAL, "dbg_encode_image_probe_ms: no launch matched epi %d", probe_epi);
        if (rc == 0) { ms[mode] = count ? (float)(total / count) : 0.f; if (launches && mode == 0) *launches = count; }
    }
    for (int i = 0; i < 2 * GemmProbe::MAX; ++i) (void)hipEventDestroy(p.ev[i]);
    for (int i = 0; i < GemmProbe::MAX; ++i) (void)hipEventDestroy(p.pre[i]);
    return rc;
}

extern "C" int clipmi_dbg_encode_image_probe_ms(const clipmi_tower* t, const void* blob_dev, const void* pixels_dev,
                                                int pix_dtype, int B, float* out_dev, void* ws_dev, size_t ws_bytes,
                                                void* stream, int probe_epi, int reps, float* kernel_ms, int* launches) {
    return encode_probe(t, blob_dev, pixels_dev, pix_dtype, B, out_dev, ws_dev, ws_bytes, stream, probe_epi, reps, 1, kernel_ms,
                        launches, nullptr, nullptr);
}

extern "C" int clipmi_dbg_encode_image_probe3_ms(const clipmi_tower* t, const void* blob_dev, const void* pixels_dev,
                                                 int pix_dtype, int B, float* out_dev, void* ws_dev, size_t ws_bytes,
                                                 void* stream, int probe_epi, int reps, float* ms3, int* launches,
                                                 int* kernel_kind, int* epi_ran) {
    return encode_probe(t, blob_dev, pixels_dev, pix_dtype, B, out_dev, ws_dev, ws_bytes, stream, probe_epi, reps, 3, ms3,
                        launches, kernel_kind, epi_ran);
}
