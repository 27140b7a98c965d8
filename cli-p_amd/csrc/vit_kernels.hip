// vit_kernels.hip — launchers for the non-GEMM tower kernels and their debug entry points.
#include "vit_kernels.hpp"
#include <hip/hip_ext.h>
#include <cstdlib>

namespace clipmi {

int launch_layernorm(const LnArgs& a, hipStream_t st) {
    if (a.M < 1) return 0;
    if (a.W % 4 != 0 || a.W > 1024) return set_err(CLIPMI_EINVAL, "layernorm: W=%d (need W %% 4 == 0, W <= 1024)", a.W);
    if (a.out_x3 && (a.W % 256 != 0 || !a.out_part)) return set_err(CLIPMI_EINVAL, "layernorm: split output needs W %% 256 == 0");
    hipLaunchKernelGGL(layernorm_kernel, dim3((a.M + 3) / 4), dim3(256), 0, st, a);
    CLIPMI_CHECK_LAUNCH("layernorm_kernel");
    return 0;
}

int launch_split_stats(const float* x_or_add, bool add, void* x3, float* part, int M, int W, hipStream_t st) {
    if (M < 1) return 0;
    if (!x_or_add || !x3 || !part || W % 256 != 0 || W < 256 || W > 1024)
        return set_err(CLIPMI_EINVAL, "split_stats: W=%d (need W %% 256 == 0, 256 <= W <= 1024)", W);
    if (add) hipLaunchKernelGGL(split_stats_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, st, x_or_add, x3, part, M, W);
    else hipLaunchKernelGGL(split_stats_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, st, x_or_add, x3, part, M, W);
    CLIPMI_CHECK_LAUNCH("split_stats_kernel");
    return 0;
}

template <int NT, bool CAUSAL, bool TR>
static int launch_attn_t(const unsigned short* qkv, unsigned short* out, int B, int L, int heads, hipStream_t st) {
    constexpr int KS = (NT + 1) / 2;
    const size_t lds = 4 * KS * 32 * 128;
    const long long items = (long long)B * heads;
    hipLaunchKernelGGL((attention_kernel<NT, CAUSAL, TR>), dim3((unsigned)((items + 3) / 4)), dim3(256), lds, st, qkv, out,
                       B, L, heads);
    CLIPMI_CHECK_LAUNCH("attention_kernel");
    return 0;
}

int launch_attention(const unsigned short* qkv, unsigned short* out, int B, int L, int heads, int causal, int tr,
                     hipStream_t st, unsigned char* out8, unsigned char* out_bs, bool* fused, hipEvent_t* probe_ev) {
    if (fused) *fused = false;
    if (B < 1) return 0;
    if (L < 1) return set_err(CLIPMI_EINVAL, "attention: L=%d", L);
    if (L > 80) {
        if (causal) return set_err(CLIPMI_EUNSUPPORTED, "attention: causal mask with L=%d > 80", L);
        // four waves of 32 queries per workgroup (128 queries share a staged K/V block; 128 VGPRs = four waves per SIMD,
        // five workgroups = 20 waves per CU by LDS). Round 1's 64-query waves (two per SIMD): 804 us per ViT-L/14@336
        // layer at B = 266 against 741 for this shape before its VALU diet, removed.
        const int qblocks = (L + 31) / 32;
        const int g4 = (qblocks + 3) / 4;
        const long long groups = (long long)B * heads * g4;
        hipLaunchKernelGGL((attention_flash_kernel<4, 2>), dim3((unsigned)groups), dim3(256), 2 * 16384, st, qkv, out, B, L, heads, g4);
        CLIPMI_CHECK_LAUNCH("attention_flash_kernel");
        return 0;
    }
    const int nt = (L + 15) / 16;
    // development switch for A/B runs on one box: CLIPMI_ATTN52=0 keeps ViT-B/32 on attention_kernel<4>
    // (CLIPMI_ATTN52: 0 = attention_kernel<4>, 1 = attention52_kernel, default 2 = attention52x4_kernel)
    static const int use52 = (int)dev_knob("CLIPMI_ATTN52", 2);
    if (L >= 49 && L <= 52 && !causal && tr == 1 && use52 == 2) {
        // one workgroup per (image, head), one query tile per wave
        const long long items = (long long)B * heads;
        const long long resident = (long long)NUM_CU * 5;               // 5 workgroups (20 waves, <= 96 VGPRs) per CU
        if (out8 && out_bs && fused) {
            hipLaunchKernelGGL(attention52x4_kernel<true>, dim3((unsigned)(items < resident ? items : resident)), dim3(256),
                               64 * 128 + 52 * 128, st, qkv, out, B, L, heads, out8, out_bs);
            *fused = true;
        } else if (probe_ev) {
            hipExtLaunchKernelGGL(attention52x4_kernel<false>, dim3((unsigned)(items < resident ? items : resident)), dim3(256),
                                  64 * 128 + 52 * 128, st, probe_ev[0], probe_ev[1], 0, qkv, out, B, L, heads, (unsigned char*)nullptr,
                                  (unsigned char*)nullptr);
        } else {
            hipLaunchKernelGGL(attention52x4_kernel<false>, dim3((unsigned)(items < resident ? items : resident)), dim3(256),
                               64 * 128 + 52 * 128, st, qkv, out, B, L, heads, (unsigned char*)nullptr, (unsigned char*)nullptr);
        }
        CLIPMI_CHECK_LAUNCH("attention52x4_kernel");
        return 0;
    }
    if (L >= 49 && L <= 52 && !causal && tr && use52) {
        // ViT-B/32: the low-register form (24 waves per CU); tr = 0 keeps the plain-read reference kernel for tests
        const long long items = (long long)B * heads;
        const size_t lds = 4 * 52 * 128;
        hipLaunchKernelGGL(attention52_kernel, dim3((unsigned)((items + 3) / 4)), dim3(256), lds, st, qkv, out, B, L, heads);
        CLIPMI_CHECK_LAUNCH("attention52_kernel");
        return 0;
    }
#define ATT(NT_)                                                                                         \
    if (nt <= NT_) {                                                                                     \
        if (causal) return tr ? launch_attn_t<NT_, true, true>(qkv, out, B, L, heads, st)                \
                              : launch_attn_t<NT_, true, false>(qkv, out, B, L, heads, st);              \
        return tr ? launch_attn_t<NT_, false, true>(qkv, out, B, L, heads, st)                           \
                  : launch_attn_t<NT_, false, false>(qkv, out, B, L, heads, st);                         \
    }
    ATT(1) ATT(2) ATT(4) ATT(5)
#undef ATT
    return set_err(CLIPMI_EUNSUPPORTED, "attention: L=%d", L);
}

int launch_quantize_rows_fp8(const unsigned short* in, unsigned char* out, float* scale, int M, int K, hipStream_t st) {
    if (M < 1) return 0;
    if (K < 8 || K % 8 != 0 || !in || !out || !scale) return set_err(CLIPMI_EINVAL, "quantize_rows_fp8: K=%d", K);
    hipLaunchKernelGGL(quantize_rows_fp8_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, in, out, scale, M, K);
    CLIPMI_CHECK_LAUNCH("quantize_rows_fp8_kernel");
    return 0;
}

int launch_quantize_rows_fp8mx(const unsigned short* in, unsigned char* out, unsigned char* bscale, int M, int K, hipStream_t st) {
    if (M < 1) return 0;
    if (K < 32 || K % 32 != 0 || !in || !out || !bscale) return set_err(CLIPMI_EINVAL, "quantize_rows_fp8mx: K=%d", K);
    hipLaunchKernelGGL(quantize_rows_fp8mx_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, in, out, bscale, M, K);
    CLIPMI_CHECK_LAUNCH("quantize_rows_fp8mx_kernel");
    return 0;
}

#ifdef CLIPMI_DEV
int launch_rows_mx_stats(const float* x, unsigned char* x8, unsigned char* bs, float* part, int M, int W, hipStream_t st) {
    if (M < 1) return 0;
    if (W < 256 || W % 256 != 0 || !x || !x8 || !bs || !part) return set_err(CLIPMI_EINVAL, "rows_mx_stats: W=%d", W);
    hipLaunchKernelGGL(rows_mx_stats_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, st, x, x8, bs, part, M, W);
    CLIPMI_CHECK_LAUNCH("rows_mx_stats_kernel");
    return 0;
}
#endif

int launch_patchify(const PatchArgs& a, hipStream_t st) {
    if (a.dtype == CLIPMI_U8 && a.P % 16 == 0 && a.R % 16 == 0 && a.patch_k == 3 * a.P * a.P && a.P * a.R <= 48 * 1024) {
        hipLaunchKernelGGL(patchify_strip_u8_kernel, dim3((unsigned)((long long)a.B * 3 * a.grid)), dim3(256), (size_t)a.P * a.R + 512, st, a);
        CLIPMI_CHECK_LAUNCH("patchify_strip_u8_kernel");
        return 0;
    }
    const long long total = (long long)a.B * a.np * (a.patch_k / 8);
    hipLaunchKernelGGL(patchify_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a);
    CLIPMI_CHECK_LAUNCH("patchify_kernel");
    return 0;
}

}  // namespace clipmi

using namespace clipmi;

extern "C" int clipmi_dbg_layernorm(const float* x_dev, const float* w_dev, const float* b_dev, void* out_dev, int M,
                                    int W, int out_bf16, void* stream) {
    if (!x_dev || !w_dev || !b_dev || !out_dev) return set_err(CLIPMI_EINVAL, "dbg_layernorm: NULL pointer");
    LnArgs a{x_dev, w_dev, b_dev, out_dev, nullptr, 1, M, W, out_bf16};
    return launch_layernorm(a, as_stream(stream));
}

// `causal` bit 0 = causal mask; bit 1 = use the 2-byte LDS reads instead of ds_read_b64_tr_b16; bit 2 = the one-wave-
// per-(image, head) form attention52_kernel where the default is attention52x4_kernel (49 <= L <= 52)
extern "C" int clipmi_dbg_attention(const void* qkv_dev, void* out_dev, int B, int L, int heads, int causal,
                                    void* stream) {
    if (!qkv_dev || !out_dev) return set_err(CLIPMI_EINVAL, "dbg_attention: NULL pointer");
    return launch_attention(static_cast<const unsigned short*>(qkv_dev), static_cast<unsigned short*>(out_dev), B, L,
                            heads, causal & 1, (causal & 2) ? 0 : ((causal & 4) ? 2 : 1), as_stream(stream));
}

extern "C" int clipmi_dbg_quantize_rows_fp8(const void* in_bf16_dev, void* out_fp8_dev, float* scale_dev, int M, int K,
                                            void* stream) {
    return launch_quantize_rows_fp8(static_cast<const unsigned short*>(in_bf16_dev), static_cast<unsigned char*>(out_fp8_dev),
                                    scale_dev, M, K, as_stream(stream));
}

extern "C" int clipmi_dbg_quantize_rows_fp8mx(const void* in_bf16_dev, void* out_fp8_dev, void* bscale_dev, int M, int K, void* stream) {
    return launch_quantize_rows_fp8mx(static_cast<const unsigned short*>(in_bf16_dev), static_cast<unsigned char*>(out_fp8_dev),
                                      static_cast<unsigned char*>(bscale_dev), M, K, as_stream(stream));
}

extern "C" int clipmi_dbg_split_stats(const float* x_dev, int add, void* x3_dev, float* part_dev, int M, int W, void* stream) {
    return launch_split_stats(x_dev, add != 0, x3_dev, part_dev, M, W, as_stream(stream));
}
