// stubs.hip — entry points not implemented yet return CLIPMI_EUNSUPPORTED (never a fallback).
#include "common.hpp"
using namespace clipmi;
extern "C" size_t clipmi_encode_image_workspace_bytes(const clipmi_tower*, int) { set_err(CLIPMI_EUNSUPPORTED, "encode_image: not built yet"); return 0; }
extern "C" int clipmi_encode_image(const clipmi_tower*, const void*, const void*, int, int, float*, int, void*, size_t, void*) { return set_err(CLIPMI_EUNSUPPORTED, "encode_image: not built yet"); }
extern "C" size_t clipmi_encode_text_workspace_bytes(const clipmi_tower*, int) { set_err(CLIPMI_EUNSUPPORTED, "encode_text: not built yet"); return 0; }
extern "C" int clipmi_encode_text(const clipmi_tower*, const void*, const int32_t*, int, float*, int, void*, size_t, void*) { return set_err(CLIPMI_EUNSUPPORTED, "encode_text: not built yet"); }
extern "C" int clipmi_dbg_gemm_bf16(const void*, const void*, const float*, void*, int, int, int, int, void*) { return set_err(CLIPMI_EUNSUPPORTED, "gemm: not built yet"); }
extern "C" int clipmi_dbg_layernorm(const float*, const float*, const float*, void*, int, int, int, void*) { return set_err(CLIPMI_EUNSUPPORTED, "layernorm: not built yet"); }
extern "C" int clipmi_dbg_attention(const void*, void*, int, int, int, int, void*) { return set_err(CLIPMI_EUNSUPPORTED, "attention: not built yet"); }
