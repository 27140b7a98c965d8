"""Weight handling for the CLIP towers: architecture table, seeded random state-dicts with OpenAI
CLIP key names, local checkpoint loading, and the packer that turns a state-dict into ONE device
blob + the POD `clipmi_tower` descriptor the C ABI consumes (include/clipmi.h).

Stands in for `clip.load("ViT-B/32", device=device, jit=False)` (reference build-index.py:18,
query-index.py:21). The upstream loader downloads weights; there is no network here, so weights
come from a LOCAL file with OpenAI key names (SURVEY.md §8b "weight-file contract") or from
`random_state_dict` (synthetic benchmarks and parity tests).
"""
import math
import os

import torch

from . import _lib

ARCHS = {
    # name: vision(width, layers, patch, res), text(width, layers, heads, ctx, vocab), embed
    "ViT-B/32": dict(v_width=768, v_layers=12, patch=32, res=224,
                     t_width=512, t_layers=12, ctx=77, vocab=49408, embed=512),
    "ViT-B/16": dict(v_width=768, v_layers=12, patch=16, res=224,
                     t_width=512, t_layers=12, ctx=77, vocab=49408, embed=512),
    "ViT-L/14": dict(v_width=1024, v_layers=24, patch=14, res=224,
                     t_width=768, t_layers=12, ctx=77, vocab=49408, embed=768),
    "ViT-L/14@336px": dict(v_width=1024, v_layers=24, patch=14, res=336,
                           t_width=768, t_layers=12, ctx=77, vocab=49408, embed=768),
    # toy with ViT-L/14@336's awkward geometry: 14-pixel patches (K = 588, zero-padded to 640) and
    # 101 tokens (> 80: flash-style attention path)
    "toy-l14": dict(v_width=128, v_layers=2, patch=14, res=140,
                    t_width=128, t_layers=2, ctx=16, vocab=512, embed=128),
    # width 256 (the smallest the FP8 path accepts), 101 tokens (flash attention), 3 layers
    "toy-256": dict(v_width=256, v_layers=3, patch=14, res=140,
                    t_width=128, t_layers=2, ctx=16, vocab=512, embed=128),
    # 2-layer toy used by kernel-level tests (SURVEY.md §8c fixture (i))
    "toy": dict(v_width=128, v_layers=2, patch=32, res=64,
                t_width=128, t_layers=2, ctx=16, vocab=512, embed=128),
}


def _resblocks(sd, prefix, width, layers, g, gain, attn_std, proj_std, fc_std):
    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std
    for i in range(layers):
        p = f"{prefix}.resblocks.{i}"
        sd[f"{p}.ln_1.weight"] = 1.0 + 0.1 * rn(width)
        sd[f"{p}.ln_1.bias"] = 0.1 * rn(width)
        sd[f"{p}.attn.in_proj_weight"] = rn(3 * width, width, std=attn_std * gain)
        sd[f"{p}.attn.in_proj_bias"] = 0.02 * rn(3 * width)
        sd[f"{p}.attn.out_proj.weight"] = rn(width, width, std=proj_std * gain)
        sd[f"{p}.attn.out_proj.bias"] = 0.02 * rn(width)
        sd[f"{p}.ln_2.weight"] = 1.0 + 0.1 * rn(width)
        sd[f"{p}.ln_2.bias"] = 0.1 * rn(width)
        sd[f"{p}.mlp.c_fc.weight"] = rn(4 * width, width, std=fc_std * gain)
        sd[f"{p}.mlp.c_fc.bias"] = 0.02 * rn(4 * width)
        sd[f"{p}.mlp.c_proj.weight"] = rn(width, 4 * width, std=proj_std * gain)
        sd[f"{p}.mlp.c_proj.bias"] = 0.02 * rn(width)


def random_state_dict(arch="ViT-B/32", seed=0, gain=1.5):
    """Seeded random weights with OpenAI CLIP state-dict names and shapes. Standard deviations
    follow the published CLIP initialisation (width^-0.5 families) times `gain` on the matrices,
    plus non-trivial LayerNorm parameters and biases so that no term of the forward is vacuous.
    torch's CPU generator is deterministic for a fixed torch version (the image pins 2.10)."""
    a = ARCHS[arch]
    g = torch.Generator(device="cpu")
    g.manual_seed(int(seed))

    def rn(*shape, std=1.0):
        return torch.randn(*shape, generator=g, dtype=torch.float32) * std

    sd = {}
    W, P, R = a["v_width"], a["patch"], a["res"]
    L = (R // P) ** 2 + 1
    scale = W ** -0.5
    sd["visual.conv1.weight"] = rn(W, 3, P, P, std=(3 * P * P) ** -0.5 * gain)
    sd["visual.class_embedding"] = rn(W, std=scale)
    sd["visual.positional_embedding"] = rn(L, W, std=scale)
    sd["visual.ln_pre.weight"] = 1.0 + 0.1 * rn(W)
    sd["visual.ln_pre.bias"] = 0.1 * rn(W)
    _resblocks(sd, "visual.transformer", W, a["v_layers"], g, gain,
               attn_std=W ** -0.5, proj_std=(W ** -0.5) * ((2 * a["v_layers"]) ** -0.5), fc_std=(2 * W) ** -0.5)
    sd["visual.ln_post.weight"] = 1.0 + 0.1 * rn(W)
    sd["visual.ln_post.bias"] = 0.1 * rn(W)
    sd["visual.proj"] = rn(W, a["embed"], std=scale * gain)

    T = a["t_width"]
    sd["token_embedding.weight"] = rn(a["vocab"], T, std=0.02 * 10)
    sd["positional_embedding"] = rn(a["ctx"], T, std=0.01 * 10)
    _resblocks(sd, "transformer", T, a["t_layers"], g, gain,
               attn_std=T ** -0.5, proj_std=(T ** -0.5) * ((2 * a["t_layers"]) ** -0.5), fc_std=(2 * T) ** -0.5)
    sd["ln_final.weight"] = 1.0 + 0.1 * rn(T)
    sd["ln_final.bias"] = 0.1 * rn(T)
    sd["text_projection"] = rn(T, a["embed"], std=T ** -0.5 * gain)
    sd["logit_scale"] = torch.tensor(math.log(1 / 0.07))
    return sd


def load_state_dict(path):
    """Read a LOCAL checkpoint into an OpenAI-named f32 state-dict: a TorchScript archive such as
    the upstream `ViT-B-32.pt` (torch.jit.load(...).state_dict()), a pickled state-dict, or a
    safetensors file."""
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(path)
    else:
        try:
            sd = torch.jit.load(path, map_location="cpu").state_dict()
        except RuntimeError:
            # a pickled state-dict: tensors and plain containers only (weights_only refuses to unpickle
            # arbitrary objects from a user-supplied path)
            sd = torch.load(path, map_location="cpu", weights_only=True)
            if "state_dict" in sd:
                sd = sd["state_dict"]
    sd = {k: v.float() for k, v in sd.items() if torch.is_tensor(v)}
    if "visual.conv1.weight" not in sd and "vision_model.embeddings.patch_embedding.weight" in sd:
        sd = from_hf_state_dict(sd)          # a Hugging Face CLIPModel checkpoint
    if "visual.conv1.weight" not in sd:
        raise ValueError(f"{path}: no 'visual.conv1.weight' — expected OpenAI CLIP ViT key names")
    return sd


def from_hf_state_dict(hf):
    """Map a Hugging Face `CLIPModel` state-dict (keys `vision_model.*`, `text_model.*`,
    `visual_projection.weight`, `text_projection.weight`) onto the OpenAI key names the packer expects:
    q/k/v projections are concatenated in that order into `in_proj_*`, `visual_projection.weight`
    ([E, W]) is `visual.proj` transposed, `pre_layrnorm` is `ln_pre`."""
    hf = {k: v.float() for k, v in hf.items() if torch.is_tensor(v)}
    sd = {}

    def layers(src, dst):
        i = 0
        while f"{src}.encoder.layers.{i}.layer_norm1.weight" in hf:
            s_, d_ = f"{src}.encoder.layers.{i}", f"{dst}.resblocks.{i}"
            sd[f"{d_}.attn.in_proj_weight"] = torch.cat([hf[f"{s_}.self_attn.{x}_proj.weight"] for x in "qkv"], 0)
            sd[f"{d_}.attn.in_proj_bias"] = torch.cat([hf[f"{s_}.self_attn.{x}_proj.bias"] for x in "qkv"], 0)
            sd[f"{d_}.attn.out_proj.weight"] = hf[f"{s_}.self_attn.out_proj.weight"]
            sd[f"{d_}.attn.out_proj.bias"] = hf[f"{s_}.self_attn.out_proj.bias"]
            for a, b in (("layer_norm1", "ln_1"), ("layer_norm2", "ln_2")):
                sd[f"{d_}.{b}.weight"], sd[f"{d_}.{b}.bias"] = hf[f"{s_}.{a}.weight"], hf[f"{s_}.{a}.bias"]
            for a, b in (("fc1", "c_fc"), ("fc2", "c_proj")):
                sd[f"{d_}.mlp.{b}.weight"], sd[f"{d_}.mlp.{b}.bias"] = hf[f"{s_}.mlp.{a}.weight"], hf[f"{s_}.mlp.{a}.bias"]
            i += 1
        return i

    sd["visual.conv1.weight"] = hf["vision_model.embeddings.patch_embedding.weight"]
    sd["visual.class_embedding"] = hf["vision_model.embeddings.class_embedding"]
    sd["visual.positional_embedding"] = hf["vision_model.embeddings.position_embedding.weight"]
    sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"] = hf["vision_model.pre_layrnorm.weight"], hf["vision_model.pre_layrnorm.bias"]
    sd["visual.ln_post.weight"], sd["visual.ln_post.bias"] = hf["vision_model.post_layernorm.weight"], hf["vision_model.post_layernorm.bias"]
    sd["visual.proj"] = hf["visual_projection.weight"].t().contiguous()
    layers("vision_model", "visual.transformer")
    sd["token_embedding.weight"] = hf["text_model.embeddings.token_embedding.weight"]
    sd["positional_embedding"] = hf["text_model.embeddings.position_embedding.weight"]
    sd["ln_final.weight"], sd["ln_final.bias"] = hf["text_model.final_layer_norm.weight"], hf["text_model.final_layer_norm.bias"]
    sd["text_projection"] = hf["text_projection.weight"].t().contiguous()
    layers("text_model", "transformer")
    if "logit_scale" in hf:
        sd["logit_scale"] = hf["logit_scale"]
    return sd


def infer_dims(sd):
    """Model dimensions from tensor shapes, the way the upstream build_model does."""
    W = sd["visual.conv1.weight"].shape[0]
    P = sd["visual.conv1.weight"].shape[-1]
    Lv = sd["visual.positional_embedding"].shape[0]
    grid = round((Lv - 1) ** 0.5)
    v_layers = len({k.split(".")[3] for k in sd if k.startswith("visual.transformer.resblocks.")})
    T = sd["ln_final.weight"].shape[0]
    t_layers = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.")})
    return dict(v_width=W, v_layers=v_layers, patch=P, res=P * grid, v_tokens=Lv,
                t_width=T, t_layers=t_layers, ctx=sd["positional_embedding"].shape[0],
                vocab=sd["token_embedding.weight"].shape[0], embed=sd["text_projection"].shape[1])


class _Blob:
    """256-byte aligned bump layout of tensors into one byte buffer."""

    def __init__(self):
        self.items = []
        self.size = 0

    def put(self, t, dtype):
        self.size = (self.size + 255) // 256 * 256
        off = self.size
        t = t.detach().to("cpu").contiguous().to(dtype).contiguous()
        self.items.append((off, t))
        self.size += t.numel() * t.element_size()
        return off

    def materialise(self, device):
        total = (self.size + 255) // 256 * 256
        buf = torch.zeros(total, dtype=torch.uint8)
        for off, t in self.items:
            raw = t.view(torch.uint8).reshape(-1) if t.dtype != torch.uint8 else t.reshape(-1)
            buf[off:off + raw.numel()] = raw
        return buf.to(device), total


FP8_MAX = 448.0          # largest finite OCP e4m3 value


def quantize_fp8_rows(w):
    """f32 [N][K] -> (uint8 [N][K] OCP e4m3 bit patterns, f32 [N] scales): scale_n = max|w_n| / 448 (1 for a zero
    row), value = RNE(w * (1 / scale)). The same rule the device applies to activation rows
    (quantize_rows_fp8_kernel) and the oracle emulates."""
    w = w.detach().to("cpu", torch.float32).contiguous()
    amax = w.abs().amax(dim=1)
    scale = torch.where(amax > 0, amax / FP8_MAX, torch.ones_like(amax))
    q = (w * (1.0 / scale)[:, None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), scale


def ln_fold_terms(w, bias, gamma, beta):
    """LN-folded linear layer (include/clipmi.h, tower ABI 3): for y = LayerNorm(x; gamma, beta) W^T + bias ->
    (Wg bf16 [N][K], colsum f32 [N], cb f32 [N]) with Wg = bf16(W * gamma) (W = the bf16-rounded weights the
    un-folded path multiplies by), colsum[n] = sum_k Wg[n][k] (of the ROUNDED Wg: it must cancel what the GEMM adds up)
    and cb[n] = sum_k beta[k] W[n][k] + bias[n]; sums in float64."""
    wb = w.detach().to("cpu").to(torch.bfloat16)
    wg = (wb.float() * gamma.detach().float()[None, :]).to(torch.bfloat16)
    colsum = wg.double().sum(dim=1)
    cb = wb.double() @ beta.detach().double() + bias.detach().double()
    return wg, colsum.float(), cb.float()


def ln_fold_terms_fp8(w, bias, gamma, beta):
    """The LN-folded linear layer on FP8 operands (tower ABI 4): -> (e4m3 bytes uint8 [N][K], w_scale f32 [N], colsum f32 [N],
    cb f32 [N]). The folded matrix Wg = W * gamma (W = the bf16-rounded weights, product in f32) is quantised per output
    channel like every FP8 weight; colsum[n] = w_scale[n] * sum_k q[n][k] is the row sum of the DEQUANTISED e4m3 values - it
    must cancel exactly what the GEMM adds up - and cb = W beta + bias as in the bf16 fold; sums in float64."""
    wb = w.detach().to("cpu").to(torch.bfloat16).float()
    wg = wb * gamma.detach().float()[None, :]
    q, scale = quantize_fp8_rows(wg)
    colsum = q.view(torch.float8_e4m3fn).double().sum(dim=1) * scale.double()
    cb = wb.double() @ beta.detach().double() + bias.detach().double()
    return q, scale, colsum.float(), cb.float()


def _pack_layers(tw, blob, sd, prefix, width, layers, fp8=False):
    bf, f32 = torch.bfloat16, torch.float32
    first = None
    stride = None
    # (CLIPMI_LN_FOLD=0: development switch for A/B runs on one box — the stand-alone LayerNorm passes)
    # (FP8 towers: the folded form - e4m3(W diag(gamma)) weights, colsum of the rounded values - exists, is parity-green and
    #  measured slower than the LayerNorm-pass tower, DESIGN.md 4.4c: development library only, CLIPMI_FP8_LN_FOLD=1)
    fold_ok = not fp8 or os.environ.get("CLIPMI_FP8_LN_FOLD", "0") == "1"
    tw.ln_fold = 1 if (fold_ok and width % 256 == 0 and os.environ.get("CLIPMI_LN_FOLD", "1") != "0") else 0
    names = [("lo_ln1_w", "ln_1.weight", f32), ("lo_ln1_b", "ln_1.bias", f32),
             ("lo_qkv_w", "attn.in_proj_weight", bf), ("lo_qkv_b", "attn.in_proj_bias", f32),
             ("lo_out_w", "attn.out_proj.weight", bf), ("lo_out_b", "attn.out_proj.bias", f32),
             ("lo_ln2_w", "ln_2.weight", f32), ("lo_ln2_b", "ln_2.bias", f32),
             ("lo_fc_w", "mlp.c_fc.weight", bf), ("lo_fc_b", "mlp.c_fc.bias", f32),
             ("lo_proj_w", "mlp.c_proj.weight", bf), ("lo_proj_b", "mlp.c_proj.bias", f32)]
    scale_field = {"lo_qkv_w": "lo_qkv_s", "lo_out_w": "lo_out_s", "lo_fc_w": "lo_fc_s", "lo_proj_w": "lo_proj_s"}
    for i in range(layers):
        base = None
        folded = {}
        if tw.ln_fold:
            p_ = f"{prefix}.resblocks.{i}"
            for fld_w, wk, bk, ln in (("lo_qkv_w", "attn.in_proj_weight", "attn.in_proj_bias", "ln_1"),
                                      ("lo_fc_w", "mlp.c_fc.weight", "mlp.c_fc.bias", "ln_2")):
                args = (sd[f"{p_}.{wk}"], sd[f"{p_}.{bk}"], sd[f"{p_}.{ln}.weight"], sd[f"{p_}.{ln}.bias"])
                if fp8:
                    q8, sc8, colsum8, cb8 = ln_fold_terms_fp8(*args)
                    folded[fld_w] = ((q8, sc8), colsum8, cb8)
                else:
                    folded[fld_w] = ln_fold_terms(*args)
        for field, key, dt in names:
            t = sd[f"{prefix}.resblocks.{i}.{key}"]
            if field in folded:
                t = folded[field][0]                   # W * diag(ln weight), already bf16
            if fp8 and field in scale_field:
                # the weights as the bf16 path stores them, then e4m3 + one scale per output channel (the LN-folded
                # matrices come quantised from ln_fold_terms_fp8: their colsum belongs to exactly these bytes)
                q, sc = t if field in folded else quantize_fp8_rows(t.to(torch.bfloat16).float())
                off = blob.put(q, torch.uint8)
                soff = blob.put(sc, f32)
                if i == 0:
                    setattr(tw, scale_field[field], soff - (base if base is not None else off))
                else:
                    assert soff - base == getattr(tw, scale_field[field])
            else:
                off = blob.put(t, dt)
            if base is None:
                base = off
            if i == 0:
                setattr(tw, field, off - base)
            else:
                assert off - base == getattr(tw, field), "layers must have identical layouts"
        if tw.ln_fold:
            for fld_s, fld_c, fld_w in (("lo_qkv_colsum", "lo_qkv_cb", "lo_qkv_w"), ("lo_fc_colsum", "lo_fc_cb", "lo_fc_w")):
                _, colsum, cb = folded[fld_w]
                for fld, t in ((fld_s, colsum), (fld_c, cb)):
                    off = blob.put(t, f32)
                    if i == 0:
                        setattr(tw, fld, off - base)
                    else:
                        assert off - base == getattr(tw, fld)
        if i == 0:
            first = base
        elif i == 1:
            stride = base - first
        if i >= 1:
            assert base - first == stride * i
    tw.off_layers = first
    tw.layer_stride = stride if stride is not None else 0


def pack_vision(sd, device, weight_format="bf16"):
    """OpenAI-named state-dict -> (Tower descriptor, uint8 device blob) for the vision tower.
    weight_format="fp8": the four linear layers of every block as OCP e4m3 + per-output-channel scales
    (BASELINE.json configs[4]); everything else as in the bf16 layout."""
    if weight_format not in ("bf16", "fp8"):
        raise ValueError("weight_format must be 'bf16' or 'fp8'")
    d = infer_dims(sd)
    W, P = d["v_width"], d["patch"]
    if W % 64 or (4 * W) % 128:
        raise ValueError(f"vision width {W}: must be a multiple of 64")
    tw = _lib.Tower()
    tw.abi_version, tw.kind = _lib.ABI_VERSION, 0
    tw.width, tw.layers, tw.heads, tw.mlp = W, d["v_layers"], W // 64, 4 * W
    tw.embed, tw.tokens, tw.patch, tw.res = d["embed"], d["v_tokens"], P, d["res"]
    k = 3 * P * P
    tw.patch_k = (k + 63) // 64 * 64
    blob = _Blob()
    conv = sd["visual.conv1.weight"].reshape(W, k)
    if tw.patch_k != k:
        conv = torch.cat([conv, torch.zeros(W, tw.patch_k - k)], dim=1)
    bf, f32 = torch.bfloat16, torch.float32
    tw.off_patch_w = blob.put(conv, bf)
    tw.off_cls = blob.put(sd["visual.class_embedding"], f32)
    tw.off_pos = blob.put(sd["visual.positional_embedding"], f32)
    tw.off_ln_pre_w = blob.put(sd["visual.ln_pre.weight"], f32)
    tw.off_ln_pre_b = blob.put(sd["visual.ln_pre.bias"], f32)
    tw.weight_format = 1 if weight_format == "fp8" else 0
    _pack_layers(tw, blob, sd, "visual.transformer", W, d["v_layers"], fp8=weight_format == "fp8")
    tw.off_ln_post_w = blob.put(sd["visual.ln_post.weight"], f32)
    tw.off_ln_post_b = blob.put(sd["visual.ln_post.bias"], f32)
    tw.off_out_proj = blob.put(sd["visual.proj"].t(), bf)          # [E][W]
    buf, total = blob.materialise(device)
    tw.blob_bytes = total
    return tw, buf


def pack_text(sd, device):
    """OpenAI-named state-dict -> (Tower descriptor, uint8 device blob) for the text tower."""
    d = infer_dims(sd)
    T = d["t_width"]
    tw = _lib.Tower()
    tw.abi_version, tw.kind = _lib.ABI_VERSION, 1
    tw.width, tw.layers, tw.heads, tw.mlp = T, d["t_layers"], T // 64, 4 * T
    tw.embed, tw.tokens, tw.vocab = d["embed"], d["ctx"], d["vocab"]
    blob = _Blob()
    bf, f32 = torch.bfloat16, torch.float32
    tw.off_tok_emb = blob.put(sd["token_embedding.weight"], f32)
    tw.off_pos = blob.put(sd["positional_embedding"], f32)
    _pack_layers(tw, blob, sd, "transformer", T, d["t_layers"])
    tw.off_ln_post_w = blob.put(sd["ln_final.weight"], f32)
    tw.off_ln_post_b = blob.put(sd["ln_final.bias"], f32)
    tw.off_out_proj = blob.put(sd["text_projection"].t(), bf)      # [E][T]
    buf, total = blob.materialise(device)
    tw.blob_bytes = total
    return tw, buf


def bf16_round_state_dict(sd):
    """The state-dict as the HIP path sees it: GEMM weights rounded to bf16 (then back to f32).
    Parity tests feed THIS to the fp32 oracle so that the comparison isolates activation
    rounding and accumulation order from the (deliberate) bf16 storage of the weights."""
    out = {}
    for k, v in sd.items():
        gemm = (k.endswith(("in_proj_weight", "out_proj.weight", "c_fc.weight", "c_proj.weight"))
                or k in ("visual.conv1.weight", "visual.proj", "text_projection"))
        out[k] = v.to(torch.bfloat16).float() if gemm else v.clone()
    return out
