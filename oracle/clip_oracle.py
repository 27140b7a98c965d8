"""clip_oracle.py — TEST INFRASTRUCTURE, not product code.

CPU fp32 restatement (plain torch ops, no nn.Module) of the two model calls on the hot path:
    model.encode_image(image)   reference build-index.py:49   (then `/ norm`, build-index.py:50)
    model.encode_text(texts)    reference query-index.py:108  (then `normalize`, query-index.py:13-17)
for the model the reference loads with clip.load("ViT-B/32", ...) (build-index.py:18,
query-index.py:21).

The arithmetic lives in a third-party dependency that is ABSENT from /root/reference:
openai/CLIP, installed by the reference from git HEAD, UNPINNED (reference setup.sh:22-24). This
file restates that project's published algorithm (clip/model.py: VisionTransformer, Transformer,
ResidualAttentionBlock with nn.MultiheadAttention, QuickGELU, CLIP.encode_image / encode_text)
over a state-dict with the upstream key names. The reference holds no tests or golden vectors
and no weights are available offline, so REAL-WEIGHT PARITY IS UNPINNED. What pins this oracle:
tests/golden/clip_*.npz, produced by tests/golden/make_clip_golden.py from an INDEPENDENT
implementation of the same architecture (transformers' CLIPVisionModelWithProjection /
CLIPTextModelWithProjection, constructed from a config object, seeded weights) — see
tests/test_oracle_clip.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import torch
import torch.nn.functional as F

LN_EPS = 1e-5   # nn.LayerNorm default, used by every LayerNorm in upstream clip/model.py


def quick_gelu(x):
    # upstream: class QuickGELU: x * torch.sigmoid(1.702 * x)
    return x * torch.sigmoid(1.702 * x)


def _ln(x, sd, name):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], LN_EPS)


_ACT_ROUND = None     # test knob: dtype that GEMM/attention INPUT activations are rounded to


def _r(x):
    """Optional emulation of a reduced-precision activation format at every matrix-product input
    (weights are handled by the caller, e.g. cli-p_amd.weights.bf16_round_state_dict). Used by
    tests to MEASURE the rounding noise a bf16 path must show, which sets the stated tolerance."""
    return x if _ACT_ROUND is None else x.to(_ACT_ROUND).to(x.dtype)


class act_round:
    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        global _ACT_ROUND
        self.prev, _ACT_ROUND = _ACT_ROUND, self.dtype

    def __exit__(self, *a):
        global _ACT_ROUND
        _ACT_ROUND = self.prev


_FP8_LINEAR = False   # test knob: the four linear layers of every block on e4m3 operands (BASELINE.json configs[4])
_FP8_ACT = "product"  # which activation quantiser the emulation uses (linear_fp8)
FP8_MAX = 448.0


def _fp8_rows(t):
    """Per-row e4m3 quantise/dequantise: scale = max|row| / 448 (1 for a zero row), q = RNE(t * (1 / scale)) -
    the rule of quantize_rows_fp8_kernel (activations) and weights.quantize_fp8_rows (weights)."""
    amax = t.abs().amax(dim=-1, keepdim=True)
    scale = torch.where(amax > 0, amax / FP8_MAX, torch.ones_like(amax))
    return (t * (1.0 / scale)).to(torch.float8_e4m3fn).to(t.dtype) * scale


def _fp8_mx_rows(t):
    """MX block quantise/dequantise - the rule of the product's fp8mx_scale_byte / fp8mx_inv / fp8_pack8 (csrc/gemm.hpp;
    attention52x4's output stage, the persistent QuickGELU GEMM's store pass, quantize_rows_fp8mx_kernel): one e8m0 scale
    2^(e - 7) per 32 consecutive values of a row, e = floor(log2 max|block|) read off the float's exponent field (scaled
    values lie in [128, 256): e4m3 holds 448), scale 1 for an all-zero block, exponent clamped at the bottom of the e8m0
    range; values RNE to e4m3."""
    shape = t.shape
    b = t.reshape(-1, shape[-1] // 32, 32)
    amax = b.abs().amax(dim=-1, keepdim=True)
    e = (amax.contiguous().view(torch.int32) >> 23) & 0xff                      # biased exponent of the block maximum
    sb = torch.where(amax == 0, torch.full_like(e, 127), torch.clamp(e - 7, min=0))
    inv = ((254 - sb) << 23).view(torch.float32)                                 # 2^(127 - sb), exact
    scale = torch.where(sb > 0, (sb << 23).view(torch.float32), torch.full_like(amax, 2.0 ** -127))
    return ((b * inv).to(torch.float8_e4m3fn).to(t.dtype) * scale).reshape(shape)


class linear_fp8:
    """Context manager: emulate the FP8 linear layers (products of e4m3 values are exact in f32, so an f32 matmul of
    the dequantised operands is what the FP8 matrix cores compute, up to summation order).
    act="row": every activation with one scale per row (round 2's quantiser; kept as the looser yardstick);
    act="product" (default): the quantisers that run since round 3 - LayerNorm's rows (the operands of in_proj and c_fc) keep
    a row scale, the rows a GEMM takes straight from a producing kernel (attention's output -> out_proj, the QuickGELU rows ->
    c_proj) travel with MX block scales (_fp8_mx_rows);
    act="fold": round 4's tower - as "product", but ln_1 / ln_2 are FOLDED into in_proj / c_fc (_ln_linear): the A operand is
    the un-normalised f32 residual row as e4m3 with MX block scales, the weights e4m3(W diag(gamma)) per output channel, and
    the result rstd (acc - mean colsum) + (W beta + b) with colsum the row sums of the dequantised weights."""

    def __init__(self, act="product"):
        assert act in ("row", "product", "fold")
        self.act = act

    def __enter__(self):
        global _FP8_LINEAR, _FP8_ACT
        self.prev, _FP8_LINEAR = (_FP8_LINEAR, _FP8_ACT), True
        _FP8_ACT = self.act
        return self

    def __exit__(self, *a):
        global _FP8_LINEAR, _FP8_ACT
        _FP8_LINEAR, _FP8_ACT = self.prev




def _linear(x, w, b, mx=False):
    """mx: this operand comes straight from a producing kernel (attention, QuickGELU) - MX block scales in the FP8 tower."""
    if _FP8_LINEAR:
        xq = _fp8_mx_rows(_r(x)) if (mx and _FP8_ACT in ("product", "fold")) else _fp8_rows(_r(x))
        return xq @ _fp8_rows(w).t() + b
    return _r(x) @ w.t() + b


def _ln_linear(x, sd, ln, w, b):
    """linear(LayerNorm(x)) - upstream's ln_1 -> in_proj and ln_2 -> c_fc. With linear_fp8(act="fold") in the product's folded
    FP8 form (csrc/gemm.hpp "LN-folded linear layers", weights.ln_fold_terms_fp8): e4m3 MX blocks of the un-normalised row."""
    if _FP8_LINEAR and _FP8_ACT == "fold":
        gamma, beta = sd[ln + ".weight"], sd[ln + ".bias"]
        mean = x.mean(dim=-1, keepdim=True)
        rstd = torch.rsqrt(x.var(dim=-1, unbiased=False, keepdim=True) + LN_EPS)
        wq = _fp8_rows(w * gamma[None, :])
        return rstd * (_fp8_mx_rows(x) @ wq.t() - mean * wq.sum(dim=1)) + (w @ beta + b)
    return _linear(_ln(x, sd, ln), w, b)


def _attention(x, sd, p, heads, mask):
    """ln_1 + nn.MultiheadAttention(width, heads) self-attention as ResidualAttentionBlock.attention
    calls it on ln_1(x) (need_weights=False, attn_mask=mask); x is the block's input [B, L, W] here (upstream uses [L, B, W];
    the arithmetic is per (batch, head) and identical)."""
    B, L, W = x.shape
    hd = W // heads
    qkv = _r(_ln_linear(x, sd, p + ".ln_1", sd[p + ".attn.in_proj_weight"], sd[p + ".attn.in_proj_bias"]))
    q, k, v = qkv.split(W, dim=-1)
    q = q.reshape(B, L, heads, hd).transpose(1, 2)
    k = k.reshape(B, L, heads, hd).transpose(1, 2)
    v = v.reshape(B, L, heads, hd).transpose(1, 2)
    s = (q * (hd ** -0.5)) @ k.transpose(-1, -2)          # MHA scales q before the product
    if mask is not None:
        s = s + mask
    a = _r(torch.softmax(s, dim=-1)) @ v
    a = _r(a.transpose(1, 2).reshape(B, L, W))
    return _linear(a, sd[p + ".attn.out_proj.weight"], sd[p + ".attn.out_proj.bias"], mx=True)


def _resblock(x, sd, p, heads, mask):
    # upstream ResidualAttentionBlock.forward: x + attention(ln_1(x)); x + mlp(ln_2(x))
    x = x + _attention(x, sd, p, heads, mask)             # _attention applies ln_1 itself (_ln_linear)
    h = _ln_linear(x, sd, p + ".ln_2", sd[p + ".mlp.c_fc.weight"], sd[p + ".mlp.c_fc.bias"])
    h = _linear(quick_gelu(h), sd[p + ".mlp.c_proj.weight"], sd[p + ".mlp.c_proj.bias"], mx=True)
    return x + h


def _n_layers(sd, prefix):
    n = 0
    while f"{prefix}.resblocks.{n}.ln_1.weight" in sd:
        n += 1
    return n


@torch.no_grad()
def encode_image(sd, image, trace=None):
    """upstream CLIP.encode_image -> VisionTransformer.forward. image f32 [B,3,R,R] -> [B,E].
    `trace` (a dict) receives intermediate activations for kernel-level tests."""
    w = sd["visual.conv1.weight"]
    W, P = w.shape[0], w.shape[-1]
    x = F.conv2d(_r(image.to(w.dtype)), w, bias=None, stride=P)          # [B, W, g, g]
    B = x.shape[0]
    x = x.reshape(B, W, -1).permute(0, 2, 1)                     # [B, g*g, W]
    cls = sd["visual.class_embedding"].reshape(1, 1, W).expand(B, 1, W)
    x = torch.cat([cls, x], dim=1) + sd["visual.positional_embedding"]
    if trace is not None:
        trace["embed"] = x.clone()
    x = _ln(x, sd, "visual.ln_pre")
    if trace is not None:
        trace["ln_pre"] = x.clone()
    heads = W // 64
    for i in range(_n_layers(sd, "visual.transformer")):
        x = _resblock(x, sd, f"visual.transformer.resblocks.{i}", heads, None)
        if trace is not None:
            trace[f"layer{i}"] = x.clone()
    x = _ln(x[:, 0, :], sd, "visual.ln_post")
    return _r(x) @ sd["visual.proj"]


@torch.no_grad()
def encode_text(sd, ids, trace=None):
    """upstream CLIP.encode_text. ids int [Q, ctx] -> [Q, E]; causal additive mask (-inf above
    the diagonal, build_attention_mask); pooled at argmax(ids) = the EOT token."""
    ids = ids.long()
    x = sd["token_embedding.weight"][ids] + sd["positional_embedding"]
    Q, L, T = x.shape
    mask = torch.full((L, L), float("-inf")).triu_(1)
    heads = T // 64
    for i in range(_n_layers(sd, "transformer")):
        x = _resblock(x, sd, f"transformer.resblocks.{i}", heads, mask)
        if trace is not None:
            trace[f"layer{i}"] = x.clone()
    x = _ln(x, sd, "ln_final")
    x = x[torch.arange(Q), ids.argmax(dim=-1)]
    return _r(x) @ sd["text_projection"]


def normalize_rows(x):
    """build-index.py:50: x / x.norm(dim=-1, keepdim=True)."""
    return x / x.norm(dim=-1, keepdim=True)


def normalize_query(v):
    """query-index.py:13-17 on a numpy array: whole-array norm, identity below 1e-9."""
    import numpy as np
    n = np.linalg.norm(v)
    return v if n < 0.000000001 else v / n
