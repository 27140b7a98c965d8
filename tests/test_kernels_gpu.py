"""-m gpu numerics of the individual HIP kernels behind encode_* (through the C ABI debug hooks)
against plain torch fp32 references of the same op computed on the SAME bf16-rounded inputs, so
the only differences are accumulation order and the kernel's own output rounding.
Tolerances: f32 outputs 2e-4 relative to the output scale (f32 accumulation over K <= 3072);
bf16 outputs one bf16 ulp (2^-8 relative) on top of that."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf16(t):
    return t.to(torch.bfloat16)


def _qgelu(x):
    return x * torch.sigmoid(1.702 * x)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (1, 128, 64), (200, 256, 128), (77, 512, 512),
                                   (6400, 768, 768), (6401, 2304, 768), (1000, 768, 3072), (196, 768, 3072),
                                   (25600, 768, 768)])           # 300 tiles of 256 x 256 = 1.17 rounds: whole rounds + remainder
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm_epilogues(clipmi, gpu, M, N, K, epi):
    L = clipmi._lib.lib()
    g = torch.Generator(device="cpu"); g.manual_seed(M * 7 + N + K + epi)
    a = _bf16(torch.randn(M, K, generator=g)).to(gpu)
    w = _bf16(torch.randn(N, K, generator=g) * K ** -0.5).to(gpu)
    bias = torch.randn(N, generator=g).to(gpu)
    ref = a.float() @ w.float().t() + bias
    if epi in (0, 1):
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=gpu)
    elif epi == 2:
        res = torch.randn(M, N, generator=g).to(gpu)
        out = res.clone()
        ref = ref + res
    else:
        out = torch.full((M, N), float("nan"), dtype=torch.float32, device=gpu)
    if epi == 1:
        ref = _qgelu(ref)
    rc = L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, epi, None)
    clipmi._lib.check(rc, "gemm")
    torch.cuda.synchronize()
    got = out.float()
    scale = ref.abs().max().item()
    tol = 2e-4 * scale + (2.0 ** -8) * scale * (epi in (0, 1))
    err = (got - ref).abs().max().item()
    assert torch.isfinite(got).all() and err <= tol, f"M={M} N={N} K={K} epi={epi}: err {err} tol {tol}"
    if M <= 128:
        # M <= 128 runs on the skinny kernel (gemm_skinny.hpp: one wave per 16 x 16 outputs, no LDS): the same bits as the
        # 128 x 128 tiled kernel (algo 1) - a row's result does not depend on how many rows travel with it
        out1 = res.clone() if epi == 2 else torch.full_like(out, float("nan"))
        clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), out1.data_ptr(), M, N, K,
                                                 epi | (1 << 8), None), "gemm algo 1")
        torch.cuda.synchronize()
        assert torch.equal(out1, out), "skinny kernel differs from the tiled kernel's bits"


@pytest.mark.parametrize("M,N,K", [(77, 1536, 512), (77, 512, 2048), (50, 768, 3072), (3, 2048, 512), (128, 528, 96)])
def test_gemm_skinny_shapes(clipmi, gpu, M, N, K):
    """The text tower's shapes for ONE prompt (query-index.py:108) and one image's, incl. N % 128 != 0 / K % 64 != 0,
    which only the skinny kernel takes."""
    L = clipmi._lib.lib()
    g = torch.Generator(device="cpu"); g.manual_seed(M + N + K)
    a = _bf16(torch.randn(M, K, generator=g)).to(gpu)
    w = _bf16(torch.randn(N, K, generator=g) * K ** -0.5).to(gpu)
    bias = torch.randn(N, generator=g).to(gpu)
    ref = a.float() @ w.float().t() + bias
    out = torch.full((M, N), float("nan"), dtype=torch.float32, device=gpu)
    clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, 3, None), "gemm")
    torch.cuda.synchronize()
    assert (out - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()


def test_gemm_layout_asymmetric(clipmi, gpu):
    """A = I-like selector with an ASYMMETRIC W catches a transposed or permuted C write."""
    L = clipmi._lib.lib()
    M, N, K = 128, 128, 128
    a = torch.zeros(M, K); a[torch.arange(M), torch.arange(M) % K] = 1.0
    w = (torch.arange(N).reshape(N, 1) * 3 + torch.arange(K).reshape(1, K) * 0.5).float() / 64.0
    a, w = _bf16(a).to(gpu), _bf16(w).to(gpu)
    out = torch.zeros(M, N, dtype=torch.float32, device=gpu)
    clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, 3, None), "gemm")
    torch.cuda.synchronize()
    assert torch.equal(out, a.float() @ w.float().t())


def test_gemm_rejects_bad_shapes(clipmi, gpu):
    L = clipmi._lib.lib()
    x = torch.zeros(16, device=gpu)
    assert L.clipmi_dbg_gemm_bf16(x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 4, 100, 64, 0, None) == 1
    assert "N % 128" in clipmi._lib.last_error()


@pytest.mark.parametrize("M,W", [(1, 768), (7, 512), (1000, 768), (33, 1024), (5, 128)])
@pytest.mark.parametrize("out_bf16", [0, 1])
def test_layernorm(clipmi, gpu, M, W, out_bf16):
    L = clipmi._lib.lib()
    g = torch.Generator(device="cpu"); g.manual_seed(M + W)
    x = (torch.randn(M, W, generator=g) * 3 + 1.5)
    x[0, 3] = 80.0                                # an outlier channel
    w = 1 + 0.1 * torch.randn(W, generator=g)
    b = 0.1 * torch.randn(W, generator=g)
    ref = torch.nn.functional.layer_norm(x, (W,), w, b, 1e-5)
    xd, wd, bd = x.to(gpu), w.to(gpu), b.to(gpu)
    out = torch.empty(M, W, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=gpu)
    clipmi._lib.check(L.clipmi_dbg_layernorm(xd.data_ptr(), wd.data_ptr(), bd.data_ptr(), out.data_ptr(), M, W, out_bf16, None), "ln")
    torch.cuda.synchronize()
    err = (out.float().cpu() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= (2e-5 + (2.0 ** -8) * out_bf16) * scale


def _attn_ref(qkv, B, L, heads, causal):
    W = heads * 64
    x = qkv.float().reshape(B, L, 3, heads, 64)
    q, k, v = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), x[:, :, 2].transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) * 0.125
    if causal:
        s = s + torch.full((L, L), float("-inf")).triu_(1)
    o = torch.softmax(s, -1) @ v
    return o.transpose(1, 2).reshape(B * L, W)


@pytest.mark.parametrize("B,L,heads,causal", [(3, 50, 12, 0), (2, 77, 8, 1), (5, 5, 2, 0), (4, 16, 2, 1),
                                              (1, 64, 12, 0), (2, 80, 8, 1), (7, 33, 3, 0), (257, 50, 12, 0)])
@pytest.mark.parametrize("no_tr", [0, 1])
def test_attention(clipmi, gpu, B, L, heads, causal, no_tr):
    Lb = clipmi._lib.lib()
    g = torch.Generator(device="cpu"); g.manual_seed(B * 100 + L)
    W = heads * 64
    qkv = _bf16(torch.randn(B * L, 3 * W, generator=g) * 1.5)
    ref = _attn_ref(qkv, B, L, heads, causal)
    qd = qkv.to(gpu)
    out = torch.full((B * L, W), float("nan"), dtype=torch.bfloat16, device=gpu)
    clipmi._lib.check(Lb.clipmi_dbg_attention(qd.data_ptr(), out.data_ptr(), B, L, heads, causal | (no_tr << 1), None), "attn")
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    # P is rounded to bf16 before the second product and the output to bf16: 2 ulps of the scale
    err = (got - ref).abs().max().item()
    assert err <= 3 * (2.0 ** -8) * ref.abs().max().item(), f"err {err} scale {ref.abs().max().item()}"


@pytest.mark.parametrize("B,L,heads", [(3, 50, 12), (257, 50, 12), (5, 49, 2), (2, 52, 3), (870, 50, 12)])
def test_attention52_forms_are_bit_identical(clipmi, gpu, B, L, heads):
    """The three kernels that can run ViT-B/32's attention (one workgroup per (image, head) with a query tile per wave =
    the default; one wave per (image, head); the generic attention_kernel) perform the same arithmetic per
    (query, key, d): identical bits."""
    Lb = clipmi._lib.lib()
    g = torch.Generator(device="cpu"); g.manual_seed(B * 31 + L)
    W = heads * 64
    qd = _bf16(torch.randn(B * L, 3 * W, generator=g) * 1.5).to(gpu)
    outs = []
    for flags in (0, 4):
        out = torch.full((B * L, W), float("nan"), dtype=torch.bfloat16, device=gpu)
        clipmi._lib.check(Lb.clipmi_dbg_attention(qd.data_ptr(), out.data_ptr(), B, L, heads, flags, None), "attn")
        torch.cuda.synchronize()
        outs.append(out)
    assert torch.isfinite(outs[0].float()).all() and torch.equal(outs[0], outs[1])
    if B <= 257:
        ref = _attn_ref(qd.cpu(), B, L, heads, 0)
        assert (outs[0].float().cpu() - ref).abs().max().item() <= 3 * (2.0 ** -8) * ref.abs().max().item()


@pytest.mark.parametrize("B,L,heads", [(2, 197, 12), (1, 257, 16), (2, 577, 16), (3, 81, 2), (1, 128, 1), (1, 129, 3)])
def test_attention_long_sequences(clipmi, gpu, B, L, heads):
    """Flash-style kernel (L > 80): online softmax over 64-key blocks, no mask."""
    Lb = clipmi._lib.lib()
    g = torch.Generator(device="cpu"); g.manual_seed(B * 1000 + L)
    W = heads * 64
    qkv = _bf16(torch.randn(B * L, 3 * W, generator=g) * 1.5)
    qkv[3, :64] *= 6.0                  # one sharp query row: the running max moves between blocks
    ref = _attn_ref(qkv, B, L, heads, 0)
    qd = qkv.to(gpu)
    out = torch.full((B * L, W), float("nan"), dtype=torch.bfloat16, device=gpu)
    clipmi._lib.check(Lb.clipmi_dbg_attention(qd.data_ptr(), out.data_ptr(), B, L, heads, 0, None), "attn")
    torch.cuda.synchronize()
    got = out.float().cpu()
    assert torch.isfinite(got).all()
    err = (got - ref).abs().max().item()
    assert err <= 3 * (2.0 ** -8) * ref.abs().max().item(), f"err {err} scale {ref.abs().max().item()}"


def test_attention_spiked_softmax(clipmi, gpu):
    """One key dominates a query by a large margin (exp underflow of the others) — exercises the
    max-subtraction; result must be that key's V row."""
    Lb = clipmi._lib.lib()
    B, L, heads = 1, 50, 1
    qkv = torch.zeros(B * L, 192)
    qkv[:, 128:] = torch.randn(L, 64)
    qkv[7, :64] = 30.0          # query 7
    qkv[21, 64:128] = 30.0      # key 21 matches it: score 30*30*64/8 = 7200
    qkv = _bf16(qkv)
    out = torch.empty(B * L, 64, dtype=torch.bfloat16, device=gpu)
    qd = qkv.to(gpu)
    clipmi._lib.check(Lb.clipmi_dbg_attention(qd.data_ptr(), out.data_ptr(), B, L, heads, 0, None), "attn")
    torch.cuda.synchronize()
    assert torch.equal(out[7].cpu(), qkv[21, 128:])


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1, 256, 128), (300, 512, 192), (1024, 768, 768),
                                   (6400, 2304, 768), (6401, 768, 3072), (25600, 3072, 768), (12544, 768, 3072)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3])
def test_gemm256_epilogues(clipmi, gpu, M, N, K, epi):
    """The 256x256 pipelined kernel (forced with bit 9 of `epi`), same references and tolerances.
    Run twice: the counted-vmcnt / staggered-barrier pipeline must be deterministic."""
    L = clipmi._lib.lib()
    g = torch.Generator(device="cpu"); g.manual_seed(M * 7 + N + K + epi)
    a = _bf16(torch.randn(M, K, generator=g)).to(gpu)
    w = _bf16(torch.randn(N, K, generator=g) * K ** -0.5).to(gpu)
    bias = torch.randn(N, generator=g).to(gpu)
    ref = a.float() @ w.float().t() + bias
    res = torch.randn(M, N, generator=g).to(gpu) if epi == 2 else None
    if epi == 2:
        ref = ref + res
    if epi == 1:
        ref = _qgelu(ref)
    outs = []
    for _ in range(2):
        if epi in (0, 1):
            out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=gpu)
        elif epi == 2:
            out = res.clone()
        else:
            out = torch.full((M, N), float("nan"), dtype=torch.float32, device=gpu)
        rc = L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, epi | (2 << 8), None)
        clipmi._lib.check(rc, "gemm256")
        torch.cuda.synchronize()
        outs.append(out.float())
    scale = ref.abs().max().item()
    tol = 2e-4 * scale + (2.0 ** -8) * scale * (epi in (0, 1))
    err = (outs[0] - ref).abs().max().item()
    assert torch.isfinite(outs[0]).all() and err <= tol, f"M={M} N={N} K={K} epi={epi}: err {err} tol {tol}"
    assert torch.equal(outs[0], outs[1])


def test_gemm256_layout_asymmetric_and_race_screen(clipmi, gpu):
    """Exact-integer data (every product and sum exact in f32): any stale or early LDS read in the
    DMA pipeline shows as a wrong integer. 30 launches on a shape with many K-tiles."""
    L = clipmi._lib.lib()
    M, N, K = 2048, 1024, 2048
    g = torch.Generator(device="cpu"); g.manual_seed(5)
    a = torch.randint(-3, 4, (M, K), generator=g).float()
    w = torch.randint(-3, 4, (N, K), generator=g).float()
    ref = (a @ w.t()).to(gpu)
    a, w = _bf16(a).to(gpu), _bf16(w).to(gpu)
    for _ in range(30):
        out = torch.zeros(M, N, dtype=torch.float32, device=gpu)
        clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, 3 | (2 << 8), None), "gemm256")
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1, 256, 128), (300, 512, 256), (6400, 2304, 768),
                                   (21750, 2304, 768), (21750, 3072, 768), (70000, 256, 128), (33000, 1024, 1024),
                                   (25601, 3072, 768)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_gemm256p_matches_gemm256(clipmi, gpu, M, N, K, epi):
    """The persistent, role-split kernel (algo 3: loader waves / storer waves, tiles walked by 256
    workgroups) performs the same MFMA sequence per output element as gemm256: bit-identical output,
    including M-edge tiles and workgroups that walk 1, 2, 3 or 4 tiles; twice for determinism.
    epi 2 (residual stream, out += acc + bias) goes out as no-return f32 atomic adds, one per element:
    the same single IEEE add as gemm256's read-modify-write."""
    L = clipmi._lib.lib()
    g = torch.Generator(device="cpu"); g.manual_seed(M * 3 + N + K + epi)
    a = _bf16(torch.randn(M, K, generator=g)).to(gpu)
    w = _bf16(torch.randn(N, K, generator=g) * K ** -0.5).to(gpu)
    bias = torch.randn(N, generator=g).to(gpu)
    ref = a.float() @ w.float().t() + bias
    res = torch.randn(M, N, generator=g).to(gpu) if epi == 2 else None
    if epi == 1:
        ref = _qgelu(ref)
    if epi == 2:
        ref = ref + res
    outs = []
    for algo in (2, 3, 3):
        if epi == 2:
            out = torch.full((M + 1, N), float("nan"), dtype=torch.float32, device=gpu)
            out[:M] = res
        else:
            out = torch.full((M + 1, N), float("nan"), dtype=torch.bfloat16, device=gpu)   # +1 guard row
        rc = L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), M, N, K, epi | (algo << 8), None)
        clipmi._lib.check(rc, "gemm256p")
        torch.cuda.synchronize()
        assert torch.isnan(out[M]).all(), "wrote past row M"
        outs.append(out[:M])
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2])
    scale = ref.abs().max().item()
    err = (outs[1].float() - ref).abs().max().item()
    assert err <= 2e-4 * scale + (2.0 ** -8) * scale * (epi != 2), f"M={M} N={N} K={K} epi={epi}: err {err}"


def test_gemm256p_race_screen(clipmi, gpu):
    """Exact small-integer data (sums exact in f32 and in bf16: |c| <= 9*128 needs 11 bits... kept
    <= 256 by construction): a stale, early or torn LDS read anywhere in the tile-to-tile pipeline
    (next tile's K-tile 0 prefetched under the epilogue, staging through buffer 1) shows as a wrong
    integer. 20 launches, 3 tiles per workgroup, many K-tiles."""
    L = clipmi._lib.lib()
    M, N, K = 12288, 4096, 1024          # 48 x 16 = 768 tiles
    g = torch.Generator(device="cpu"); g.manual_seed(9)
    a = torch.randint(-1, 2, (M, K), generator=g).float()
    w = (torch.rand(N, K, generator=g) < 0.05).float() * torch.randint(-1, 2, (N, K), generator=g).float()
    ref = a.to(gpu) @ w.to(gpu).t()
    assert ref.abs().max().item() <= 256          # exactly representable in bf16
    a, w = _bf16(a).to(gpu), _bf16(w).to(gpu)
    ref = ref.to(torch.bfloat16)
    for _ in range(20):
        out = torch.zeros(M, N, dtype=torch.bfloat16, device=gpu)
        clipmi._lib.check(L.clipmi_dbg_gemm_bf16(a.data_ptr(), w.data_ptr(), None, out.data_ptr(), M, N, K, 0 | (3 << 8), None), "gemm256p")
        torch.cuda.synchronize()
        assert torch.equal(out, ref)


def test_gemm256p_rejects(clipmi, gpu):
    L = clipmi._lib.lib()
    x = torch.zeros(1 << 20, dtype=torch.bfloat16, device=gpu)
    # K = 192: three K-tiles (odd) -> refused; f32 epilogues -> refused
    assert L.clipmi_dbg_gemm_bf16(x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 256, 256, 192, 0 | (3 << 8), None) == 1
    assert L.clipmi_dbg_gemm_bf16(x.data_ptr(), x.data_ptr(), None, x.data_ptr(), 256, 256, 128, 3 | (3 << 8), None) == 1


# ---- LN-folded linear layers (csrc/gemm.hpp): LayerNorm folded into the GEMM that consumes it, residual stream kept
# ---- split as rows of [W bf16 hi | W int8 lo] -------------------------------------------------------------------------------
def _ln_fold_case(gpu, M, W, N, seed):
    g = torch.Generator(device="cpu"); g.manual_seed(seed)
    x = torch.randn(M, W, generator=g) * 2 + 0.3
    x[:, 5] += 40.0                                           # an outlier channel, as in real CLIP residual streams
    gamma = 1 + 0.1 * torch.randn(W, generator=g)
    beta = 0.1 * torch.randn(W, generator=g)
    w = _bf16(torch.randn(N, W, generator=g) * W ** -0.5)
    bias = 0.1 * torch.randn(N, generator=g)
    return x.to(gpu), gamma, beta, w, bias


def _views(x3, W):
    """(hi bf16 [M][W], u uint8 [M][W]) views of split rows x3 (uint8 [M][3 W]: W bf16 values, then W biased remainders)."""
    return x3[:, :2 * W].view(torch.bfloat16), x3[:, 2 * W:]


def _join(x3, W):
    """f32 values of split rows: bits = (bits(hi) << 16) + (u << 8) - 0x8000  (csrc/gemm.hpp split_join)."""
    hi, lo = _views(x3, W)
    bits = (hi.contiguous().view(torch.int16).to(torch.int32) << 16) + (lo.to(torch.int32) << 8) - 0x8000
    return bits.view(torch.float32)


def _split(clipmi, L, x, add=None, x3=None):
    M, W = x.shape
    if x3 is None:
        x3 = torch.zeros(M, 3 * W, dtype=torch.uint8, device=x.device)
    part = torch.full((M, W // 256, 2), float("nan"), dtype=torch.float32, device=x.device)
    clipmi._lib.check(L.clipmi_dbg_split_stats(x.data_ptr(), 1 if add else 0, x3.data_ptr(), part.data_ptr(), M, W, None), "split_stats")
    torch.cuda.synchronize()
    return x3, part


def _check_split(x, x3, part):
    """hi = bf16(x) (round to nearest even); u = clamp(round((bits(x) - (bits(hi) << 16)) / 256) + 128, 0, 255) on the f32 bit
    patterns; hi | lo within 2^-16 of x (2^-15 where the remainder is clamped); part = per-256-column (sum, sum of squares)."""
    W = x.shape[1]
    hi, lo = _views(x3, W)
    assert torch.equal(hi, x.to(torch.bfloat16))
    d = x.contiguous().view(torch.int32) - (hi.contiguous().view(torch.int16).to(torch.int32) << 16)
    want_u = torch.clamp(((d + 128) >> 8) + 128, 0, 255).to(torch.uint8)
    assert torch.equal(lo, want_u)
    back = _join(x3, W)
    clamped = ((d + 128) >> 8) > 127                                   # remainder +128 does not fit: 2^-15 there (0.4 % of values)
    assert ((back - x).abs() <= torch.where(clamped, 2.0 ** -15, 2.0 ** -16) * x.abs() + 1e-37).all()
    if x.numel() > 100000:
        assert clamped.float().mean().item() < 0.01
    xs = x.double().reshape(x.shape[0], W // 256, 256)
    assert (part[..., 0].double() - xs.sum(-1)).abs().max().item() <= 3e-6 * xs.abs().sum(-1).max().item()
    assert ((part[..., 1].double() - (xs * xs).sum(-1)) / (xs * xs).sum(-1)).abs().max().item() <= 3e-6


@pytest.mark.parametrize("M,W", [(1, 768), (7, 512), (1003, 768), (33, 1024), (260, 256)])
def test_split_stats(clipmi, gpu, M, W):
    L = clipmi._lib.lib()
    x, _, _, _, _ = _ln_fold_case(gpu, M, W, 256, M + W)
    x[0, 0], x[0, 1], x[0, 2] = 0.0, -0.0, 1e-30                 # zeros and a value whose bf16 neighbourhood is tiny
    if M > 1:
        x[1, 3] = float(torch.tensor(0x3F807FC0, dtype=torch.int32).view(torch.float32))     # remainder rounds to +128: the clamped case
        x[1, 4] = -3.0e18
    x3, part = _split(clipmi, L, x)
    _check_split(x, x3, part)
    # add form: rows = add + x3, in place
    g = torch.Generator(device="cpu"); g.manual_seed(M)
    add = torch.randn(M, W, generator=g).to(gpu)
    want = add + _join(x3, W)
    x3b, part2 = _split(clipmi, L, add, add=True, x3=x3.clone())
    _check_split(want, x3b, part2)


@pytest.mark.parametrize("M,W,N", [(1, 768, 2304), (77, 512, 1536), (6400, 768, 3072), (6401, 768, 2304), (300, 1024, 4096),
                                   (70000, 768, 2304), (30000, 1024, 3072), (25601, 768, 768)])
@pytest.mark.parametrize("epi", [5, 6])
def test_gemm_ln_folded_consumer(clipmi, gpu, M, W, N, epi):
    """out = [quick_gelu](LayerNorm(x; gamma, beta) W^T + bias) through split_stats + the LN-folded epilogue with the
    packer's folded weights, against torch fp32 on the bf16-rounded plain weights; every kernel that can run the shape
    returns the SAME bits (batch-size invariance of the encoder rests on that)."""
    L = clipmi._lib.lib()
    x, gamma, beta, w, bias = _ln_fold_case(gpu, M, W, N, M + W + N + epi)
    wg, colsum, cb = clipmi.weights.ln_fold_terms(w.float(), bias, gamma, beta)
    wg, colsum, cb = wg.to(gpu), colsum.to(gpu), cb.to(gpu)
    x3, part = _split(clipmi, L, x)
    ref = torch.nn.functional.layer_norm(x, (W,), gamma.to(gpu), beta.to(gpu), 1e-5) @ w.float().to(gpu).t() + bias.to(gpu)
    if epi == 6:
        ref = _qgelu(ref)
    outs = {}
    for algo in (0, 1, 2, 3):
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=gpu)
        rc = L.clipmi_dbg_gemm_ln(x3.data_ptr(), wg.data_ptr(), cb.data_ptr(), colsum.data_ptr(), part.data_ptr(), out.data_ptr(),
                                  M, N, W, epi | (algo << 8), None)
        if rc != 0 and algo == 3:
            assert N > 3840 or (N == 3072 and W == 1024 and False)       # bias + colsum rows + the tile's partials must fit LDS
            continue
        clipmi._lib.check(rc, f"gemm_ln algo {algo}")
        torch.cuda.synchronize()
        outs[algo] = out
    scale = ref.abs().max().item()
    # the A operand is bf16(x) (one rounding of the un-normalised value) and the folded weights are rounded once more:
    # 2^-8 of the output scale each
    err = (outs[0].float() - ref).abs().max().item()
    assert torch.isfinite(outs[0].float()).all() and err <= 3.0 * (2.0 ** -8) * scale, f"err {err} scale {scale}"
    for algo, out in outs.items():
        assert torch.equal(out, outs[0]), f"algo {algo} differs from the default kernel's bits"


@pytest.mark.parametrize("M,N,K", [(1, 768, 768), (77, 512, 2048), (10, 512, 512), (128, 768, 768), (100, 1024, 1024), (17, 768, 3072),
                                   (300, 512, 2048), (6400, 768, 768), (6401, 768, 3072), (70000, 768, 768),
                                   (43500, 768, 3072), (1000, 1024, 1024), (25600, 768, 768), (25601, 768, 3072)])
def test_gemm_resid_ln_producer(clipmi, gpu, M, N, K):
    """Split rows x3 += a W^T + bias with the statistics partials of the new rows: the persistent kernel's fused store pass
    (algo 3) and GEMM-into-scratch + split_stats (algos 1, 2) give identical bits in both outputs (the rejected
    two-workgroups-per-CU form, gemm2w - DESIGN 4.4g - left the tree in round 5); the new rows equal torch's to f32 GEMM
    accuracy + the 2^-16 of the split."""
    L = clipmi._lib.lib()
    g = torch.Generator(device="cpu"); g.manual_seed(M + N + K)
    a = _bf16(torch.randn(M, K, generator=g)).to(gpu)
    w = _bf16(torch.randn(N, K, generator=g) * K ** -0.5).to(gpu)
    bias = torch.randn(N, generator=g).to(gpu)
    x0 = (torch.randn(M, N, generator=g) * 2).to(gpu)
    x0[:, 7] += 30.0
    x30, _ = _split(clipmi, L, x0)
    xold = _join(x30, N)
    res = {}
    for algo in (0, 1, 2, 3):           # 0: the shape's own choice (M <= 128: the skinny kernel + split_stats)
        buf = x30.clone()
        part = torch.full((M, N // 256, 2), float("nan"), dtype=torch.float32, device=gpu)
        tmp = torch.empty(M, N, dtype=torch.float32, device=gpu)
        clipmi._lib.check(L.clipmi_dbg_gemm_resid_ln(a.data_ptr(), w.data_ptr(), bias.data_ptr(), buf.data_ptr(),
                                                     part.data_ptr(), tmp.data_ptr(), M, N, K, algo, None),
                          f"gemm_resid_ln algo {algo}")
        torch.cuda.synchronize()
        res[algo] = (buf, part)
    ref = a.float() @ w.float().t() + bias + xold
    x33, part3 = res[3]
    new = _join(x33, N)
    assert (new - ref).abs().max().item() <= (2e-4 + 2.0 ** -15) * ref.abs().max().item()
    hi3, lo3 = _views(x33, N)
    assert torch.equal(hi3, new.to(torch.bfloat16)) or (hi3.float() - new).abs().max().item() <= 2.0 ** -8 * new.abs().max().item()
    xs = new.double().reshape(M, N // 256, 256)
    assert (part3[..., 0].double() - xs.sum(-1)).abs().max().item() <= 1e-4 * xs.abs().sum(-1).max().item()
    for algo in (0, 1, 2):
        for got, want, what in zip(res[algo], res[3], ("split rows", "part")):
            assert torch.equal(got, want), f"algo {algo} vs the fused store pass: {what} differs"


@pytest.mark.parametrize("M,W,K", [(1, 512, 512), (10, 512, 2048), (77, 512, 512), (50, 768, 3072), (128, 1024, 1024), (17, 256, 1024)])
@pytest.mark.parametrize("epi", [5, 6])
def test_skinny_leaf_producer_and_consumer_equal_the_partials_path(clipmi, gpu, M, W, K, epi):
    """One prompt / one image (M <= 128, round 5): the skinny residual GEMM updates the split rows in place and hands the statistics
    on as 4-column leaves; the skinny LN-folded consumer runs the canonical reduction tree on them. Split rows and the
    consumer's output equal, bit for bit, the residual GEMM + split_stats pass + consumer on partials - so a single prompt still
    equals row 0 of a batch that ran on the tiled kernels."""
    L = clipmi._lib.lib()
    N = 3 * W
    x0, gamma, beta, wc, bc = _ln_fold_case(gpu, M, W, N, M + W + K + epi)
    wg, colsum, cb = clipmi.weights.ln_fold_terms(wc.float(), bc, gamma, beta)
    wg, colsum, cb = wg.to(gpu), colsum.to(gpu), cb.to(gpu)
    g = torch.Generator(device="cpu"); g.manual_seed(M * 3 + K)
    a = _bf16(torch.randn(M, K, generator=g)).to(gpu)
    wr = _bf16(torch.randn(W, K, generator=g) * K ** -0.5).to(gpu)
    br = torch.randn(W, generator=g).to(gpu)
    x3, _ = _split(clipmi, L, x0)
    # reference: residual GEMM (scratch rows + split_stats) then the consumer on partials
    x3_ref = x3.clone()
    part = torch.full((M, W // 256, 2), float("nan"), dtype=torch.float32, device=gpu)
    tmp = torch.empty(M, W, dtype=torch.float32, device=gpu)
    clipmi._lib.check(L.clipmi_dbg_gemm_resid_ln(a.data_ptr(), wr.data_ptr(), br.data_ptr(), x3_ref.data_ptr(), part.data_ptr(),
                                                 tmp.data_ptr(), M, W, K, 0, None), "resid_ln")
    ref = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=gpu)
    clipmi._lib.check(L.clipmi_dbg_gemm_ln(x3_ref.data_ptr(), wg.data_ptr(), cb.data_ptr(), colsum.data_ptr(), part.data_ptr(),
                                           ref.data_ptr(), M, N, W, epi, None), "gemm_ln")
    # leaf path
    x3_new = x3.clone()
    leaf = torch.full((M, W // 4, 2), float("nan"), dtype=torch.float32, device=gpu)
    clipmi._lib.check(L.clipmi_dbg_gemm_resid_ln_leaf(a.data_ptr(), wr.data_ptr(), br.data_ptr(), x3_new.data_ptr(), leaf.data_ptr(),
                                                      M, W, K, None), "resid_ln_leaf")
    out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device=gpu)
    clipmi._lib.check(L.clipmi_dbg_gemm_ln_leaf(x3_new.data_ptr(), wg.data_ptr(), cb.data_ptr(), colsum.data_ptr(), leaf.data_ptr(),
                                                out.data_ptr(), M, N, W, epi, None), "gemm_ln_leaf")
    torch.cuda.synchronize()
    assert torch.equal(x3_new, x3_ref), "split rows of the in-place skinny producer differ from GEMM + split_stats"
    assert torch.isfinite(leaf).all()
    assert torch.equal(out, ref), "consumer on leaves differs from consumer on partials"


@pytest.mark.parametrize("n_px", [224, 336])
def test_resize_crop_on_device_matches_pillow(clipmi, gpu, n_px):
    """Row a2 on the device (csrc/resize.hip + the coefficient tables of decode_worker.resize_plan): bit for bit the pixels
    Pillow's bicubic resize + centre crop give (pipeline.load_uint8 is the host form of the same transform), for
    down- and up-scaling, extreme aspect ratios and sizes that need one pass only."""
    import os, tempfile
    import numpy as np
    from PIL import Image
    rng = np.random.default_rng(11)
    sizes = [(224, 224), (336, 336), (225, 224), (224, 300), (336, 500), (100, 90), (30, 500), (500, 30), (1600, 1200),
             (1200, 1600), (448, 448), (223, 223), (1024, 768), (640, 480), (17, 23)]
    sizes += [(int(rng.integers(20, 1000)), int(rng.integers(20, 1000))) for _ in range(12)]
    images = []
    for k, (w, h) in enumerate(sizes):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        if k % 3 == 1:                                   # smooth content exercises the rounding of long tap sums
            a = ((np.linspace(0, 255, w)[None, :, None] + np.linspace(0, 120, h)[:, None, None]) % 256 * np.ones((1, 1, 3))).astype(np.uint8)
        images.append(a)
    got = clipmi.resize.resize_crop_device(images, n_px, gpu).cpu().numpy()
    d = tempfile.mkdtemp()
    for k, a in enumerate(images):
        p = os.path.join(d, f"{k}.png")
        Image.fromarray(a).save(p)
        want = clipmi.pipeline.load_uint8(p, n_px)
        assert np.array_equal(got[k], want), f"image {k} {a.shape}: {(got[k].astype(int) - want).__abs__().max()}"
