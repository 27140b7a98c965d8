"""-m gpu parity of clipmi_encode_image / clipmi_encode_text (through the C ABI, via the
cli-p_amd.model mirror) against oracle/clip_oracle.py on the same seeded weights and inputs, and
against the committed golden vectors.

Stated tolerance (floating point; BASELINE.json asks for one): the HIP path stores GEMM weights
and GEMM-input activations in bf16 with f32 accumulation, f32 residual stream, f32
LayerNorm/softmax. Its deviation from the fp32 oracle (fed the SAME bf16-rounded weights) must be
within 3x the deviation that the oracle itself shows when ITS matrix-product inputs are rounded
to bf16 (measured in the test, typically 1e-2..4e-2 absolute on outputs of magnitude ~5), and the
row-wise cosine similarity to the oracle must be >= 0.9995 (>= 0.999 against the golden file,
which used unrounded weights)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, GOLD)
sys.path.insert(0, ROOT)
import clip_case  # noqa: E402
from oracle import clip_oracle  # noqa: E402

pytestmark = pytest.mark.gpu


def _cos(a, b):
    return torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=-1)


def _tolerances(clipmi, sd, fn, x):
    sdr = clipmi.weights.bf16_round_state_dict(sd)
    ref = fn(sdr, x)
    with clip_oracle.act_round(torch.bfloat16):
        emu = fn(sdr, x)
    return ref, (emu - ref).abs().max().item()


@pytest.mark.parametrize("name", ["toy_seed0", "vitb32_seed0", "vitb32_outlier", "vitb32_realstats", "toyl14_seed3", "vitb16_seed2", "vitl14_seed4"])
def test_encode_image_matches_oracle(clipmi, gpu, name):
    sd = clip_case.state_dict(name)
    images, _ = clip_case.inputs(name)
    model = clipmi.CLIP(sd, device=gpu)
    got = model.encode_image(images).cpu()
    ref, noise = _tolerances(clipmi, sd, clip_oracle.encode_image, images)
    err = (got - ref).abs().max().item()
    cos = _cos(got, ref).min().item()
    print(f"{name}: image err {err:.4g} (bf16-emulation noise {noise:.4g}), min cosine {cos:.6f}")
    assert torch.isfinite(got).all()
    assert err <= 3 * noise + 1e-3, f"err {err} vs measured bf16 noise {noise}"
    assert cos >= 0.9995
    gold = torch.from_numpy(np.load(os.path.join(GOLD, f"clip_{name}.npz"))["image_embeds"])
    assert _cos(got, gold).min().item() >= 0.999
    # fused normalise == build-index.py:50 on the unfused result
    gotn = model.encode_image(images, normalize=True).cpu()
    assert torch.allclose(gotn, got / got.norm(dim=-1, keepdim=True), atol=2e-6)


@pytest.mark.parametrize("name", ["toy_seed0", "vitb32_seed0", "vitb32_outlier", "vitb32_realstats", "toyl14_seed3", "vitl14_seed4"])
def test_encode_text_matches_oracle(clipmi, gpu, name):
    sd = clip_case.state_dict(name)
    _, ids = clip_case.inputs(name)
    model = clipmi.CLIP(sd, device=gpu)
    got = model.encode_text(ids).cpu()
    ref, noise = _tolerances(clipmi, sd, clip_oracle.encode_text, ids)
    err = (got - ref).abs().max().item()
    cos = _cos(got, ref).min().item()
    print(f"{name}: text err {err:.4g} (bf16-emulation noise {noise:.4g}), min cosine {cos:.6f}")
    assert err <= 3 * noise + 1e-3 and cos >= 0.9995
    gold = torch.from_numpy(np.load(os.path.join(GOLD, f"clip_{name}.npz"))["text_embeds"])
    assert _cos(got, gold).min().item() >= 0.999


def test_encode_image_l14_geometry(clipmi, gpu):
    """ViT-L/14@336's geometry in miniature: 14-pixel patches (patch K 588 -> 640), 101 tokens
    (flash attention), checked against the oracle with the same measured tolerance."""
    sd = clipmi.weights.random_state_dict("toy-l14", seed=3)
    g = torch.Generator(device="cpu"); g.manual_seed(9)
    images = torch.randn(5, 3, 140, 140, generator=g)
    model = clipmi.CLIP(sd, device=gpu)
    assert model.vision.patch_k == 640 and model.vision.tokens == 101
    got = model.encode_image(images).cpu()
    ref, noise = _tolerances(clipmi, sd, clip_oracle.encode_image, images)
    err = (got - ref).abs().max().item()
    cos = _cos(got, ref).min().item()
    print(f"toy-l14: image err {err:.4g} (bf16-emulation noise {noise:.4g}), min cosine {cos:.6f}")
    assert err <= 3 * noise + 1e-3 and cos >= 0.9995


def test_encode_image_vit_b16_full_size(clipmi, gpu):
    """ViT-B/16 (197 tokens: flash attention with a 5-key tail block; 16-px patches, K = 768), two images."""
    sd = clipmi.weights.random_state_dict("ViT-B/16", seed=2)
    g = torch.Generator(device="cpu"); g.manual_seed(11)
    images = torch.randn(2, 3, 224, 224, generator=g)
    model = clipmi.CLIP(sd, device=gpu)
    assert model.vision.tokens == 197 and model.vision.patch_k == 768
    got = model.encode_image(images).cpu()
    ref, noise = _tolerances(clipmi, sd, clip_oracle.encode_image, images)
    err = (got - ref).abs().max().item()
    cos = _cos(got, ref).min().item()
    print(f"ViT-B/16: image err {err:.4g} (bf16-emulation noise {noise:.4g}), min cosine {cos:.6f}")
    assert err <= 3 * noise + 1e-3 and cos >= 0.9995


def test_encode_image_vit_l14_336_full_size(clipmi, gpu):
    """BASELINE.json configs[3]: ViT-L/14@336px (24 layers, width 1024, 577 tokens, 768-D) on seeded
    weights, against the oracle (one image: the CPU oracle's two forward passes at this size are most of the test's time)."""
    sd = clipmi.weights.random_state_dict("ViT-L/14@336px", seed=0)
    g = torch.Generator(device="cpu"); g.manual_seed(10)
    images = torch.randn(2, 3, 336, 336, generator=g)
    model = clipmi.CLIP(sd, device=gpu)
    assert model.vision.tokens == 577 and model.embed_dim == 768
    got = model.encode_image(images).cpu()[:1]
    ref, noise = _tolerances(clipmi, sd, clip_oracle.encode_image, images[:1])
    err = (got - ref).abs().max().item()
    cos = _cos(got, ref).min().item()
    print(f"ViT-L/14@336px: image err {err:.4g} (bf16-emulation noise {noise:.4g}), min cosine {cos:.6f}")
    assert err <= 3 * noise + 1e-3 and cos >= 0.9995
    # its 768-D vectors search through the same top-k path (E = 768)
    idx = clipmi.IndexFlatIP(768, device=gpu)
    idx.add(model.encode_image(images, normalize=True))
    D, I = idx.search(model.encode_image(images[:1], normalize=True).cpu().numpy(), 2)
    assert I[0, 0] == 0 and abs(D[0, 0] - 1.0) < 1e-3


def test_encode_text_vit_l14_text_tower(clipmi, gpu):
    """The ViT-L/14 text tower (width 768, 12 heads, 77 tokens): the LN-folded path on a width the ViT-B/32 towers do not
    have on the text side (3 statistics segments, causal attention), three prompts against the oracle."""
    sd = clipmi.weights.random_state_dict("ViT-L/14", seed=4)
    d = clipmi.weights.infer_dims(sd)
    g = torch.Generator(device="cpu"); g.manual_seed(12)
    ids = torch.zeros(3, d["ctx"], dtype=torch.int64)
    for r, eot in enumerate((4, 33, d["ctx"] - 1)):
        ids[r, 0] = d["vocab"] - 2
        ids[r, 1:eot] = torch.randint(1, d["vocab"] - 2, (eot - 1,), generator=g)
        ids[r, eot] = d["vocab"] - 1
    model = clipmi.CLIP(sd, device=gpu)
    assert model.text.width == 768 and model.text.ln_fold == 1
    got = model.encode_text(ids).cpu()
    ref, noise = _tolerances(clipmi, sd, clip_oracle.encode_text, ids)
    err = (got - ref).abs().max().item()
    cos = _cos(got, ref).min().item()
    print(f"ViT-L/14 text: err {err:.4g} (bf16-emulation noise {noise:.4g}), min cosine {cos:.6f}")
    assert err <= 3 * noise + 1e-3 and cos >= 0.9995


def test_encode_image_batch_invariance_and_dtypes(clipmi, gpu):
    """Rows do not depend on their batch neighbours or on the batch size (M-tail handling), and
    uint8 input with the fused transform tail equals pre-normalised f32 input."""
    sd = clip_case.state_dict("vitb32_seed0")
    model = clipmi.CLIP(sd, device=gpu)
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    u8 = torch.randint(0, 256, (131, 3, 224, 224), generator=g, dtype=torch.uint8)
    mean = torch.tensor(clipmi.model.CLIP_MEAN).reshape(1, 3, 1, 1)
    std = torch.tensor(clipmi.model.CLIP_STD).reshape(1, 3, 1, 1)
    f32 = (u8.float() / 255.0 - mean) / std
    a = model.encode_image(u8).cpu()
    b = model.encode_image(f32).cpu()
    assert _cos(a, b).min().item() >= 0.99999 and (a - b).abs().max().item() < 2e-2
    one = model.encode_image(f32[5:6]).cpu()
    three = model.encode_image(f32[4:7]).cpu()
    assert torch.equal(one[0], b[5]) and torch.equal(three[1], b[5])
    ref = clip_oracle.encode_image(clipmi.weights.bf16_round_state_dict(sd), f32[:2])
    assert _cos(b[:2], ref).min().item() >= 0.9995


def test_encode_text_rows_after_eot_are_ignored(clipmi, gpu):
    sd = clip_case.state_dict("vitb32_seed0")
    _, ids = clip_case.inputs("vitb32_seed0")
    model = clipmi.CLIP(sd, device=gpu)
    a = model.encode_text(ids).cpu()
    ids2 = ids.clone(); ids2[0, 6:] = 11
    b = model.encode_text(ids2).cpu()
    assert torch.equal(a[0], b[0])
    # Q = 1 (the reference's query shape) equals row 0 of the batch: host ids run the tower on EOT + 1 = 6 positions only and
    # on the skinny GEMM kernel (M = 6 rows), device-resident ids on all 77 positions - the same bits either way
    assert torch.equal(model.encode_text(ids[:1]).cpu()[0], a[0])
    assert torch.equal(model.encode_text(ids[:1].to(gpu)).cpu()[0], a[0])
    assert torch.equal(model.encode_text(ids.to(gpu)).cpu(), a)


def test_end_to_end_index_and_query(clipmi, gpu, topk_oracle):
    """cfg-1 in miniature: encode images, store normalised rows, image-similarity query
    (query-index.py:86-99) and text query against the index; ids bit-exact vs the oracle top-k
    over the SAME stored vectors."""
    sd = clip_case.state_dict("vitb32_seed0")
    model = clipmi.CLIP(sd, device=gpu)
    g = torch.Generator(device="cpu"); g.manual_seed(3)
    imgs = torch.randn(64, 3, 224, 224, generator=g)
    feats = model.encode_image(imgs, normalize=True)
    idx = clipmi.IndexFlatIP(512, device=gpu)
    idx.add(feats)
    stored = feats.cpu().numpy()
    D, I = idx.search(stored[10:11], 11)
    Ds, Is = topk_oracle.topk(stored, stored[10:11], 11)
    assert np.array_equal(I, Is) and np.array_equal(D.view(np.uint32), Ds.view(np.uint32)) and I[0, 0] == 10
    _, ids = clip_case.inputs("vitb32_seed0")
    q = model.encode_text(ids[:1], normalize=True).cpu().numpy()
    D, I = idx.search(q, 11)
    Ds, Is = topk_oracle.topk(stored, q, 11)
    assert np.array_equal(I, Is) and np.array_equal(D.view(np.uint32), Ds.view(np.uint32))


def test_real_checkpoint_parity_opt_in(clipmi, gpu):
    """SURVEY.md §8c last row: with CLIPMI_REAL_WEIGHTS=/path/to/ViT-B-32.pt (the TorchScript archive the
    reference's clip.load reads, build-index.py:18) the HIP bf16 path is compared with the fp32 oracle on the REAL
    weights. No checkpoint exists offline, so this reports itself as skipped unless the variable is set."""
    path = os.environ.get("CLIPMI_REAL_WEIGHTS")
    if not path:
        pytest.skip("CLIPMI_REAL_WEIGHTS not set: real-weight parity unpinned (no checkpoint offline)")
    sd = clipmi.weights.load_state_dict(path)
    d = clipmi.weights.infer_dims(sd)
    g = torch.Generator(device="cpu"); g.manual_seed(77)
    images = torch.randn(4, 3, d["res"], d["res"], generator=g)
    ids = torch.zeros(3, d["ctx"], dtype=torch.int64)
    for r, eot in enumerate((5, 20, d["ctx"] - 1)):
        ids[r, 0] = d["vocab"] - 2
        ids[r, 1:eot] = torch.randint(1, d["vocab"] - 2, (eot - 1,), generator=g)
        ids[r, eot] = d["vocab"] - 1
    model = clipmi.CLIP(sd, device=gpu)
    for got, fn, x, what in ((model.encode_image(images).cpu(), clip_oracle.encode_image, images, "image"),
                             (model.encode_text(ids).cpu(), clip_oracle.encode_text, ids, "text")):
        ref, noise = _tolerances(clipmi, sd, fn, x)
        err = (got - ref).abs().max().item()
        cos = _cos(got, ref).min().item()
        print(f"real weights {what}: err {err:.4g} (bf16-emulation noise {noise:.4g}), min cosine {cos:.6f}")
        assert err <= 3 * noise + 1e-3 and cos >= 0.999


def test_encode_image_two_sequences_in_flight(clipmi, gpu):
    """Round 5: encode_image cuts an input of at least two one-round chunks into one-round chunks that alternate between the caller's
    stream and one internal stream (CLIP.image_lanes). Same bits as one sequence after the other on one stream, whatever comes
    next on the caller's stream sees the finished result, and a caller's own raw stream keeps everything on that stream."""
    model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=gpu)
    one = model.image_chunk(limit=model.max_batch // 2)
    assert one == 435
    B = 3 * one
    lanes = model.image_lanes(B)
    assert [l for _, _, l in lanes] == [0, 1, 0] and lanes[-1][1] == B
    assert model.image_lanes(one) == [(0, one, 0)] and model.image_lanes(2 * one + 5) == [(0, 2 * one + 5, 0)]
    g = torch.Generator(device="cpu"); g.manual_seed(3)
    images = torch.randint(0, 256, (B, 3, 224, 224), generator=g, dtype=torch.uint8).to(gpu)
    got = model.encode_image(images, normalize=True)
    total = got.sum()                                     # enqueued on the caller's stream right behind the call
    model.chunks_in_flight = 1
    ref = model.encode_image(images, normalize=True)
    torch.cuda.synchronize()
    assert torch.equal(got, ref) and torch.isfinite(got).all()
    assert torch.equal(total, ref.sum())
