"""CPU checks that pin oracle/clip_oracle.py: its outputs on seeded weights equal the committed
golden vectors, which an independent implementation (transformers' CLIP classes) produced —
tests/golden/make_clip_golden.py. Tolerance 2e-5 absolute on outputs of magnitude ~10 (fp32,
different op order)."""
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


@pytest.mark.parametrize("name", list(clip_case.CASES))
def test_oracle_matches_independent_golden(name):
    g = np.load(os.path.join(GOLD, f"clip_{name}.npz"))
    assert str(g["torch_version"]) == torch.__version__, "goldens were made with another torch: regenerate"
    sd = clip_case.state_dict(name)
    images, ids = clip_case.inputs(name)
    assert float(images.double().sum().item()) == float(g["image_checksum"])
    assert np.array_equal(ids.numpy(), g["ids"])
    img = clip_oracle.encode_image(sd, images).numpy()
    txt = clip_oracle.encode_text(sd, ids).numpy()
    assert np.abs(img).max() > 1.0 and np.abs(txt).max() > 1.0         # not vacuous
    assert np.abs(img - g["image_embeds"]).max() < 2e-5 * max(1.0, np.abs(img).max())
    assert np.abs(txt - g["text_embeds"]).max() < 2e-5 * max(1.0, np.abs(txt).max())


def test_text_rows_after_eot_do_not_matter():
    """Causal mask: tokens after EOT cannot influence the pooled row."""
    sd = clip_case.state_dict("toy_seed0")
    _, ids = clip_case.inputs("toy_seed0")
    a = clip_oracle.encode_text(sd, ids)
    ids2 = ids.clone()
    ids2[0, 6:] = 3          # junk after EOT at position 5 (all < EOT id)
    b = clip_oracle.encode_text(sd, ids2)
    assert torch.equal(a[0], b[0])


def test_normalize_helpers():
    x = torch.tensor([[3.0, 4.0], [0.0, 2.0]])
    assert torch.allclose(clip_oracle.normalize_rows(x), torch.tensor([[0.6, 0.8], [0.0, 1.0]]))
    z = np.zeros((1, 4), np.float32)
    assert clip_oracle.normalize_query(z) is z


def test_hf_named_checkpoint_maps_onto_openai_names(tmp_path):
    """weights.from_hf_state_dict / load_state_dict on a Hugging Face CLIPModel checkpoint (random, built
    from a config: no network): the mapped state-dict through the oracle equals the HF model's own output."""
    import clipmi
    from safetensors.torch import save_file
    from transformers import CLIPConfig, CLIPModel, CLIPTextConfig, CLIPVisionConfig
    torch.manual_seed(0)
    vc = dict(hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2, image_size=64,
              patch_size=32, hidden_act="quick_gelu", layer_norm_eps=1e-5)
    tc = dict(vocab_size=512, hidden_size=128, intermediate_size=512, num_hidden_layers=2, num_attention_heads=2,
              max_position_embeddings=16, hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=511, bos_token_id=510,
              pad_token_id=0)
    model = CLIPModel(CLIPConfig(text_config=tc, vision_config=vc, projection_dim=128)).eval()
    for p_ in model.parameters():
        p_.data.normal_(0, 0.05)
    path = str(tmp_path / "hf_clip.safetensors")
    save_file({k: v.contiguous() for k, v in model.state_dict().items() if "position_ids" not in k}, path)
    sd = clipmi.weights.load_state_dict(path)
    assert clipmi.weights.infer_dims(sd) == dict(v_width=128, v_layers=2, patch=32, res=64, v_tokens=5, t_width=128,
                                                 t_layers=2, ctx=16, vocab=512, embed=128)
    images, ids = clip_case.inputs("toy_seed0")
    with torch.no_grad():
        want_i = model.get_image_features(pixel_values=images)
        want_t = model.get_text_features(input_ids=ids)
    want_i = want_i if torch.is_tensor(want_i) else want_i.pooler_output
    want_t = want_t if torch.is_tensor(want_t) else want_t.pooler_output
    assert (clip_oracle.encode_image(sd, images) - want_i).abs().max() < 1e-4
    assert (clip_oracle.encode_text(sd, ids) - want_t).abs().max() < 1e-4


def _module_tree(sd):
    """nn.Module hierarchy whose state_dict() has exactly the given dotted names (what the upstream TorchScript
    archive ViT-B-32.pt exposes through torch.jit.load(...).state_dict())."""
    root = torch.nn.Module()
    for k, v in sd.items():
        m = root
        parts = k.split(".")
        for p_ in parts[:-1]:
            if not hasattr(m, p_):
                m.add_module(p_, torch.nn.Module())
            m = getattr(m, p_)
        m.register_parameter(parts[-1], torch.nn.Parameter(v.clone(), requires_grad=False))
    return root


def test_torchscript_archive_and_pickle_loaders(tmp_path):
    """weights.load_state_dict's TorchScript branch (the format of the reference's ViT-B-32.pt, build-index.py:18):
    a seeded toy model saved with torch.jit.save comes back name for name, bit for bit; the pickled state-dict
    branch loads with weights_only (no arbitrary unpickling) and refuses a pickle that carries other objects."""
    import clipmi
    sd = clip_case.state_dict("toy_seed0")
    arch = str(tmp_path / "toy.pt")
    torch.jit.save(torch.jit.script(_module_tree(sd)), arch)
    back = clipmi.weights.load_state_dict(arch)
    assert set(back) == set(sd) and all(torch.equal(back[k], sd[k].float()) for k in sd)
    assert clipmi.weights.infer_dims(back) == clipmi.weights.infer_dims(sd)
    images, ids = clip_case.inputs("toy_seed0")
    assert torch.equal(clip_oracle.encode_image(back, images), clip_oracle.encode_image(sd, images))
    pk = str(tmp_path / "toy_sd.pt")
    torch.save({"state_dict": sd}, pk)
    back2 = clipmi.weights.load_state_dict(pk)
    assert all(torch.equal(back2[k], sd[k].float()) for k in sd)

    class Evil:
        def __reduce__(self):
            return (os.system, ("true",))
    bad = str(tmp_path / "evil.pt")
    torch.save({"visual.conv1.weight": sd["visual.conv1.weight"], "x": Evil()}, bad)
    with pytest.raises(Exception):
        clipmi.weights.load_state_dict(bad)


def test_mx_block_emulation_follows_the_stated_rule():
    """oracle/clip_oracle._fp8_mx_rows (the yardstick of the FP8 tower's parity tests) restates the product's MX quantiser
    (csrc/gemm.hpp fp8mx_*): per 32 values scale 2^(e - 7) with e = floor(log2 max|block|), all-zero block -> scale 1,
    RNE to e4m3; checked against an independent ldexp / frexp formulation and on hand-made blocks."""
    from oracle import clip_oracle
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(64, 256, generator=g) * torch.rand(64, 1, generator=g) * 50).to(torch.bfloat16).float()
    x[3] = 0
    x[4, 32:64] = 0
    x[5, 7] = 3.0e38
    x[6, :32] = torch.tensor([448.0] + [1.0] * 31)
    got = clip_oracle._fp8_mx_rows(x)
    b = x.reshape(64, 8, 32)
    amax = b.abs().amax(-1, keepdim=True)
    _, ex = torch.frexp(amax)                               # amax = m 2^ex, m in [0.5, 1): floor(log2 amax) = ex - 1
    sc = torch.where(amax == 0, torch.ones_like(amax), torch.ldexp(torch.ones_like(amax), (ex - 1 - 7).clamp(min=-127)))
    want = ((b / sc).to(torch.float8_e4m3fn).float() * sc).reshape(64, 256)
    assert torch.equal(got, want)
    assert (got[3] == 0).all() and (got[4, 32:64] == 0).all()
    assert got[6, 0] == 448.0 and got[6, 1] == 1.0          # 448 = 1.75 * 2^8 -> scale 2, 224 and 0.5 are e4m3 values
    blk = (got - x).reshape(64, 8, 32).abs().amax(-1)
    assert (blk <= amax.squeeze(-1) * 2.0 ** -4).all()      # half an ulp of 3 mantissa bits at the top of the block's range


def test_realstats_fixture_has_the_statistics_it_claims():
    """tests/golden/clip_case.py "vitb32_realstats" (VERDICT r04 item 6): what real CLIP residual streams show and seeded
    weights do not - massive channels at a few tokens only, 50-400 x the median magnitude, and a residual norm growing several
    times over the 12 layers - measured on the fp32 oracle's trace, both towers."""
    sd = clip_case.state_dict("vitb32_realstats")
    images, ids = clip_case.inputs("vitb32_realstats")
    for fn, x, few in ((clip_oracle.encode_image, images[:2], 0.07), (clip_oracle.encode_text, ids, 0.02)):
        tr = {}
        fn(sd, x, trace=tr)
        ratios, norms = [], []
        for l in range(12):
            a = tr[f"layer{l}"].abs()
            ratios.append((a.max() / a.median()).item())
            norms.append(tr[f"layer{l}"].norm(dim=-1).mean().item())
        a0 = tr["layer0"].abs()
        massive_rows = (a0.amax(dim=-1) > 50 * a0.median()).float().mean().item()
        assert 50 <= min(ratios) and max(ratios) <= 400, ratios
        assert 0 < massive_rows <= few, massive_rows                 # token-specific in the early layers
        assert norms[11] / norms[0] >= 5.0, norms
