"""Generates tests/golden/clip_*.npz from an implementation INDEPENDENT of oracle/clip_oracle.py:
transformers' CLIP classes built from config objects (no network, no pretrained files), loaded
with the same seeded weights. Run from the repo root in the build container:
    python tests/golden/make_clip_golden.py
(The upstream openai/CLIP package and weights are not available offline; see oracle header.)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import clip_case  # noqa: E402
import clipmi  # noqa: E402
from transformers import (CLIPTextConfig, CLIPTextModelWithProjection, CLIPVisionConfig,  # noqa: E402
                          CLIPVisionModelWithProjection)


def _map_layers(sd, src, dst, layers, W, out):
    for i in range(layers):
        s, d = f"{src}.resblocks.{i}", f"{dst}.encoder.layers.{i}"
        wq, wk, wv = sd[f"{s}.attn.in_proj_weight"].split(W, dim=0)
        bq, bk, bv = sd[f"{s}.attn.in_proj_bias"].split(W, dim=0)
        out[f"{d}.self_attn.q_proj.weight"], out[f"{d}.self_attn.q_proj.bias"] = wq, bq
        out[f"{d}.self_attn.k_proj.weight"], out[f"{d}.self_attn.k_proj.bias"] = wk, bk
        out[f"{d}.self_attn.v_proj.weight"], out[f"{d}.self_attn.v_proj.bias"] = wv, bv
        out[f"{d}.self_attn.out_proj.weight"] = sd[f"{s}.attn.out_proj.weight"]
        out[f"{d}.self_attn.out_proj.bias"] = sd[f"{s}.attn.out_proj.bias"]
        for a, b in (("ln_1", "layer_norm1"), ("ln_2", "layer_norm2")):
            out[f"{d}.{b}.weight"], out[f"{d}.{b}.bias"] = sd[f"{s}.{a}.weight"], sd[f"{s}.{a}.bias"]
        for a, b in (("c_fc", "fc1"), ("c_proj", "fc2")):
            out[f"{d}.mlp.{b}.weight"], out[f"{d}.mlp.{b}.bias"] = sd[f"{s}.mlp.{a}.weight"], sd[f"{s}.mlp.{a}.bias"]


def hf_models(sd):
    d = clipmi.weights.infer_dims(sd)
    vc = CLIPVisionConfig(hidden_size=d["v_width"], intermediate_size=4 * d["v_width"],
                          num_hidden_layers=d["v_layers"], num_attention_heads=d["v_width"] // 64,
                          image_size=d["res"], patch_size=d["patch"], projection_dim=d["embed"],
                          hidden_act="quick_gelu", layer_norm_eps=1e-5, attn_implementation="eager")
    tc = CLIPTextConfig(vocab_size=d["vocab"], hidden_size=d["t_width"], intermediate_size=4 * d["t_width"],
                        num_hidden_layers=d["t_layers"], num_attention_heads=d["t_width"] // 64,
                        max_position_embeddings=d["ctx"], projection_dim=d["embed"], hidden_act="quick_gelu",
                        layer_norm_eps=1e-5, eos_token_id=d["vocab"] - 1, bos_token_id=d["vocab"] - 2,
                        pad_token_id=0, attn_implementation="eager")
    vm = CLIPVisionModelWithProjection(vc).eval()
    tm = CLIPTextModelWithProjection(tc).eval()
    v = {"vision_model.embeddings.class_embedding": sd["visual.class_embedding"],
         "vision_model.embeddings.patch_embedding.weight": sd["visual.conv1.weight"],
         "vision_model.embeddings.position_embedding.weight": sd["visual.positional_embedding"],
         "vision_model.pre_layrnorm.weight": sd["visual.ln_pre.weight"],
         "vision_model.pre_layrnorm.bias": sd["visual.ln_pre.bias"],
         "vision_model.post_layernorm.weight": sd["visual.ln_post.weight"],
         "vision_model.post_layernorm.bias": sd["visual.ln_post.bias"],
         "visual_projection.weight": sd["visual.proj"].t().contiguous()}
    _map_layers(sd, "visual.transformer", "vision_model", d["v_layers"], d["v_width"], v)
    t = {"text_model.embeddings.token_embedding.weight": sd["token_embedding.weight"],
         "text_model.embeddings.position_embedding.weight": sd["positional_embedding"],
         "text_model.final_layer_norm.weight": sd["ln_final.weight"],
         "text_model.final_layer_norm.bias": sd["ln_final.bias"],
         "text_projection.weight": sd["text_projection"].t().contiguous()}
    _map_layers(sd, "transformer", "text_model", d["t_layers"], d["t_width"], t)
    missing, unexpected = vm.load_state_dict(v, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    missing, unexpected = tm.load_state_dict(t, strict=False)
    assert not unexpected and all("position_ids" in m for m in missing), (missing, unexpected)
    return vm, tm


if __name__ == "__main__":
    torch.manual_seed(0)
    only = sys.argv[1:]                      # optional: regenerate just these cases
    for name in clip_case.CASES:
        if only and name not in only:
            continue
        sd = clip_case.state_dict(name)
        images, ids = clip_case.inputs(name)
        vm, tm = hf_models(sd)
        with torch.no_grad():
            img = vm(pixel_values=images).image_embeds.numpy()
            txt = tm(input_ids=ids).text_embeds.numpy()
        np.savez_compressed(os.path.join(HERE, f"clip_{name}.npz"), image_embeds=img, text_embeds=txt,
                            ids=ids.numpy(), image_checksum=np.float64(images.double().sum().item()),
                            torch_version=torch.__version__)
        print(name, img.shape, float(np.abs(img).max()), txt.shape, float(np.abs(txt).max()))
