"""FP8 tower with folded LayerNorms (development library; CLIPMI_DEV_LIB=1 CLIPMI_FP8_LN_FOLD=1): parity against the fp32 oracle
and against the oracle's emulation of exactly this arithmetic, with the tolerances of tests/test_fp8_gpu.py. Prints one line
per fixture ending in "parity ok"; exit code 1 on a miss. Run by tests/test_fp8_gpu.py in a child process."""
import os, sys
os.environ.setdefault("CLIPMI_DEV_LIB", "1")
os.environ.setdefault("CLIPMI_FP8_LN_FOLD", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))     # a test helper: lives under tests/ (it imports the oracle)
import torch
import clipmi
import clip_case
from oracle import clip_oracle
ERR_FACTOR, COS_SLACK_REF, COS_FLOOR = 1.5, 1e-3, 0.99          # tests/test_fp8_gpu.py FP8_*
dev = torch.device("cuda:0")
cosf = lambda a, b: torch.nn.functional.cosine_similarity(a.double(), b.double(), dim=-1).min().item()
ok = True
for name in ("vitb32_seed0", "vitb32_outlier"):
    sd = clip_case.state_dict(name)
    images, _ = clip_case.inputs(name)
    model = clipmi.CLIP(sd, device=dev, vision_weights="fp8")
    assert model.vision.weight_format == 1 and model.vision.ln_fold == 1, "the folded FP8 tower was not selected"
    got = model.encode_image(images).cpu()
    sdr = clipmi.weights.bf16_round_state_dict(sd)
    ref = clip_oracle.encode_image(sdr, images)
    with clip_oracle.act_round(torch.bfloat16), clip_oracle.linear_fp8(act="fold"):
        emu = clip_oracle.encode_image(sdr, images)
    with clip_oracle.act_round(torch.bfloat16), clip_oracle.linear_fp8(act="product"):
        emu_p = clip_oracle.encode_image(sdr, images)
    noise, err = (emu - ref).abs().max().item(), (got - ref).abs().max().item()
    cos, cos_emu = cosf(got, ref), cosf(got, emu)
    good = (torch.isfinite(got).all().item() and err <= ERR_FACTOR * noise + 1e-3 and cos >= max(COS_FLOOR, cosf(emu, ref) - COS_SLACK_REF)
            and cos_emu > cos)
    ok = ok and good
    print(f"{name}: folded fp8 tower err {err:.4g} (emulation noise {noise:.4g}; LayerNorm-pass tower's emulation "
          f"{(emu_p - ref).abs().max().item():.4g}), cosine to fp32 oracle {cos:.5f} (emulation itself {cosf(emu, ref):.5f}), to the "
          f"emulation {cos_emu:.5f}: {'parity ok' if good else 'PARITY MISS'}", flush=True)
sys.exit(0 if ok else 1)
