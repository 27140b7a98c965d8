"""FP8 tower: embeddings of B random images to a file (development aid / child of test_fp8_fused_producers...).
usage: [CLIPMI_FP8_FUSE=0] python tools/fp8_fuse_check.py B out.pt"""
import sys, os, time
os.environ.setdefault("CLIPMI_DEV_LIB", "1")   # A/B knobs: development library only
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
B = int(sys.argv[1]); out = sys.argv[2]
dev = torch.device("cuda:0")
sd = clipmi.weights.random_state_dict("ViT-B/32", seed=11)
model = clipmi.CLIP(sd, device=dev, vision_weights="fp8")
g = torch.Generator(device="cpu"); g.manual_seed(B)
images = torch.randn(B, 3, 224, 224, generator=g)
e = model.encode_image(images)
torch.cuda.synchronize()
x = images.to(dev)
for _ in range(2): model.encode_image(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): model.encode_image(x)
torch.cuda.synchronize()
print(f"B={B}: {B * 5 / (time.perf_counter() - t0):.0f} images/s", flush=True)
torch.save(e.cpu(), out)
