import sys, os, time
import torch
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/cli-p_amd") else os.getcwd())
import clipmi
dev = torch.device("cuda:0")
model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=dev)
for B in (870, 1000, 1024, 1305, 1740, 2610):
    x = torch.randint(0, 256, (B, 3, 224, 224), device=dev, dtype=torch.uint8)
    res = {}
    for rnd in range(2):
        for fl in (1, 2):
            model.chunks_in_flight = fl
            for _ in range(3): model.encode_image(x, normalize=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(8): model.encode_image(x, normalize=True)
            torch.cuda.synchronize()
            res.setdefault(fl, []).append(8 * B / (time.perf_counter() - t0))
    print(f"B={B}: one stream {max(res[1]):.0f} img/s, two sequences in flight {max(res[2]):.0f} img/s  lanes {model.image_lanes(B)[:4]}", flush=True)
