"""Does running two independent batches on two HIP streams overlap the HBM-bound kernels of one
with the MFMA-bound kernels of the other? (development aid)"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
dev = torch.device("cuda:0")
sd = clipmi.weights.random_state_dict("ViT-B/32", seed=0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 435
models = [clipmi.CLIP(sd, device=dev) for _ in range(2)]
xs = [torch.randint(0, 256, (B, 3, 224, 224), device=dev, dtype=torch.uint8) for _ in range(2)]
streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
def run(nstreams, reps):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        for i in range(nstreams):
            with torch.cuda.stream(streams[i]):
                models[i].encode_image(xs[i], normalize=True)
    for s in streams[:nstreams]:
        torch.cuda.current_stream().wait_stream(s)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)
for n in (1, 2):
    run(n, 2)
    ms = run(n, 6)
    print(f"B={B} streams={n}: {n*6*B/ms*1e3:.0f} img/s", flush=True)
