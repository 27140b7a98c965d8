"""Two half-batches on two HIP streams, each persistent GEMM capped at half the CUs (CLIPMI_GEMM_GRID=128, development library), the
second stream started a fraction of a layer late: do one half's HBM-bound store passes / attention run beside the other half's
K-loops? (development aid; round 1's CU-mask version and round 5's rerun: DESIGN 8.1)"""
import sys, os, time
os.environ.setdefault("CLIPMI_DEV_LIB", "1")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import clipmi
dev = torch.device("cuda:0")
sd = clipmi.weights.random_state_dict("ViT-B/32", seed=0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 435
lag = int(sys.argv[2]) if len(sys.argv) > 2 else 0          # cycles of torch.cuda._sleep in front of stream 2's first call
NS = int(sys.argv[3]) if len(sys.argv) > 3 else 2
models = [clipmi.CLIP(sd, device=dev) for _ in range(NS)]
xs = [torch.randint(0, 256, (B, 3, 224, 224), device=dev, dtype=torch.uint8) for _ in range(NS)]
outs = [torch.empty((B, 512), dtype=torch.float32, device=dev) for _ in range(NS)]
streams = [torch.cuda.Stream(device=dev) for _ in range(NS)]
def run(nstreams, reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if nstreams >= 2 and lag:
        with torch.cuda.stream(streams[1]):
            torch.cuda._sleep(lag)
    for r in range(reps):
        for i in range(nstreams):
            with torch.cuda.stream(streams[i]):
                models[i].encode_image(xs[i], normalize=True, out=outs[i])
    torch.cuda.synchronize()
    return time.perf_counter() - t0
for n in (1, NS):
    run(n, 3)
    best = min(run(n, 10) for _ in range(3))
    print(f"grid={os.environ.get('CLIPMI_GEMM_GRID', '256')} B={B} streams={n} lag={lag}: {n * 10 * B / best:.0f} img/s", flush=True)
