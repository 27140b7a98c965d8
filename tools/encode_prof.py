"""One batch size of encode_image for rocprofv3 kernel traces (development aid)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import clipmi
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device=dev)
x = torch.randint(0, 256, (B, 3, 224, 224), device=dev, dtype=torch.uint8)
for _ in range(5): model.encode_image(x, normalize=True)
torch.cuda.synchronize()
