import io, sys, time
import numpy as np, torch
from PIL import Image
sys.path.insert(0, ".")
import clipmi
from clipmi import jpeg
rng = np.random.default_rng(1)
def enc(a, **kw):
    buf = io.BytesIO(); Image.fromarray(a).save(buf, format="JPEG", **kw); return buf.getvalue()
dev = torch.device("cuda:0")
files = [enc(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8), quality=95) for _ in range(64)]
for label, fl in (("same x870", [files[0]] * 870), ("same x1", [files[0]]), ("img5 x1", [files[5]]), ("img17 x1", [files[17]]), ("64 distinct", files), ("870 mixed", (files * 14)[:870])):
    items = [jpeg.parse(b) for b in fl]
    for _ in range(2):
        out, recs, status = jpeg.decode_device(items, dev); torch.cuda.synchronize()
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.time(); out, recs, status = jpeg.decode_device(items, dev); torch.cuda.synchronize(); ts.append(time.time() - t0)
    print(label, "min wall ms %.2f" % (min(ts) * 1e3), flush=True)
