"""How long a decode worker's stage_jpeg takes on a photo-sized file, alone and with 16 processes at once (development aid)."""
import io, os, sys, time, tempfile, multiprocessing as mp
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "cli-p_amd"))
import decode_worker as dw

def work(path, reps, q):
    region = np.zeros(4 << 20, np.uint8)
    dw.stage_jpeg(path, 224, region)
    t0 = time.time()
    for _ in range(reps):
        dw.stage_jpeg(path, 224, region)
    q.put((time.time() - t0) / reps * 1e3)

if __name__ == "__main__":
    from PIL import Image
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:1500, 0:2000]
    a = np.clip(np.stack([127 + 100 * np.sin(xx / 9.0 + yy / 17.0), 127 + 100 * np.cos(xx / 13.0 - yy / 7.0), (xx * 3 + yy * 2) % 256], -1) + rng.normal(0, 12, (1500, 2000, 3)), 0, 255).astype(np.uint8)
    d = tempfile.mkdtemp(); p = os.path.join(d, "a.jpg"); Image.fromarray(a).save(p, quality=85)
    print(os.path.getsize(p) >> 10, "KB", "cpus", len(os.sched_getaffinity(0)))
    for n in (1, 4, 16):
        q = mp.Queue(); ps = [mp.Process(target=work, args=(p, 50, q)) for _ in range(n)]
        [x.start() for x in ps]; r = [q.get() for _ in ps]; [x.join() for x in ps]
        print(n, "processes: ms per file", ["%.2f" % v for v in sorted(r)][:4], "...", "%.2f" % max(r))
