#!/usr/bin/env python3
"""bench.py — the hot path of ps-auxw/CLI-P on MI355X: ViT-B/32 encode_image throughput and exact
flat inner-product top-50 search, on synthetic data, one process per GPU.

    python bench.py --gpus N --steps K --warmup W

N > 1 is accepted both ways: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py
--gpus N ...` (RANK / LOCAL_RANK / WORLD_SIZE in the environment), or as plain `python bench.py --gpus N`,
in which case this process — before it has touched a GPU — starts that launcher as a CHILD process, relays
its output (rank 0's JSON line) and exits with its code.

A "step" is one pass of the hot path over one batch: encode a batch of B synthetic 224x224 uint8
images on every rank (weak: per-GPU work fixed), then — timed in its own bracket — one search
batch of Q queries for the best K = 50 + 1 (query-index.py:111: k + offset + 1) over the 10M x 512
f32 matrix split contiguously over the ranks (strong: total rows fixed), with ONE all-gather of the
per-rank partial lists and the K-way merge. Rank 0 prints ONE JSON line.

`value` is images/s (BASELINE.json's first metric, configs[1]); the second metric (queries/s) and its own
roofline and CPU baseline are in the "search" object of the same line. Further objects in the same line:
  encode_sustained      the same encode step for >= 1 M images (configs[1] says "encode 1M images": ~11 s)
  encode_fp8            configs[4] encode half (FP8 linear layers)
  encode_vitl14_336     configs[3]: ViT-L/14@336px, own MFMA roofline
  search_shard_12p5m    configs[4] search half: one rank's 12.5 M-row share of a 100 M x 512 DB (+ the
                        all-gather merge when N > 1: at N = 8 this IS the 100 M-row sharded search)
  cpu_baseline          configs[0] as SURVEY.md §8(d) specifies it (256 JPEGs, B = 1; B = 32 beside it;
                        flat-IP top-11 over the 256 x 512 result for 16 image-id + 16 text queries)
"""
import argparse
import ctypes as C
import io
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import clipmi  # noqa: E402

FLOP_PER_IMAGE = 8_817_623_040          # ViT-B/32, L = 50, 2 FLOP per MAC
FLOP_PER_IMAGE_L14_336 = 381.92e9       # ViT-L/14@336px
PEAK_BF16_TFLOPS = 2500.0               # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0                   # MI355X HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    # 435 images x 50 tokens = 85 row-tiles of 256: the four GEMM shapes of a layer then need 765 / 255 /
    # 1020 / 255 tiles = whole rounds of the 256 CUs (tile quantisation is the first-order batch effect).
    # 870 = 170 row-tiles is the next such size: twice the tiles per workgroup for the persistent GEMM
    # (qkv, c_fc), +2-3 % images/s over 435 on the same box
    ap.add_argument("--batch", type=int, default=870, help="images per GPU per step")
    ap.add_argument("--rows", type=int, default=10_000_000, help="total rows of the flat index")
    ap.add_argument("--queries", type=int, default=64, help="queries per search batch (64 = one pass of the coarse scan)")
    ap.add_argument("--exact-only", action="store_true", help="search with the exact f32 scan only (no coarse pass)")
    ap.add_argument("--coarse", choices=["int8", "bf16"], default="int8",
                    help="coarse copy scanned before the exact f32 re-scoring (results are identical either way)")
    ap.add_argument("--k", type=int, default=50, help="results per query (K = k + 1 is searched)")
    ap.add_argument("--sustained-images", type=int, default=1_000_000, help="images of the encode_sustained leg (0 = skip)")
    ap.add_argument("--shard-rows", type=int, default=12_500_000, help="rows per rank of the search_shard leg (0 = skip)")
    ap.add_argument("--l14-batch", type=int, default=266, help="images per step of the ViT-L/14@336px leg (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp8", action="store_true", help="skip the FP8 (configs[4]) encode measurement")
    ap.add_argument("--large-q", type=int, default=-1,
                    help="queries of the one-call-many-queries leg (default 1024; 0 = skip; --quick skips it unless given)")
    ap.add_argument("--encode-in-flight", type=int, default=2, choices=[1, 2],
                    help="kernel sequences of ONE encode step kept in flight (CLIP.chunks_in_flight): 2 = the product's default "
                         "(two half-batches on two streams), 1 = one sequence on one stream - the form the rocprofv3 summaries under "
                         "profiles/ are taken in, so that per-kernel averages are averages over one shape and no overlap")
    ap.add_argument("--quick", action="store_true",
                    help="headline encode + search only: no sustained / fp8 / ViT-L / shard legs, no CPU baseline "
                         "(profiling passes and tests)")
    a = ap.parse_args()
    if a.quick:
        a.sustained_images = a.shard_rows = a.l14_batch = 0
        a.no_cpu_baseline = a.no_fp8 = True
    if a.large_q < 0:
        a.large_q = 0 if a.quick else 1024
    return a


def spawn_ranks(a):
    """`python bench.py --gpus N` without a launcher: start N ranks as a child `torch.distributed.run` BEFORE this
    process touches the GPU (never re-exec a process that has initialised it), relay, exit with its code."""
    have = torch.cuda.device_count()            # does not initialise the GPU on this image
    if have < a.gpus:
        print(json.dumps({"error": f"--gpus {a.gpus} but only {have} GPU(s) visible", "n_gpus": a.gpus, "value": None}),
              flush=True)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def bracket(dist, world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def timed(fn, steps, warmup, dist, world, spread=None):
    """The contract's bracket (barrier + synchronize on both sides of EXACTLY `steps` steps, max over ranks). `spread`:
    a dict that receives per-step min / median / max of a SECOND, per-step-synchronised run of the same steps (host
    clock around fn + synchronize): where a single step stalls (first touch of a fresh allocation, a clock ramp) it shows
    up here instead of hiding inside the mean. The second run is not part of the reported time."""
    for _ in range(warmup):
        fn()
    bracket(dist, world)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    bracket(dist, world)
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if spread is not None:
        per = []
        for _ in range(min(steps, 20)):
            t1 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            per.append((time.perf_counter() - t1) * 1e3)
        per.sort()
        spread.update({"per_step_synchronised_ms": {"min": per[0], "median": per[len(per) // 2], "max": per[-1], "n": len(per)}})
    return dt


def pmc_traffic(key):
    """(HBM bytes per launch, provenance) from the NEWEST committed rocprofv3 --pmc summary
    (profiles/*_pmc_traffic.json, written by tools/pmc_traffic.py from FETCH_SIZE / WRITE_SIZE with the gfx950
    corrections). A summary taken with another build of libclipmi.so is refused (traffic = null): the file
    records the source digest of the library it profiled."""
    pdir = os.path.join(ROOT, "profiles")
    files = sorted(f for f in os.listdir(pdir) if f.endswith("_pmc_traffic.json")) if os.path.isdir(pdir) else []
    if not files:
        return None, "no profiles/*_pmc_traffic.json"
    f = files[-1]
    try:
        d = json.load(open(os.path.join(pdir, f)))
    except Exception as e:
        return None, f"{f}: unreadable ({e})"
    cur = clipmi.build.source_digest()
    if d.get("lib_digest") != cur:
        return None, f"{f}: taken with library digest {str(d.get('lib_digest'))[:12]}, this build is {cur[:12]} — stale, not reported"
    if key not in d:
        return None, f"{f}: no entry {key}"
    return d[key], f"profiles/{f}"


def make_jpegs(n):
    from PIL import Image
    rng = np.random.default_rng(0)
    blobs = []
    for _ in range(n):
        buf = io.BytesIO()
        Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)).save(buf, format="JPEG", quality=95)
        blobs.append(buf.getvalue())
    return blobs


def files_to_vectors_leg(model, pool):
    """build-index.py's loop end to end: JPEG FILES -> vectors on the host, 870 images per batch (the CLI's default), in the
    product's default form - the decode workers read and parse the files, csrc/jpeg.hip decodes them on the device
    (clipmi_jpeg_decode_rgb8: Pillow's bytes), clipmi_resize_crop_rgb8, HIP encode - and, beside it, with Pillow decoding in the
    worker processes (device_jpeg_kb = 0: rounds 1-4's path, bound by the box's CPU share). Reported beside the headline, never as
    it. The bench's files are uniform noise at quality 95: no end-of-block symbols, the Huffman decode's worst case (its
    subsequences never re-synchronise: a serial chain per image)."""
    import shutil
    import tempfile
    n = 3 * 870
    d = tempfile.mkdtemp(prefix="clipmi_bench_")
    try:
        for i, blob in enumerate(make_jpegs(n)):
            with open(os.path.join(d, f"img_{i:05d}.jpg"), "wb") as f:
                f.write(blob)
        paths = sorted(os.path.join(d, f) for f in os.listdir(d) if f.startswith("img_")) * 4      # 12 batches: every file is decoded four times
        res = {}
        for name, kb in (("device_decode", None), ("pillow_decode", 0)):
            for _ in clipmi.pipeline.encode_files(model, paths[:870], batch=870, pool=pool, device_jpeg_kb=kb):
                pass
            st = {}
            t0 = time.perf_counter()
            got = 0
            for ok, feats, bad in clipmi.pipeline.encode_files(model, paths, batch=870, pool=pool, device_jpeg_kb=kb, stats=st):
                got += len(ok)
            dt = time.perf_counter() - t0
            nb = len(paths) / 870
            res[name] = {"images_per_s": got / dt, "images": got, "device_decoded": int(st.get("jpeg_files", 0)),
                         "stage_ms_per_batch": {"workers": 1e3 * st.get("decode_s", 0.0) / nb, "to_device": 1e3 * st.get("copy_s", 0.0) / nb,
                                                "gpu": 1e3 * st.get("encode_s", 0.0) / nb}}
        # beside the noise files (the Huffman decode's worst case), files with the statistics of photographs: smooth content with
        # a little noise at quality 85 - end-of-block symbols in every block, so the subsequences re-synchronise
        from PIL import Image
        rng = np.random.default_rng(1)
        yy, xx = np.mgrid[0:224, 0:224]
        for i in range(870):
            base = np.stack([127 + 100 * np.sin(xx / (5.0 + i % 7) + yy / 17.0), 127 + 100 * np.cos(xx / 13.0 - yy / (4.0 + i % 5)),
                             (xx * 3 + yy * 2 + i) % 256], -1)
            Image.fromarray(np.clip(base + rng.normal(0, 12, (224, 224, 3)), 0, 255).astype(np.uint8)).save(
                os.path.join(d, f"photo_{i:05d}.jpg"), quality=85)
        ppaths = sorted(os.path.join(d, f) for f in os.listdir(d) if f.startswith("photo_")) * 8
        for _ in clipmi.pipeline.encode_files(model, ppaths[:870], batch=870, pool=pool):
            pass
        st = {}
        t0 = time.perf_counter()
        got = 0
        for ok, feats, bad in clipmi.pipeline.encode_files(model, ppaths, batch=870, pool=pool, stats=st):
            got += len(ok)
        photo = {"value": got / (time.perf_counter() - t0), "unit": "images/s", "images": got, "device_decoded": int(st.get("jpeg_files", 0)),
                 "data": "synthetic 224x224 JPEG files, smooth content + noise (sigma 12), quality 85: ~19 KB each, 870 files x 8"}
    finally:
        shutil.rmtree(d, ignore_errors=True)
    dd = res["device_decode"]
    return {"value": dd["images_per_s"], "unit": "images/s", "images": dd["images"], "decode_processes": pool.n, "batch": 870,
            "photo_like_files": photo,
            "data": "synthetic 224x224 JPEG files (uniform noise, quality 95: ~58 KB each) on local disk, 2610 files x 4",
            "decode": "device (clipmi_jpeg_decode_rgb8) for %d of %d files" % (dd["device_decoded"], dd["images"]),
            "stage_ms_per_batch": dd["stage_ms_per_batch"],
            "bound": ("gpu: the Huffman chains of noise files + the encode step" if dd["device_decoded"] * 2 > dd["images"] else
                      "host decode (the files did not take the device decoder: shared memory too small, or not baseline JPEG)"),
            "pillow_decode_in_workers": {"value": res["pillow_decode"]["images_per_s"], "bound": "host decode",
                                         "stage_ms_per_batch": res["pillow_decode"]["stage_ms_per_batch"]}}


def cpu_baseline_cfg1(sd):
    """BASELINE.json configs[0] on the host cores, as SURVEY.md §8(d) lays it out: 256 synthetic 224x224 JPEGs
    (rng 0, quality 95) -> decode + transform + fp32 encode at B = 1 per image (reference semantics,
    build-index.py:47-50) + normalise, through oracle/clip_oracle.py ("port": clip / faiss are not installable
    here); the same at B = 32 as a courtesy; then exact flat-IP top-10(+1) over the 256 x 512 result for 16
    image-id queries (query-index.py:86-99) and 16 synthetic token rows through the text tower (:107-111)."""
    from PIL import Image
    from oracle import clip_oracle
    n = 256
    blobs = make_jpegs(n)
    tf = clipmi.make_transform(224)

    def run_b1(items):
        rows = []
        t0 = time.perf_counter()
        for b in items:
            x = tf(Image.open(io.BytesIO(b))).unsqueeze(0)
            f = clip_oracle.encode_image(sd, x)
            rows.append((f / f.norm(dim=-1, keepdim=True)).numpy().astype("float32"))
        return time.perf_counter() - t0, rows

    # B = 1 matmuls do not scale to every host core: try torch's default thread count (what the
    # reference would get) and moderate ones, keep the fastest, and say which was used
    default_threads = torch.get_num_threads()
    best_threads, best_rate = default_threads, 0.0
    for nt in sorted({default_threads, min(default_threads, 16), min(default_threads, 32)}):
        torch.set_num_threads(nt)
        run_b1(blobs[:2])
        rate = 6 / run_b1(blobs[:6])[0]
        if rate > best_rate:
            best_threads, best_rate = nt, rate
    torch.set_num_threads(best_threads)
    dt1, rows = run_b1(blobs)
    mat = np.concatenate(rows, axis=0)                                   # [256, 512] = the index
    t0 = time.perf_counter()
    for lo in range(0, n, 32):
        x = torch.stack([tf(Image.open(io.BytesIO(b))) for b in blobs[lo:lo + 32]])
        f = clip_oracle.encode_image(sd, x)
        f = f / f.norm(dim=-1, keepdim=True)
    dt32 = time.perf_counter() - t0
    # query side: K = 10 + 1
    rng = np.random.default_rng(3)
    ids = np.zeros((16, 77), dtype=np.int64)
    for r in range(16):
        eot = int(rng.integers(4, 30))
        ids[r, 0] = 49406
        ids[r, 1:eot] = rng.integers(1, 49406, eot - 1)
        ids[r, eot] = 49407
    t0 = time.perf_counter()
    tq = []
    for r in range(16):
        f = clip_oracle.encode_text(sd, torch.from_numpy(ids[r:r + 1])).numpy().astype("float32")
        nrm = np.linalg.norm(f)
        tq.append(f if nrm < 1e-9 else f / nrm)
    dt_text = time.perf_counter() - t0
    queries = [mat[i:i + 1] for i in range(16)] + tq
    t0 = time.perf_counter()
    for q in queries:                                                    # Q = 1 per call, as the REPL does
        s = (mat @ q.T)[:, 0]
        top = np.lexsort((np.arange(n), -s))[:11]
        _ = s[top]
    dt_search = time.perf_counter() - t0
    torch.set_num_threads(default_threads)
    return {"value": n / dt1, "unit": "images/s", "cores": best_threads, "kind": "port",
            "sample": f"configs[0]: {n} synthetic 224x224 JPEGs (rng 0, quality 95): Pillow decode + transform + oracle "
                      f"fp32 encode at B=1 per image (reference semantics) + normalise, torch {torch.__version__} CPU, "
                      f"{best_threads} threads (fastest of default {default_threads} / 32 / 16), {dt1:.1f} s",
            "images_per_s_b32": n / dt32,
            "search_top11_over_256_rows": {"queries": 32, "queries_per_s": 32 / dt_search,
                                           "text_encode_ms_per_query": dt_text / 16 * 1e3,
                                           "what": "16 image-id + 16 text queries, numpy f32 dot + exact (score desc, id asc) "
                                                   "top-11 per query; text tower through the fp32 oracle"},
            "host_cpus": os.cpu_count()}


def cpu_baseline_search(rows_total, Q, K):
    """faiss IndexFlatIP stand-in on the host cores: numpy f32 `db @ q.T` + top-K, on a bounded
    row sample, scaled linearly to the full row count."""
    n = min(rows_total, 1_000_000)
    rng = np.random.default_rng(1)
    db = rng.standard_normal((n, 512), dtype=np.float32)
    q = rng.standard_normal((Q, 512), dtype=np.float32)

    def once():
        s = db @ q.T
        idx = np.argpartition(-s, K, axis=0)[:K]
        return np.take_along_axis(s, idx, axis=0)
    once()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    dt = (time.perf_counter() - t0) / reps
    scale = rows_total / n
    return {"value": Q / (dt * scale), "unit": "queries/s", "cores": os.cpu_count(), "kind": "port",
            "sample": f"numpy f32 matmul + argpartition top-{K} over {n} x 512 rows, Q={Q}, time scaled x{scale:g} "
                      f"to {rows_total} rows"}


def unit_rows_device(n, dev, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    db = torch.empty((n, 512), dtype=torch.float32, device=dev)
    chunk = 1 << 20
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        blk = torch.randn((e - s, 512), generator=g, device=dev)
        db[s:e] = blk / blk.norm(dim=1, keepdim=True)
    return db


def scan_probe(L, idx, q, Qp, K, dev, kind):
    """The dominant search kernel alone: HIP events recorded by the library around it on the launch stream.
    -> (ms, survivors per query or None, algorithmic bytes, kernel name, traffic key)"""
    db = idx.matrix()
    n_local = db.shape[0]
    os_ = torch.empty((64, K), dtype=torch.float32, device=dev)
    oi_ = torch.empty((64, K), dtype=torch.int64, device=dev)
    scan_ms = C.c_float(0)
    survivors = C.c_longlong(-1)
    qg = 1 if Qp <= 16 else 2 if Qp <= 32 else 4
    if kind == "int8":
        db8, meta, amax, rmax = idx.matrix_i8()
        ws = torch.empty(L.clipmi_topk_ip_coarse_workspace_bytes(n_local, 512, Qp, K), dtype=torch.uint8, device=dev)
        clipmi._lib.check(L.clipmi_dbg_topk_coarse_i8_scan_ms(db.data_ptr(), db8.data_ptr(), meta.data_ptr(), amax, n_local, 512,
                                                              rmax, q.data_ptr(), Qp, K, os_.data_ptr(), oi_.data_ptr(),
                                                              ws.data_ptr(), ws.numel(), clipmi._lib.stream_ptr(dev), 10,
                                                              C.byref(scan_ms), C.byref(survivors)), "coarse_i8_scan_ms")
        return (scan_ms.value, survivors.value / Qp, n_local * (512 + 8),     # int8 row + its (scale, error norm) pair
                f"scan_coarse_kernel<512,{qg},*,true> (three launches stream the copy once)", "scan_coarse_i8_bytes_per_pass")
    if kind == "bf16":
        dbh, rmax = idx.matrix_bf16()
        ws = torch.empty(L.clipmi_topk_ip_coarse_workspace_bytes(n_local, 512, Qp, K), dtype=torch.uint8, device=dev)
        clipmi._lib.check(L.clipmi_dbg_topk_coarse_scan_ms(db.data_ptr(), dbh.data_ptr(), n_local, 512, rmax, q.data_ptr(), Qp,
                                                           K, os_.data_ptr(), oi_.data_ptr(), ws.data_ptr(), ws.numel(),
                                                           clipmi._lib.stream_ptr(dev), 10, C.byref(scan_ms),
                                                           C.byref(survivors)), "coarse_scan_ms")
        return (scan_ms.value, survivors.value / Qp, n_local * 512 * 2,
                f"scan_coarse_kernel<512,{qg},false,false>", "scan_coarse_bytes_per_launch")
    ws = torch.empty(L.clipmi_topk_ip_workspace_bytes(n_local, 512, Qp, K), dtype=torch.uint8, device=dev)
    clipmi._lib.check(L.clipmi_dbg_topk_scan_ms(db.data_ptr(), n_local, 512, q.data_ptr(), Qp, K, os_.data_ptr(),
                                                oi_.data_ptr(), ws.data_ptr(), ws.numel(), clipmi._lib.stream_ptr(dev),
                                                10, C.byref(scan_ms)), "scan_ms")
    return (scan_ms.value, None, n_local * 512 * 4, f"scan_topk_f32_kernel<512,false,{2 if Qp > 16 else 1}>",
            "scan_bytes_per_launch")


_SEARCH_STREAMS = {}


def search_leg(L, a, dev, dist, world, rank, rows_total, n_local, seed, steps, warmup, large_q=0):
    """One search measurement: this rank's `n_local` rows of a `rows_total`-row index, Q queries, K = k + 1."""
    K, Q = a.k + 1, a.queries
    db = unit_rows_device(n_local, dev, seed + rank)          # every shard its own rows (no cross-shard duplicates)
    idx = clipmi.IndexFlatIP(512, device=dev, coarse=None if a.exact_only else a.coarse)
    idx.add(db)
    coarse = idx.uses_coarse()
    kind = (a.coarse if coarse else "f32")
    if coarse:                   # the coarse copy + row-norm bounds are part of the index, built once
        idx.matrix_i8() if kind == "int8" else idx.matrix_bf16()
    gq = torch.Generator(device=dev)
    gq.manual_seed(2)
    q = torch.randn((Q, 512), generator=gq, device=dev)
    q = q / q.norm(dim=1, keepdim=True)
    searcher = clipmi.ShardedFlatIP(idx, rows_total) if world > 1 else idx
    res = [None]

    def search_step():
        res[0] = searcher.search_device(q, K)
    sp_one, sp_two = {}, {}
    dt_one = timed(search_step, steps, warmup, dist, world, sp_one)
    assert (res[0][1][:, 0] >= 0).all()
    ref_i = res[0][1].clone()
    # the same K steps with TWO batches in flight on two HIP streams (every call owns its stream's workspace and
    # result buffers): one batch's short latency-bound kernels (pre-pass, re-scoring, selects, merge) run beside the
    # other batch's HBM-bound scan. Results are identical; throughput is what BASELINE.json's queries/s asks for.
    nfl = int(os.environ.get("CLIPMI_BENCH_IN_FLIGHT", "2"))
    # the SAME two streams for every search leg of the process: this ROCm gives a process's first streams their own hardware
    # queue and maps later ones onto one shared queue - round 3's shard leg created its own pair after the
    # first leg's, both landed on one queue and "two in flight" ran back to back (1.45 ms against 1.21 with two queues)
    if (str(dev), nfl) not in _SEARCH_STREAMS:
        _SEARCH_STREAMS[(str(dev), nfl)] = [torch.cuda.Stream(device=dev) for _ in range(nfl)]
    streams = _SEARCH_STREAMS[(str(dev), nfl)]
    turn = [0]

    def search_step2():
        with torch.cuda.stream(streams[turn[0] % nfl]):
            res[0] = searcher.search_device(q, K)
        turn[0] += 1
    dt_two = timed(search_step2, steps, nfl * ((warmup + nfl - 1) // nfl), dist, world, sp_two)
    torch.cuda.synchronize()
    assert torch.equal(res[0][1], ref_i)
    # headline of the leg = ONE declared form, the same in every round and leg: two calls in flight (ADVICE r04 - a best-of-two
    # chosen after measuring biases the figure upward and makes rounds incomparable); the other form is published beside it.
    # CLIPMI_BENCH_IN_FLIGHT=1 makes the two forms the same.
    dt_s = dt_two
    headline_form = "two_batches_in_flight"
    Qp = min(Q, 64 if coarse else 32)            # queries of ONE pass
    scan_ms, surv, scan_bytes, scan_name, traffic_key = scan_probe(L, idx, q, Qp, K, dev, kind)
    scan_gbs = scan_bytes / (scan_ms * 1e-3) / 1e9
    passes = (Q + Qp - 1) // Qp
    traffic, tsrc = pmc_traffic(traffic_key)
    out = {"value": Q * steps / dt_s, "unit": "queries/s", "ms_per_step": dt_s / steps * 1e3, "steps": steps,
           "batches_in_flight": nfl, "headline_form": headline_form,
           "one_batch_in_flight": dict({"value": Q * steps / dt_one, "ms_per_step": dt_one / steps * 1e3}, **sp_one),
           "two_batches_in_flight": dict({"value": Q * steps / dt_two, "ms_per_step": dt_two / steps * 1e3}, **sp_two),
           "dtype": ("int8 coarse scan (i32 MFMA) + f32 exact re-scoring" if kind == "int8" else
                     "bf16 coarse scan + f32 exact re-scoring" if kind == "bf16" else "f32"),
           "path": ("coarse-then-exact (clipmi_topk_ip_coarse_i8)" if kind == "int8" else
                    "coarse-then-exact (clipmi_topk_ip_coarse)" if kind == "bf16" else "exact scan (clipmi_topk_ip)"),
           "queries_per_batch": Q, "K": K, "rows_per_gpu": n_local, "rows_total": rows_total,
           "roofline": {"bound": "hbm", "kernel": scan_name, "achieved": scan_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": scan_gbs / PEAK_HBM_GBS, "traffic": traffic, "traffic_source": tsrc,
                        "kernel_ms": scan_ms, "algorithmic_bytes_per_launch": scan_bytes,
                        "coarse_survivors_per_query": surv,
                        "whole_call_gbs_per_gpu": scan_bytes * passes * steps / dt_s / 1e9,
                        "whole_call_frac": scan_bytes * passes * steps / dt_s / 1e9 / PEAK_HBM_GBS,
                        "whole_call_frac_one_in_flight": scan_bytes * passes * steps / dt_one / 1e9 / PEAK_HBM_GBS}}
    # ONE search call of 1024 queries (query-index.py:111 is one index.search call whatever Q): the int8 path takes it as ONE
    # wide pass of the copy (csrc/topk.hip scan_coarse_wide_kernel: query tiles resident in LDS, integer MFMA). Two rooflines:
    # SURVEY.md 8(d)'s bytes (the copy streamed once per call) against HBM, and 2 N Q E int8 operations against the dense
    # int8 MFMA peak (2 x the bf16 figure: MI355X_MICROARCH.md "Matrix cores"). Reported beside the 64-query number.
    if coarse and large_q > 0:
        gq.manual_seed(3)
        qL = torch.randn((large_q, 512), generator=gq, device=dev)
        qL = qL / qL.norm(dim=1, keepdim=True)

        def big_step():
            res[0] = searcher.search_device(qL, K)
        sp_big = {}
        dt_big = timed(big_step, 5, 2, dist, world, sp_big)
        ms_call = dt_big / 5 * 1e3
        copy_bytes = n_local * (512 + 8) if kind == "int8" else n_local * 1024
        ops = 2.0 * n_local * large_q * 512
        out["one_call_many_queries"] = dict({
            "queries": large_q, "value": large_q * 5 / dt_big, "unit": "queries/s", "ms_per_call": ms_call, "steps": 5,
            "path": "wide pass (clipmi_topk_ip_coarse_i8, Q > 64)" if kind == "int8" else "64-query passes on two streams",
            "roofline_hbm": {"bound": "hbm", "algorithmic_bytes_per_call": copy_bytes,
                             "achieved": copy_bytes / (ms_call * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": copy_bytes / (ms_call * 1e-3) / 1e9 / PEAK_HBM_GBS},
            "roofline_mfma_int8": {"bound": "mfma", "ops_per_call": ops, "achieved": ops / (ms_call * 1e-3) / 1e12,
                                   "peak": 2 * PEAK_BF16_TFLOPS, "unit": "TOP/s",
                                   "frac": ops / (ms_call * 1e-3) / 1e12 / (2 * PEAK_BF16_TFLOPS),
                                   "note": "whole call (sample, scans, exact re-scoring, selects) over the scan's operations"}},
            **sp_big)
        if kind == "int8":
            # 2 x large_q queries in ONE call = two wide chunks, which IndexFlatIP alternates between two streams
            qL2 = torch.cat([qL, qL.flip(0)])

            def big2_step():
                res[0] = searcher.search_device(qL2, K)
            dt2 = timed(big2_step, 3, 1, dist, world)
            out["one_call_many_queries"]["one_call_2x"] = {
                "queries": 2 * large_q, "value": 2 * large_q * 3 / dt2, "unit": "queries/s", "ms_per_call": dt2 / 3 * 1e3, "steps": 3,
                "note": "two wide chunks of one call, alternating between two streams (IndexFlatIP._search_pipelined)"}
    if large_q > 0:
        sweep = {}
        # ONE call in flight each. Beyond 64 queries the int8 path is the wide pass (first form to 191 queries, the second form's
        # balanced tiles from 192: VERDICT r04 item 3 asked for 128 / 256 / 640 beside 1024)
        gq.manual_seed(4)
        qw = torch.randn((640, 512), generator=gq, device=dev)
        qw = qw / qw.norm(dim=1, keepdim=True)
        for Qs in (1, 16) + ((128, 256, 512, 640) if coarse and kind == "int8" else ()):
            qs_ = (q[:Qs] if Qs <= Q else qw[:Qs]).contiguous()

            def sweep_step():
                res[0] = searcher.search_device(qs_, K)
            reps = 10 if Qs <= 64 else 5
            dts = timed(sweep_step, reps, 2, dist, world)
            sweep[str(Qs)] = round(Qs * reps / dts, 1)
        sweep[str(Q)] = round(Q * steps / dt_one, 1)
        if "one_call_many_queries" in out:
            sweep[str(large_q)] = round(out["one_call_many_queries"]["value"], 1)
        sweep = {k: sweep[k] for k in sorted(sweep, key=int)}
        out["sweep_by_queries"] = sweep              # one call in flight each: the latency view (queries/s = Q / call time)
    del searcher, idx, db
    torch.cuda.empty_cache()
    return out


def main():
    a = parse()
    # (CLIPMI_BENCH_FORCE_SPAWN=1 takes the self-spawning path with one rank too: the test of that path on a 1-GPU box)
    if "WORLD_SIZE" not in os.environ and (a.gpus > 1 or os.environ.get("CLIPMI_BENCH_FORCE_SPAWN") == "1"):
        sys.exit(spawn_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        print(json.dumps({"error": f"--gpus {a.gpus} but WORLD_SIZE={world}", "n_gpus": a.gpus, "value": None}), flush=True)
        sys.exit(2)
    # decode workers for the files-to-vectors leg are child programs: start them before this process touches the GPU
    decode_pool = None
    if world == 1 and not a.quick and not a.no_cpu_baseline:
        decode_pool = clipmi.pipeline.DecodePool(clipmi.indexer.default_workers())
    assert torch.cuda.is_available(), "bench.py needs a GPU (the HIP path has no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    L = clipmi._lib.lib()

    # ---------------- encode: synthetic uint8 images resident in HBM, random-init ViT-B/32 -------
    sd = clipmi.weights.random_state_dict("ViT-B/32", seed=0)
    model = clipmi.CLIP(sd, device=dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1234 + rank)
    B = a.batch
    images = torch.randint(0, 256, (B, 3, 224, 224), generator=g, device=dev, dtype=torch.uint8)
    enc_out = [None]

    def enc_step():
        enc_out[0] = model.encode_image(images, normalize=True)
    # Two forms of the same step (round 5), as the search leg has: ONE kernel sequence of B images on one stream (rounds 1-4's
    # step; the form every per-kernel figure of `roofline` is measured in), and the product's default since round 5 - the step cut
    # into one-round chunks that alternate between the caller's stream and one internal stream (CLIP.image_lanes: 870 images =
    # two sequences of 435 in flight). The headline is ONE declared form: --encode-in-flight (default 2); both are published.
    forms = {}
    for fl in ((1, 2) if a.encode_in_flight == 2 else (1,)):      # (--encode-in-flight 1: the profiled runs hold ONE form only)
        model.chunks_in_flight = fl
        sp_ = {}
        dt_ = timed(enc_step, a.steps, a.warmup, dist, world, sp_)
        forms[fl] = (dt_, sp_)
    model.chunks_in_flight = a.encode_in_flight
    dt_enc, sp_enc = forms[a.encode_in_flight]
    img_per_s = world * B * a.steps / dt_enc
    assert torch.isfinite(enc_out[0]).all()
    enc_forms = {("one_sequence" if fl == 1 else "two_sequences_in_flight"):
                 dict({"value": world * B * a.steps / forms[fl][0], "ms_per_step": forms[fl][0] / a.steps * 1e3,
                       "whole_step_frac": FLOP_PER_IMAGE * B * a.steps / forms[fl][0] / 1e12 / PEAK_BF16_TFLOPS,
                       "sequences": [hi - lo for lo, hi, _ in (model.image_lanes(B) if fl == 2 else [(0, B, 0)])]}, **forms[fl][1])
                 for fl in sorted(forms)}

    # configs[1] reads "encode 1M images": the same step repeated until >= 1 M images per GPU have gone through
    # (about 11 s of sustained bf16 MFMA load: the clock the chip holds under it is part of the answer)
    sustained = None
    if a.sustained_images > 0:
        steps_s = (a.sustained_images + B - 1) // B
        dt_sus = timed(enc_step, steps_s, 1, dist, world)
        sustained = {"value": world * B * steps_s / dt_sus, "unit": "images/s", "images_per_gpu": B * steps_s,
                     "steps": steps_s, "seconds": dt_sus, "ms_per_step": dt_sus / steps_s * 1e3,
                     "whole_step_frac": FLOP_PER_IMAGE * B * steps_s / dt_sus / 1e12 / PEAK_BF16_TFLOPS}

    # BASELINE.json configs[4] beside the headline (never `value`: reduced precision): the same step with the
    # block linear layers on the FP8 matrix cores (e4m3 weights, activations quantised per row on the fly)
    fp8_info = None
    if not a.no_fp8:
        ref16 = enc_out[0].clone()
        model8 = clipmi.CLIP(sd, device=dev, vision_weights="fp8")
        model8.chunks_in_flight = a.encode_in_flight

        def enc8_step():
            enc_out[0] = model8.encode_image(images, normalize=True)
        steps8 = max(3, a.steps // 2)
        sp8 = {}
        dt8 = timed(enc8_step, steps8, 2, dist, world, sp8)
        cos = torch.nn.functional.cosine_similarity(enc_out[0].double(), ref16.double(), dim=-1)
        fp8_info = {"metric": "images/sec ViT-B/32 encode, FP8 (e4m3) linear layers", "value": world * B * steps8 / dt8,
                    "unit": "images/s", "ms_per_step": dt8 / steps8 * 1e3, "steps": steps8, "dtype": "fp8 e4m3 x e4m3 -> f32",
                    "min_cosine_to_bf16_path": float(cos.min()), **sp8,
                    "ratio_to_bf16_headline": world * B * steps8 / dt8 / img_per_s,
                    "note": "BASELINE.json configs[4] parity case; not the headline (configs[1] is bf16). e4m3 activations carry MX "
                            "block scales written by the producing kernels (DESIGN.md 4.4c)"}
        del model8

    # SURVEY.md 8(d) cfg-2 sweep: images/s at B in {64, 128, 256, 512} (compact: 10 synchronised steps each), beside the
    # headline's B - the number a caller gets without knowing which batch sizes fill whole rounds of GEMM tiles
    enc_sweep = None
    if not a.quick:
        enc_sweep = {}
        for Bs in (64, 128, 256, 512):
            imgs_s = images[:Bs] if Bs <= B else torch.randint(0, 256, (Bs, 3, 224, 224), generator=g, device=dev, dtype=torch.uint8)

            def sweep_step():
                enc_out[0] = model.encode_image(imgs_s, normalize=True)
            dts = timed(sweep_step, 10, 2, dist, world)
            enc_sweep[str(Bs)] = round(world * Bs * 10 / dts, 1)

    # ONE prompt through the text tower (the REPL's shape: query-index.py:107-108, Q = 1): latency per query, host ids as the
    # tokenizer hands them over (EOT at position 9: the tower runs on 10 positions, skinny GEMM kernels), beside the CPU
    # baseline's text_encode_ms_per_query
    text_one = None
    if not a.quick:
        ids1 = torch.zeros(1, 77, dtype=torch.int64)
        ids1[0, 0] = 49406; ids1[0, 1:9] = torch.randint(1, 40000, (8,)); ids1[0, 9] = 49407
        ids77 = ids1.clone(); ids77[0, 9] = 5; ids77[0, 76] = 49407

        def text_step():
            enc_out[0] = model.encode_text(ids1, normalize=True)

        def text_step77():
            enc_out[0] = model.encode_text(ids77, normalize=True)
        dt_t = timed(text_step, 100, 5, dist, world)
        dt_t77 = timed(text_step77, 100, 5, dist, world)
        text_one = {"ms_per_query": dt_t / 100 * 1e3, "ms_per_query_full_77_tokens": dt_t77 / 100 * 1e3, "queries": 1,
                    "what": "encode_text + normalise of ONE prompt, back-to-back calls on one stream (10 tokens incl. SOT / EOT; "
                            "and a prompt that fills all 77 positions)"}

    # The encode step's GEMMs timed IN SITU: HIP events recorded by the library around its launches on the launch stream.
    # The DOMINANT family by time is the residual producer gemm256p<7, false> (out_proj K = 768 + c_proj K = 3072: 24 launches,
    # ~41 % of the step): `roofline` reports it; the c_fc GEMM (28 % of the step, the kernel `roofline` named up to round 3)
    # is beside it as `roofline.c_fc`.
    M, N, Kd = B * 50, 3072, 768
    need = L.clipmi_encode_image_workspace_bytes(model.vision, B)
    ews = torch.empty(need, dtype=torch.uint8, device=dev)
    eout = torch.empty((B, 512), dtype=torch.float32, device=dev)

    def probe(epi):
        ms3, nl, kkind, kepi = (C.c_float * 3)(), C.c_int(0), C.c_int(-1), C.c_int(-1)
        clipmi._lib.check(L.clipmi_dbg_encode_image_probe3_ms(model.vision, model._vblob.data_ptr(), images.data_ptr(),
                                                              clipmi._lib.U8, B, eout.data_ptr(), ews.data_ptr(), ews.numel(),
                                                              clipmi._lib.stream_ptr(dev), epi, 3, ms3, C.byref(nl), C.byref(kkind),
                                                              C.byref(kepi)), "encode_image_probe3")
        sym = (["gemm_bf16_nt_kernel<%d>", "gemm256_bf16_nt_kernel<%d>", "gemm256p_bf16_nt_kernel<%d, false>"][kkind.value]
               % kepi.value) if 0 <= kkind.value <= 2 else "?"
        return [float(v) for v in ms3], nl.value, sym
    # The kernel's in-situ duration: completion of the kernel directly in front -> completion of this one. For c_fc that is
    # ms3[2] (the GEMM in front is the residual producer of the same block, no kernel between them); for the residual
    # producers ms3[1] (an event recorded in front of the launch: out_proj follows attention, c_proj follows c_fc) - the
    # kernel + one kernel boundary, the estimator that agrees with rocprofv3's dispatch duration (profiles/). The launch's
    # own begin -> end events (ms3[0]) read ~14 % long in a back-to-back stream: the begin stamp is taken when the packet is
    # processed, before the previous kernel has drained. All estimators are in the line.
    ms_fc, nl_fc, sym_fc = probe(1)
    fc_ms = ms_fc[2] if ms_fc[2] > 0 else ms_fc[1] if ms_fc[1] > 0 else ms_fc[0]
    fc_tflops = 2.0 * M * N * Kd / (fc_ms * 1e-3) / 1e12
    rs_flop = (2.0 * M * 768 * 768 + 2.0 * M * 768 * 3072) / 2.0           # per launch, averaged over the pair
    rs_parts = None
    try:
        # completion of the kernel directly in front -> completion of this one (ms3[2], the estimator that agrees with
        # rocprofv3): c_proj (K = 3072) follows c_fc, out_proj (K = 768) follows attention, whose launch carries its own end
        # stamp in this mode. Average of the pair = the family.
        ms_cp, nl_cp, sym_rs = probe(2 | (3072 << 8))
        ms_op, nl_op, _ = probe(2 | (768 << 8))
        cp_ms = ms_cp[2] if ms_cp[2] > 0 else ms_cp[1]
        op_ms = ms_op[2] if ms_op[2] > 0 else ms_op[1]
        rs_ms, nl_rs = (cp_ms + op_ms) / 2.0, nl_cp + nl_op
        ms_rs = [(ms_cp[0] + ms_op[0]) / 2.0, (ms_cp[1] + ms_op[1]) / 2.0, 0.0]
        rs_parts = {"c_proj_K3072": {"kernel_ms": cp_ms, "tflops": 2.0 * M * 768 * 3072 / (cp_ms * 1e-3) / 1e12,
                                     "estimator": "completion_to_completion", "completion_to_completion": ms_cp[2],
                                     "event_in_front_to_end": ms_cp[1], "launch_begin_to_end_events": ms_cp[0]},
                    "out_proj_K768": {"kernel_ms": op_ms, "tflops": 2.0 * M * 768 * 768 / (op_ms * 1e-3) / 1e12,
                                      "estimator": "completion_to_completion" if ms_op[2] > 0 else "event_in_front_to_end",
                                      "completion_to_completion": ms_op[2], "event_in_front_to_end": ms_op[1],
                                      "launch_begin_to_end_events": ms_op[0]}}
    except clipmi.ClipmiError:
        # small batches (fewer tiles than CUs): the residual producer runs as GEMM-into-scratch + split_stats, which this
        # probe does not bracket - report the c_fc GEMM in its place (not the headline configuration)
        ms_rs, nl_rs, sym_rs, rs_ms, rs_flop = ms_fc, nl_fc, sym_fc + " [residual producer not on the persistent kernel at this batch: c_fc reported]", fc_ms, 2.0 * M * N * Kd
    rs_tflops = rs_flop / (rs_ms * 1e-3) / 1e12
    del ews, eout, model, images
    torch.cuda.empty_cache()

    # ---------------- configs[3]: ViT-L/14@336px (24 layers, width 1024, 577 tokens, 768-D) -----------------------
    l14 = None
    if a.l14_batch > 0:
        sdl = clipmi.weights.random_state_dict("ViT-L/14@336px", seed=0)
        ml = clipmi.CLIP(sdl, device=dev)
        del sdl
        Bl = a.l14_batch
        gi = torch.Generator(device=dev)
        gi.manual_seed(77 + rank)
        imgs_l = torch.randint(0, 256, (Bl, 3, 336, 336), generator=gi, device=dev, dtype=torch.uint8)
        lo_ = [None]

        def l14_step():
            lo_[0] = ml.encode_image(imgs_l, normalize=True)
        steps_l = max(3, a.steps // 4)
        sp_l = {}
        dt_l = timed(l14_step, steps_l, 1, dist, world, sp_l)
        assert torch.isfinite(lo_[0]).all()
        tf_l = FLOP_PER_IMAGE_L14_336 * Bl * steps_l / dt_l / 1e12
        l14 = {"metric": "images/sec ViT-L/14@336px encode (BASELINE.json configs[3])", "value": world * Bl * steps_l / dt_l,
               "unit": "images/s", "ms_per_step": dt_l / steps_l * 1e3, "steps": steps_l, "images_per_gpu_per_step": Bl,
               "dtype": "bf16", "flop_per_image": FLOP_PER_IMAGE_L14_336, **sp_l,
               "roofline": {"bound": "mfma", "kernel": "whole step (24 layers; GEMMs gemm256p/gemm256, flash attention)",
                            "achieved": tf_l, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": tf_l / PEAK_BF16_TFLOPS,
                            "traffic": None}}
        del ml, imgs_l
        torch.cuda.empty_cache()

    # ---------------- search: 10M x 512 f32 split over the ranks, Q queries, K = k + 1 -----------
    lo, hi = clipmi.shard_bounds(a.rows, world, rank)
    search = search_leg(L, a, dev, dist, world, rank, a.rows, hi - lo, 1000, a.steps, a.warmup, large_q=a.large_q)
    search["metric"] = f"queries/sec top-{a.k} over {a.rows}x512 flat IP (exact results, f32 scores)"
    search["scaling"] = "strong"
    # configs[4] search half: 12.5 M rows on every rank (weak): at N = 8 a 100 M x 512 DB
    shard = None
    if a.shard_rows > 0:
        shard = search_leg(L, a, dev, dist, world, rank, a.shard_rows * world, a.shard_rows, 5000, a.steps, a.warmup)
        shard["metric"] = (f"queries/sec top-{a.k} over {a.shard_rows * world}x512 flat IP, {a.shard_rows} rows per GPU "
                           f"(BASELINE.json configs[4] search half)")
        shard["scaling"] = "weak"

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    traffic, tsrc = pmc_traffic("gemm_resid_bytes_per_launch")
    traffic_fc, tsrc_fc = pmc_traffic("gemm_c_fc_bytes_per_launch")
    K, Q = a.k + 1, a.queries
    out = {
        "metric": "images/sec ViT-B/32 encode", "value": img_per_s, "unit": "images/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt_enc / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic", **sp_enc,
        "headline_form": "two_sequences_in_flight" if a.encode_in_flight == 2 else "one_sequence", "encode_forms": enc_forms,
        "config": {"workload": f"BASELINE.json configs[1]: ViT-B/32 bf16 encode of synthetic 224x224 uint8 images "
                               f"(random-init weights, L=50, 12 layers), {B} images per GPU per step, fused normalise, "
                               f"inputs resident in HBM; then exact-result flat-IP top-{K} (k={a.k}+1, "
                               f"query-index.py:111) over {a.rows} x 512 f32 split over {world} GPU(s), Q={Q} per batch",
                   "images_per_gpu_per_step": B, "index_rows_total": a.rows, "queries_per_batch": Q, "K": K},
        "roofline": {"bound": "mfma",
                     "kernel": sym_rs + f" (residual producers: attn.out_proj K=768 and mlp.c_proj K=3072, M={M} N=768, + bias + "
                                        f"split-residual update + LayerNorm statistics; the dominant kernel by time: 24 launches per step)",
                     "achieved": rs_tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": rs_tflops / PEAK_BF16_TFLOPS, "traffic": traffic, "traffic_source": tsrc,
                     "kernel_ms": rs_ms, "launches_timed": nl_rs,
                     "algorithmic_flop_per_launch": rs_flop,
                     "kernel_ms_estimators": {"event_in_front_to_end": ms_rs[1], "launch_begin_to_end_events": ms_rs[0],
                                              "used": "mean of c_proj and out_proj, each completion of the kernel in front -> own completion"},
                     "per_shape": rs_parts,
                     "c_fc": {"kernel": sym_fc + f" (MLP c_fc + LayerNorm fold + bias + QuickGELU, M={M} N={N} K={Kd}; 12 launches per step)",
                              "achieved": fc_tflops, "frac": fc_tflops / PEAK_BF16_TFLOPS, "kernel_ms": fc_ms,
                              "launches_timed": nl_fc, "traffic": traffic_fc, "traffic_source": tsrc_fc,
                              "kernel_ms_estimators": {"completion_to_completion": ms_fc[2], "event_in_front_to_end": ms_fc[1],
                                                       "launch_begin_to_end_events": ms_fc[0]}},
                     "kernel_figures_form": "one_sequence: every per-kernel figure above is measured inside ONE kernel sequence of "
                                            f"{B} images on one stream (M = {M}: the library's own probe), the form the rocprofv3 summaries "
                                            "under profiles/ are taken in (bench.py --encode-in-flight 1); with two sequences in flight two "
                                            "launches of half the rows share the chip and a launch's own duration says nothing",
                     "whole_step_tflops_per_gpu": FLOP_PER_IMAGE * B * a.steps / dt_enc / 1e12,
                     "whole_step_frac": FLOP_PER_IMAGE * B * a.steps / dt_enc / 1e12 / PEAK_BF16_TFLOPS,
                     "whole_step_frac_one_sequence": enc_forms["one_sequence"]["whole_step_frac"]},
        "search": search,
    }
    if text_one is not None:
        out["encode_text_one_prompt"] = text_one
    if enc_sweep is not None:
        out["sweeps"] = {"encode_images_per_s_by_batch": enc_sweep,
                         "search_queries_per_s_by_queries": search.pop("sweep_by_queries", None)}
    if sustained is not None:
        out["encode_sustained"] = sustained
    if fp8_info is not None:
        out["encode_fp8"] = fp8_info
    if l14 is not None:
        out["encode_vitl14_336"] = l14
    if shard is not None:
        out["search_shard_12p5m"] = shard
    if decode_pool is not None:
        out["files_to_vectors"] = files_to_vectors_leg(clipmi.CLIP(sd, device=dev), decode_pool)
        decode_pool.close()
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_cfg1(sd)
        out["search"]["cpu_baseline"] = cpu_baseline_search(a.rows, Q, K)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
