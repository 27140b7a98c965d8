#!/usr/bin/env python3
"""bench.py — the hot path of ps-auxw/CLI-P on MI355X: ViT-B/32 encode_image throughput and exact
flat inner-product top-50 search, on synthetic data, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch: encode a batch of B synthetic 224x224 uint8
images on every rank (weak: per-GPU work fixed), then — timed in its own bracket — one search
batch of Q queries for the best K = 50 + 1 (query-index.py:111: k + offset + 1) over the 10M x 512
f32 matrix split contiguously over the ranks (strong: total rows fixed), with ONE all-gather of the
per-rank partial lists and the K-way merge. Rank 0 prints ONE JSON line.

`value` is images/s (BASELINE.json's first metric); the second metric (queries/s) and its own
roofline and CPU baseline are in the "search" object of the same line.
"""
import argparse
import ctypes as C
import io
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import clipmi  # noqa: E402

FLOP_PER_IMAGE = 8_817_623_040          # ViT-B/32, L = 50, 2 FLOP per MAC
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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp8", action="store_true", help="skip the FP8 (configs[4]) encode measurement")
    return ap.parse_args()


def bracket(dist, world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def timed(fn, steps, warmup, dist, world):
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
    return dt


def pmc_traffic(key):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/*_pmc_traffic.json,
    written by tools/pmc_traffic.py from FETCH_SIZE/WRITE_SIZE with the gfx950 corrections)."""
    best = None
    pdir = os.path.join(ROOT, "profiles")
    if os.path.isdir(pdir):
        for f in sorted(os.listdir(pdir)):
            if f.endswith("_pmc_traffic.json"):
                try:
                    d = json.load(open(os.path.join(pdir, f)))
                    if key in d:
                        best = d[key]
                except Exception:
                    pass
    return best


def cpu_baseline_encode(sd):
    """Reference semantics on the host cores: JPEG decode + transform + fp32
    encode at B = 1 per image (build-index.py:47-50) through oracle/clip_oracle.py ("port")."""
    from PIL import Image
    from oracle import clip_oracle
    rng = np.random.default_rng(0)
    n = 48
    blobs = []
    for _ in range(n):
        buf = io.BytesIO()
        Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)).save(buf, format="JPEG", quality=95)
        blobs.append(buf.getvalue())
    tf = clipmi.make_transform(224)

    def run(items):
        t0 = time.perf_counter()
        for b in items:
            x = tf(Image.open(io.BytesIO(b))).unsqueeze(0)
            f = clip_oracle.encode_image(sd, x)
            f = f / f.norm(dim=-1, keepdim=True)
        return time.perf_counter() - t0

    # B = 1 matmuls do not scale to every host core: try torch's default thread count (what the
    # reference would get) and a moderate one, keep the faster, and say which was used
    default_threads = torch.get_num_threads()
    best_threads, best_rate = default_threads, 0.0
    for nt in sorted({default_threads, min(default_threads, 16), min(default_threads, 32)}):
        torch.set_num_threads(nt)
        run(blobs[:2])
        rate = 6 / run(blobs[:6])
        if rate > best_rate:
            best_threads, best_rate = nt, rate
    torch.set_num_threads(best_threads)
    dt = run(blobs)
    torch.set_num_threads(default_threads)
    return {"value": n / dt, "unit": "images/s", "cores": best_threads, "kind": "port",
            "sample": f"{n} synthetic 224x224 JPEGs (quality 95): Pillow decode + transform + oracle fp32 "
                      f"encode at B=1 per image (reference semantics), torch {torch.__version__} CPU, "
                      f"{best_threads} threads (fastest of default {default_threads} / 32 / 16)"}


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


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
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
    dt_enc = timed(enc_step, a.steps, a.warmup, dist, world)
    img_per_s = world * B * a.steps / dt_enc
    assert torch.isfinite(enc_out[0]).all()

    # BASELINE.json configs[4] beside the headline (never `value`: reduced precision): the same step with the
    # block linear layers on the FP8 matrix cores (e4m3 weights, activations quantised per row on the fly)
    fp8_info = None
    if not a.no_fp8:
        ref16 = enc_out[0].clone()
        model8 = clipmi.CLIP(sd, device=dev, vision_weights="fp8")

        def enc8_step():
            enc_out[0] = model8.encode_image(images, normalize=True)
        steps8 = max(3, a.steps // 2)
        dt8 = timed(enc8_step, steps8, 2, dist, world)
        cos = torch.nn.functional.cosine_similarity(enc_out[0].double(), ref16.double(), dim=-1)
        fp8_info = {"metric": "images/sec ViT-B/32 encode, FP8 (e4m3) linear layers", "value": world * B * steps8 / dt8,
                    "unit": "images/s", "ms_per_step": dt8 / steps8 * 1e3, "steps": steps8, "dtype": "fp8 e4m3 x e4m3 -> f32",
                    "min_cosine_to_bf16_path": float(cos.min()),
                    "note": "BASELINE.json configs[4] parity case; not the headline (configs[1] is bf16)"}
        del model8

    # dominant encode kernel (MLP c_fc GEMM + bias + QuickGELU, 12 launches per step), timed IN SITU:
    # HIP events recorded by the library around each of its launches on the launch stream
    M, N, Kd = B * 50, 3072, 768
    need = L.clipmi_encode_image_workspace_bytes(model.vision, B)
    ews = torch.empty(need, dtype=torch.uint8, device=dev)
    eout = torch.empty((B, 512), dtype=torch.float32, device=dev)
    kms, nl = C.c_float(0), C.c_int(0)
    clipmi._lib.check(L.clipmi_dbg_encode_image_probe_ms(model.vision, model._vblob.data_ptr(), images.data_ptr(),
                                                         clipmi._lib.U8, B, eout.data_ptr(), ews.data_ptr(), ews.numel(),
                                                         clipmi._lib.stream_ptr(dev), 1, 3, C.byref(kms), C.byref(nl)),
                      "encode_image_probe")
    gemm_ms = kms.value
    gemm_tflops = 2.0 * M * N * Kd / (gemm_ms * 1e-3) / 1e12
    del ews, eout

    # ---------------- search: 10M x 512 f32 split over the ranks, Q queries, K = k + 1 -----------
    K = a.k + 1
    Q = a.queries
    lo, hi = clipmi.shard_bounds(a.rows, world, rank)
    n_local = hi - lo
    gd = torch.Generator(device=dev)
    gd.manual_seed(1000 + rank)  # every shard its own rows (no cross-shard duplicates)
    db = torch.empty((n_local, 512), dtype=torch.float32, device=dev)
    chunk = 1 << 20
    for s in range(0, n_local, chunk):
        e = min(n_local, s + chunk)
        blk = torch.randn((e - s, 512), generator=gd, device=dev)
        db[s:e] = blk / blk.norm(dim=1, keepdim=True)
    idx = clipmi.IndexFlatIP(512, device=dev, coarse=None if a.exact_only else a.coarse)
    idx.add(db)
    coarse = idx.uses_coarse()
    i8 = coarse and a.coarse == "int8"
    if coarse:                   # the coarse copy + row-norm bounds are part of the index, built once
        idx.matrix_i8() if i8 else idx.matrix_bf16()
    gq = torch.Generator(device=dev)
    gq.manual_seed(2)
    q = torch.randn((Q, 512), generator=gq, device=dev)
    q = q / q.norm(dim=1, keepdim=True)
    searcher = clipmi.ShardedFlatIP(idx, a.rows) if world > 1 else idx
    res = [None]

    def search_step():
        res[0] = searcher.search_device(q, K)
    dt_s = timed(search_step, a.steps, a.warmup, dist, world)
    qps = Q * a.steps / dt_s
    assert (res[0][1][:, 0] >= 0).all()

    # dominant search kernel alone, HIP events recorded by the library around it on the launch stream
    Qp = min(Q, 64 if coarse else 32)            # queries of ONE pass
    os_ = torch.empty((64, K), dtype=torch.float32, device=dev)
    oi_ = torch.empty((64, K), dtype=torch.int64, device=dev)
    scan_ms = C.c_float(0)
    survivors = C.c_longlong(-1)
    if i8:
        db8, meta, amax, rmax = idx.matrix_i8()
        ws = torch.empty(L.clipmi_topk_ip_coarse_workspace_bytes(n_local, 512, Qp, K), dtype=torch.uint8, device=dev)
        clipmi._lib.check(L.clipmi_dbg_topk_coarse_i8_scan_ms(db.data_ptr(), db8.data_ptr(), meta.data_ptr(), amax, n_local, 512,
                                                              rmax, q.data_ptr(), Qp, K, os_.data_ptr(), oi_.data_ptr(),
                                                              ws.data_ptr(), ws.numel(), clipmi._lib.stream_ptr(dev), 10,
                                                              C.byref(scan_ms), C.byref(survivors)), "coarse_i8_scan_ms")
        scan_bytes = n_local * (512 + 8)         # int8 row + its (scale, error norm) pair
        scan_name = f"scan_coarse_kernel<512,{1 if Qp <= 16 else 2 if Qp <= 32 else 4},false,true>"
        traffic_key = "scan_coarse_i8_bytes_per_launch"
    elif coarse:
        dbh, rmax = idx.matrix_bf16()
        ws = torch.empty(L.clipmi_topk_ip_coarse_workspace_bytes(n_local, 512, Qp, K), dtype=torch.uint8, device=dev)
        clipmi._lib.check(L.clipmi_dbg_topk_coarse_scan_ms(db.data_ptr(), dbh.data_ptr(), n_local, 512, rmax, q.data_ptr(), Qp,
                                                           K, os_.data_ptr(), oi_.data_ptr(), ws.data_ptr(), ws.numel(),
                                                           clipmi._lib.stream_ptr(dev), 10, C.byref(scan_ms),
                                                           C.byref(survivors)), "coarse_scan_ms")
        scan_bytes = n_local * 512 * 2
        scan_name = f"scan_coarse_kernel<512,{1 if Qp <= 16 else 2 if Qp <= 32 else 4},false,false>"
        traffic_key = "scan_coarse_bytes_per_launch"
    else:
        ws = torch.empty(L.clipmi_topk_ip_workspace_bytes(n_local, 512, Qp, K), dtype=torch.uint8, device=dev)
        clipmi._lib.check(L.clipmi_dbg_topk_scan_ms(db.data_ptr(), n_local, 512, q.data_ptr(), Qp, K, os_.data_ptr(),
                                                    oi_.data_ptr(), ws.data_ptr(), ws.numel(), clipmi._lib.stream_ptr(dev),
                                                    10, C.byref(scan_ms)), "scan_ms")
        scan_bytes = n_local * 512 * 4
        scan_name = f"scan_topk_f32_kernel<512,false,{2 if Qp > 16 else 1}>"
        traffic_key = "scan_bytes_per_launch"
    scan_gbs = scan_bytes / (scan_ms.value * 1e-3) / 1e9

    if rank != 0:
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return

    out = {
        "metric": "images/sec ViT-B/32 encode", "value": img_per_s, "unit": "images/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt_enc / a.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": f"ViT-B/32 bf16 encode of synthetic 224x224 uint8 images (random-init weights, "
                               f"L=50, 12 layers), {B} images per GPU per step, fused normalise, inputs resident "
                               f"in HBM; then exact-result flat-IP top-{K} (k={a.k}+1, query-index.py:111) over "
                               f"{a.rows} x 512 f32 split over {world} GPU(s), Q={Q} per batch",
                   "images_per_gpu_per_step": B, "index_rows_total": a.rows, "queries_per_batch": Q, "K": K},
        "roofline": {"bound": "mfma", "kernel": ("gemm256p_bf16_nt_kernel<1>" if (N // 256) * ((M + 255) // 256) > 256
                                                 else "gemm256_bf16_nt_kernel<1>") +
                                                f" (MLP c_fc + bias + QuickGELU, M={M} N={N} K={Kd})",
                     "achieved": gemm_tflops, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": gemm_tflops / PEAK_BF16_TFLOPS, "traffic": pmc_traffic("gemm_c_fc_bytes_per_launch"),
                     "kernel_ms": gemm_ms, "launches_timed": nl.value,
                     "whole_step_tflops_per_gpu": FLOP_PER_IMAGE * B * a.steps / dt_enc / 1e12,
                     "whole_step_frac": FLOP_PER_IMAGE * B * a.steps / dt_enc / 1e12 / PEAK_BF16_TFLOPS},
        "search": {"metric": f"queries/sec top-{a.k} over {a.rows}x512 flat IP (exact results, f32 scores)", "value": qps,
                   "unit": "queries/s", "ms_per_step": dt_s / a.steps * 1e3, "scaling": "strong",
                   "dtype": ("int8 coarse scan (i32 MFMA) + f32 exact re-scoring" if i8 else
                             "bf16 coarse scan + f32 exact re-scoring" if coarse else "f32"),
                   "path": ("coarse-then-exact (clipmi_topk_ip_coarse_i8)" if i8 else
                            "coarse-then-exact (clipmi_topk_ip_coarse)" if coarse else "exact scan (clipmi_topk_ip)"),
                   "queries_per_batch": Q, "rows_per_gpu": n_local,
                   "roofline": {"bound": "hbm", "kernel": scan_name,
                                "achieved": scan_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                "frac": scan_gbs / PEAK_HBM_GBS, "traffic": pmc_traffic(traffic_key),
                                "kernel_ms": scan_ms.value, "algorithmic_bytes_per_launch": scan_bytes,
                                "coarse_survivors_per_query": (survivors.value / Qp) if survivors.value >= 0 else None,
                                "whole_call_gbs_per_gpu": scan_bytes * ((Q + Qp - 1) // Qp) * a.steps / dt_s / 1e9}},
    }
    if fp8_info is not None:
        out["encode_fp8"] = fp8_info
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline_encode(sd)
        out["search"]["cpu_baseline"] = cpu_baseline_search(a.rows, Q, K)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
