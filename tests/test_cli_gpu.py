"""-m gpu end-to-end checks of the drop-in surface: the build/query scripts on a small synthetic
photo directory, and the N>1 search path (RCCL process group, clipmi_merge_topk) with one rank."""
import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, unit_rows

pytestmark = pytest.mark.gpu


def _load_script(name):
    spec = importlib.util.spec_from_file_location(name.replace("-", "_"), os.path.join(ROOT, name))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_build_then_query_scripts(clipmi, gpu, tmp_path, monkeypatch, topk_oracle, capsys):
    from PIL import Image
    d = tmp_path / "lib"
    d.mkdir()
    rng = np.random.default_rng(0)
    for i in range(37):
        Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)).save(str(d / f"img_{i:05d}.jpg"), quality=95)
    (d / "broken.png").write_bytes(b"nope")
    monkeypatch.chdir(tmp_path)
    monkeypatch.setenv("CLIPMI_RANDOM_WEIGHTS", "0")
    monkeypatch.setenv("CLIPMI_BATCH", "16")
    monkeypatch.delenv("CLIPMI_WEIGHTS", raising=False)
    bi = _load_script("build-index.py")
    bi.main([str(d) + "/"])
    out = capsys.readouterr().out
    assert out.count(".") >= 37 and "#" in out and "Preparing index for 37 entries..." in out and out.rstrip().endswith("Done!")
    # vectors in the store are the normalised HIP embeddings of the decoded pixels
    db = clipmi.store.VectorStore("vectors.lmdb", dim=512)
    model, transform = clipmi.load("ViT-B/32", device=gpu, seed=0)
    key = str(d) + "/img_00003.jpg"
    ref = model.encode_image(transform(Image.open(key)).unsqueeze(0), normalize=True).cpu().numpy()
    got = db.get_vector(key)
    assert np.abs(got - ref).max() < 2e-3 and abs(np.linalg.norm(got) - 1) < 1e-5
    # query side: image-similarity query through the REPL, checked against the oracle over the stored rows
    qi = _load_script("query-index.py")
    index = clipmi.read_index("images.index", device=gpu)
    mat, _ = db.assemble()
    script = iter(["c 10", "i 5", "q"])
    lines = []
    qi.repl(model, index, db, inp=lambda p: next(script), out=lambda s: lines.append(s))
    res = [l for l in lines if l.count(" ") == 2 and l.split()[1].isdigit() and "/img_" in l]
    D, I = topk_oracle.topk(mat, mat[5:6], 11)
    assert I[0][0] == 5 and [int(l.split()[1]) for l in res] == list(I[0][1:])
    db.close()


def _nccl_worker(tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import clipmi
    from conftest import TopkOracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(29600 + os.getpid() % 1000)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        rng = np.random.default_rng(5)
        db = unit_rows(rng, 20000, 512)
        q = unit_rows(rng, 3, 512)
        idx = clipmi.IndexFlatIP(512, device="cuda:0")
        idx.add(db)
        sh = clipmi.ShardedFlatIP(idx, 20000)
        D, I = sh.search(q, 51)
        Ds, Is = TopkOracle().topk(db, q, 51)
        ok = np.array_equal(I, Is) and np.array_equal(D.view(np.uint32), Ds.view(np.uint32))
        # a caller holding two results keeps both (search_device used to hand out one reused buffer)
        qa, qb = torch.from_numpy(q).cuda(), torch.from_numpy(q[::-1].copy()).cuda()
        ra = sh.search_device(qa, 51)
        rb = sh.search_device(qb, 51)
        torch.cuda.synchronize()
        ok = ok and ra[1].data_ptr() != rb[1].data_ptr() and np.array_equal(ra[1].cpu().numpy(), Is) and \
            np.array_equal(rb[1].cpu().numpy(), Is[::-1])
        open(os.path.join(tmp, "ok"), "w").write("1" if ok else "0")
    finally:
        dist.destroy_process_group()


def test_sharded_search_rccl_single_rank(tmp_path):
    """The RCCL all-gather + device merge path with world_size 1 (one GPU per box here; world 2 runs
    on gloo in tests/test_host_logic.py). Own process: a process group outlives nothing else."""
    code = f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); " \
           f"import test_cli_gpu as t; t._nccl_worker({str(tmp_path)!r})"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert (tmp_path / "ok").read_text() == "1"


def _nccl_worker2(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import clipmi
    from conftest import TopkOracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ranks = clipmi.ranks.Ranks("cuda", rank=rank, world=world, local=rank).init()
    try:
        rng = np.random.default_rng(6)                    # the same rows on every rank; each keeps its shard
        N = 200_003
        db = unit_rows(rng, N, 512)
        db[N - 1] = db[7]                                  # duplicate across the shard boundary
        q = unit_rows(rng, 70, 512)
        lo, hi = clipmi.shard_bounds(N, world, rank)
        idx = clipmi.IndexFlatIP(512, device=ranks.device, coarse="int8")
        idx.add(db[lo:hi])
        sh = clipmi.ShardedFlatIP(idx, N, group=ranks.data)
        D, I = sh.search(q, 51)
        if rank == 0:
            Ds, Is = TopkOracle().topk(db, q, 51)
            ok = np.array_equal(I, Is) and np.array_equal(D.view(np.uint32), Ds.view(np.uint32))
            open(os.path.join(tmp, "ok2"), "w").write("1" if ok else "0")
    finally:
        ranks.close()


def test_sharded_search_rccl_two_ranks(tmp_path):
    """Two ranks on two GPUs: per-shard exact top-K, ONE RCCL all-gather over xGMI, the merge kernel — bit-exact against the
    oracle over the whole matrix. Needs >= 2 GPUs: skipped (and reported as skipped) on the one-GPU boxes."""
    if torch.cuda.device_count() < 2:
        pytest.skip("needs >= 2 GPUs (the N > 1 host logic runs on gloo in tests/test_host_logic.py)")
    import torch.multiprocessing as mp
    mp.spawn(_nccl_worker2, args=(2, 29500 + (os.getpid() + 29) % 2000, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok2").read_text() == "1"


BENCH_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "search")


def _check_bench_line(stdout):
    import json
    line = [l for l in stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    for k in BENCH_KEYS:
        assert k in d
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["value"] > 0 and d["search"]["value"] > 0
    assert d["roofline"]["bound"] == "mfma" and d["search"]["roofline"]["bound"] == "hbm"
    return d


def test_bench_contract_under_torchrun_one_rank(tmp_path):
    """bench.py launched the way the driver launches N>1 (torch.distributed.run), with one rank and a
    small index: one JSON line with the contract's keys; the small configs[3]/configs[4] legs are in the line."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--rows", "200000", "--batch", "64", "--no-cpu-baseline", "--no-fp8", "--sustained-images", "256",
           "--shard-rows", "150000", "--l14-batch", "0"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _check_bench_line(r.stdout)
    assert d["encode_sustained"]["images_per_gpu"] >= 256 and d["search_shard_12p5m"]["rows_per_gpu"] == 150000
    assert d["search_shard_12p5m"]["value"] > 0 and d["search_shard_12p5m"]["roofline"]["bound"] == "hbm"


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no launcher in the environment must start the ranks itself (a child
    torch.distributed.run, before the parent touches the GPU) and relay rank 0's line. One GPU per box here, so the
    path is taken with N = 1 through CLIPMI_BENCH_FORCE_SPAWN=1."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["CLIPMI_BENCH_FORCE_SPAWN"] = "1"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--rows", "200000",
           "--batch", "64", "--quick"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    _check_bench_line(r.stdout)
    # asking for more GPUs than the box has is a JSON error record and a non-zero exit, not a traceback
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--quick"], capture_output=True,
                       text=True, timeout=120, cwd=ROOT, env=env)
    assert r.returncode != 0 and '"error"' in r.stdout


def _resize_pipeline_worker(tmp):
    """Own process, the PRODUCT's start order (indexer.main, bench.py): the decode workers are started BEFORE this process
    touches the GPU; only then the model is built."""
    sys.path.insert(0, ROOT)
    import clipmi
    from PIL import Image
    rng = np.random.default_rng(21)
    specs = [(1024, 768, "RGB", "jpg"), (600, 900, "RGB", "jpg"), (224, 224, "RGB", "jpg"), (224, 400, "RGB", "png"),
             (90, 70, "RGB", "png"), (640, 480, "L", "png"), (500, 300, "RGBA", "png"), (1600, 1200, "RGB", "jpg"),
             (333, 777, "RGB", "jpg"), (3000, 2000, "RGB", "jpg")]
    paths = []
    for i, (w, h, mode, ext) in enumerate(specs):
        ch = {"RGB": 3, "L": 1, "RGBA": 4}[mode]
        a = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        p = os.path.join(tmp, f"f{i:02d}.{ext}" if i != 1 else "f01 new\nline\tand tab.jpg")    # odd file names
        Image.fromarray(a[:, :, 0] if ch == 1 else a, mode).save(p)
        paths.append(p)
    bad = os.path.join(tmp, "f99.jpg")
    with open(bad, "wb") as f:
        f.write(b"broken")
    files = paths[:4] + [bad] + paths[4:]
    with clipmi.pipeline.DecodePool(3) as pool:                     # before anything initialises the GPU
        assert not torch.cuda.is_initialized()
        model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device="cuda:0")
        host = list(clipmi.pipeline.encode_files(model, files, batch=4, pool=pool, device_resize_mb=0, device_jpeg_kb=0))
        devr = list(clipmi.pipeline.encode_files(model, files, batch=4, pool=pool, device_resize_mb=8, device_jpeg_kb=0))   # 3000x2000 stays on the host
        (_, _, _, full), _, _ = pool.decode(files[:4], 224, copy=False, full_cap=8 << 20)
        assert sorted(full) == [0, 1, 3] or sorted(full) == [0, 1]      # photo-sized RGB files travel at full size
    assert [h[0] for h in host] == [d[0] for d in devr] and [h[2] for h in host] == [d[2] for d in devr]
    assert sum(len(h[0]) for h in host) == len(paths) and host[1][2] == [bad]
    for h, d in zip(host, devr):
        assert np.array_equal(h[1], d[1])
    open(os.path.join(tmp, "ok"), "w").write("1")


def _jpeg_pipeline_worker(tmp):
    """Own process, the product's start order. Baseline JPEG files of every sampling, grey, optimised tables, already-sized,
    tall and photo-sized, beside files the device decoder must leave to Pillow (progressive, PNG, CMYK, a file over the size
    cap), a file whose entropy-coded data ends early (the device reports it, Pillow decides) and a broken one."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import clipmi
    from PIL import Image
    from test_jpeg import smooth
    rng = np.random.default_rng(31)
    paths = []

    def put(name, img, **kw):
        p = os.path.join(tmp, name)
        img.save(p, **kw)
        paths.append(p)
        return p

    for i, (w, h, sub, q) in enumerate([(224, 224, 2, 95), (640, 480, 2, 85), (300, 500, 1, 90), (224, 300, 0, 75), (1600, 1200, 2, 80),
                                        (90, 70, 2, 60), (225, 223, 1, 92)]):
        a = smooth(rng, h, w) if i % 2 else rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        put(f"a{i:02d}.jpg", Image.fromarray(a), quality=q, subsampling=sub)
    put("b_grey.jpg", Image.fromarray(smooth(rng, 300, 260)[..., 0]), quality=85, optimize=True)
    put("b_rst.jpg", Image.fromarray(smooth(rng, 360, 480)), quality=88, restart_marker_rows=1)
    put("c_prog.jpg", Image.fromarray(smooth(rng, 400, 300)), quality=85, progressive=True)
    put("d.png", Image.fromarray(smooth(rng, 250, 350)))
    put("e_cmyk.jpg", Image.fromarray(smooth(rng, 240, 320)).convert("CMYK"), quality=85)
    put("f_big.jpg", Image.fromarray(rng.integers(0, 256, (900, 1200, 3), dtype=np.uint8)), quality=95)     # > 256 KB: Pillow's
    cut = put("g_cut.jpg", Image.fromarray(smooth(rng, 320, 320)), quality=90)
    blob = open(cut, "rb").read()
    open(cut, "wb").write(blob[:len(blob) * 2 // 3] + b"\xff\xd9")            # parses, but the data ends early
    bad = os.path.join(tmp, "h_broken.jpg")
    with open(bad, "wb") as f:
        f.write(b"broken")
    files = paths[:5] + [bad] + paths[5:]
    import warnings
    warnings.simplefilter("ignore")
    with clipmi.pipeline.DecodePool(3) as pool:                     # before anything initialises the GPU
        assert not torch.cuda.is_initialized()
        model = clipmi.CLIP(clipmi.weights.random_state_dict("ViT-B/32", seed=0), device="cuda:0")
        host = list(clipmi.pipeline.encode_files(model, files, batch=5, pool=pool, device_resize_mb=0, device_jpeg_kb=0))
        devj = list(clipmi.pipeline.encode_files(model, files, batch=5, pool=pool, device_resize_mb=0, device_jpeg_kb=256))
        both = list(clipmi.pipeline.encode_files(model, files, batch=5, pool=pool, device_resize_mb=8, device_jpeg_kb=256))
        grps = list(clipmi.pipeline.encode_files(model, files, batch=5, pool=pool, device_resize_mb=0, device_jpeg_kb=256, jpeg_group_mb=1))   # a decode launch per file or two
        (_, _, _, full), _, _ = pool.decode(files[:5], 224, copy=False, full_cap=256 << 10, full_mode=2)
        assert {0, 2, 3} <= set(full) and all(v[0] == 3 for v in full.values())      # (files over the size cap stay with Pillow)
    for other in (devj, both, grps):
        assert [h[0] for h in host] == [d[0] for d in other] and [h[2] for h in host] == [d[2] for d in other]
        for h, d in zip(host, other):
            assert (h[1] is None and d[1] is None) or np.array_equal(h[1], d[1])
    failed = [p for h in host for p in h[2]]
    assert bad in failed and len(failed) in (1, 2)                  # the cut file: whatever Pillow decides, both paths agree
    open(os.path.join(tmp, "ok"), "w").write("1")


def test_pipeline_jpeg_decode_on_device_gives_the_same_vectors(tmp_path):
    """encode_files with the JPEG decode on the device (clipmi_jpeg_decode_rgb8 + clipmi_resize_crop_rgb8) returns the
    vectors of the all-Pillow path bit for bit, alone and beside the full-size-RGB path, with the same failed files."""
    code = f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); " \
           f"import test_cli_gpu as t; t._jpeg_pipeline_worker({str(tmp_path)!r})"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert (tmp_path / "ok").read_text() == "1"


def test_pipeline_resize_on_device_gives_the_same_vectors(tmp_path):
    """encode_files with the resize on the device (full-size RGB images -> clipmi_resize_crop_rgb8) returns the vectors of
    the all-host path bit for bit - the device computes the same pixels - for photo-sized, tall, small, already-sized,
    grey, RGBA and broken files in one batch sequence, incl. a file whose NAME holds a newline and a tab. Runs in its own
    process with the product's start order (decode workers first, GPU second)."""
    code = f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); " \
           f"import test_cli_gpu as t; t._resize_pipeline_worker({str(tmp_path)!r})"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert (tmp_path / "ok").read_text() == "1"
