"""CPU tests of the host side: shard arithmetic, index file round trip, and the N>1 search path
over gloo with world_size 2 (local scan supplied by the oracle; the collective + merge logic is
what is under test)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, unit_rows


def test_image_chunk_fills_whole_tile_rounds(clipmi):
    """Inputs larger than max_batch are encoded in chunks whose token rows fill whole rounds of 256 x 256 tiles on 256 CUs
    for the narrowest GEMM (model.py image_chunk): 870 images for ViT-B/32, never more than max_batch."""
    for (W, Lv), want in {(768, 50): 870, (768, 197): 998, (1024, 577): 1022}.items():
        m = clipmi.CLIP.__new__(clipmi.CLIP)
        m.dims, m.max_batch = {"v_width": W, "v_tokens": Lv}, 1024
        c = m.image_chunk()
        assert c == want and c <= m.max_batch
        tiles = (c * Lv + 255) // 256 * (W // 256)                    # 256 x 256 output tiles of the N = W GEMMs
        assert tiles / ((tiles + 255) // 256 * 256) > 0.99           # the last round of 256 CUs is full too
    m.dims, m.max_batch, m.round_chunks = {"v_width": 768, "v_tokens": 50}, 1024, True
    assert m.image_chunks(512) == [512] and m.image_chunks(1024) == [1024]          # up to max_batch: one sequence
    assert m.image_chunks(1740) == [870, 870] and m.image_chunks(1305) == [870, 435] and m.image_chunks(2000) == [870, 870, 260]
    m.max_batch = 10                                   # smaller than one round of tiles: max_batch-sized chunks
    assert m.image_chunk() == 0 and m.image_chunks(25) == [10, 10, 5]


def test_shard_bounds_cover_exactly(clipmi):
    for n in (0, 1, 7, 8, 10_000_000):
        for w in (1, 2, 3, 8):
            b = [clipmi.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_index_file_roundtrip(clipmi, tmp_path):
    rng = np.random.default_rng(0)
    x = unit_rows(rng, 33, 512)
    idx = clipmi.IndexFlatIP(512, device="cpu")
    idx.add(x[:10]); idx.add(x[10:])
    assert idx.ntotal == 33 and idx.is_trained
    idx.train(x)           # no-op, accepted
    idx.nprobe = 32        # accepted, ignored
    p = str(tmp_path / "images.index")
    clipmi.write_index(idx, p)
    back = clipmi.read_index(p, device="cpu")
    assert back.ntotal == 33 and np.array_equal(back.matrix().numpy(), x)
    with open(p, "r+b") as f:
        f.truncate(100)
    with pytest.raises(ValueError):
        clipmi.read_index(p, device="cpu")


def test_faiss_flat_ip_file_layout(clipmi, tmp_path):
    """next-3: byte layout of the faiss IndexFlatIP serialisation (restated from upstream, unpinned)."""
    import struct
    rng = np.random.default_rng(1)
    x = unit_rows(rng, 7, 512)
    idx = clipmi.IndexFlatIP(512, device="cpu")
    idx.add(x)
    p = str(tmp_path / "images.index")
    clipmi.write_index(idx, p, format="faiss")
    raw = open(p, "rb").read()
    assert raw[:4] == b"IxFI" and len(raw) == 4 + 4 + 8 + 8 + 8 + 1 + 4 + 8 + 7 * 512 * 4
    d, n, d1, d2 = struct.unpack("<iqqq", raw[4:32])
    assert (d, n, d1, d2) == (512, 7, 1 << 20, 1 << 20)
    assert struct.unpack("<Bi", raw[32:37]) == (1, 0) and struct.unpack("<Q", raw[37:45]) == (7 * 512,)
    assert raw[45:] == x.astype("<f4").tobytes()
    back = clipmi.read_index(p, device="cpu")
    assert back.ntotal == 7 and np.array_equal(back.matrix().numpy(), x)
    open(p, "wb").write(b"IwXX" + raw[4:])
    with pytest.raises(ValueError, match="not a clipmi"):
        clipmi.read_index(p, device="cpu")


def _write_faiss_ivfflat(path, x, nlist, sparse, rng):
    """An IndexIVFFlat file as faiss impl/index_write.cpp lays it out (what the reference's build-index.py:
    80-81,109 produces): restated from upstream by the test, independently of the reader."""
    import struct
    n, d = x.shape
    cent = unit_rows(rng, nlist, d)
    assign = np.argmax(x @ cent.T, axis=1) if not sparse else rng.integers(0, 3, n)     # sparse: 3 of nlist lists used
    hdr = lambda d_, n_: struct.pack("<iqqq", d_, n_, 1 << 20, 1 << 20) + struct.pack("<Bi", 1, 0)
    with open(path, "wb") as f:
        f.write(b"IwFl" + hdr(d, n) + struct.pack("<QQ", nlist, 1))
        f.write(b"IxFI" + hdr(d, nlist) + struct.pack("<Q", nlist * d) + cent.astype("<f4").tobytes())
        f.write(struct.pack("<B", 0) + struct.pack("<Q", 0))                             # direct map: none
        f.write(b"ilar" + struct.pack("<QQ", nlist, 4 * d))
        sizes = np.bincount(assign, minlength=nlist).astype("<u8")
        if sparse:
            nz = np.nonzero(sizes)[0]
            pairs = np.stack([nz.astype("<u8"), sizes[nz]], axis=1).reshape(-1)
            f.write(b"sprs" + struct.pack("<Q", pairs.size) + pairs.astype("<u8").tobytes())
        else:
            f.write(b"full" + struct.pack("<Q", nlist) + sizes.tobytes())
        for l in range(nlist):
            ids = np.nonzero(assign == l)[0].astype("<i8")
            if ids.size:
                f.write(x[ids].astype("<f4").tobytes() + ids.tobytes())


@pytest.mark.parametrize("sparse", [False, True])
def test_reads_reference_ivfflat_index_file(clipmi, tmp_path, sparse):
    """next-3: an `images.index` written by the reference (IndexIVFFlat, 100 lists, inner product) opens and
    its rows come back in id order (format restated from upstream; PARITY UNPINNED: no faiss here)."""
    rng = np.random.default_rng(7)
    x = unit_rows(rng, 333, 512)
    p = str(tmp_path / "images.index")
    _write_faiss_ivfflat(p, x, 100, sparse, rng)
    idx = clipmi.read_index(p, device="cpu")
    idx.nprobe = 32                                   # query-index.py:30
    assert idx.ntotal == 333 and np.array_equal(idx.matrix().numpy(), x)
    raw = bytearray(open(p, "rb").read())
    open(p, "wb").write(raw[:-100])
    with pytest.raises(ValueError):
        clipmi.read_index(p, device="cpu")


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import clipmi
    from conftest import TopkOracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(99)          # same data on every rank
        N, Q, K = 5003, 5, 51
        db = unit_rows(rng, N, 512)
        db[N - 1] = db[3]                          # duplicate across shards
        q = unit_rows(rng, Q, 512)
        orc = TopkOracle()
        lo, hi = clipmi.shard_bounds(N, world, rank)

        def local_search(qq, k, base):
            assert base == lo
            return orc.topk(db[lo:hi], np.asarray(qq), k, id_base=lo)

        sh = clipmi.ShardedFlatIP(None, N, local_search=local_search)
        D, I = sh.search(q, K)
        Dw, Iw = orc.topk(db, q, K)
        ok = np.array_equal(I, Iw) and np.array_equal(D.view(np.uint32), Dw.view(np.uint32))
        open(os.path.join(tmp, f"ok{rank}"), "w").write("1" if ok else "0")
    finally:
        dist.destroy_process_group()


def test_sharded_search_gloo_world2(tmp_path):
    import torch.multiprocessing as mp
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert (tmp_path / "ok0").read_text() == "1" and (tmp_path / "ok1").read_text() == "1"


def test_read_index_row_ranges(clipmi, tmp_path):
    """One rank's shard straight from the file (seek, no full load): both flat formats, id_base = lo."""
    rng = np.random.default_rng(3)
    x = unit_rows(rng, 101, 512)
    idx = clipmi.IndexFlatIP(512, device="cpu")
    idx.add(x)
    for fmt in ("clipmi", "faiss"):
        p = str(tmp_path / f"i.{fmt}")
        clipmi.write_index(idx, p, format=fmt)
        assert clipmi.index.index_rows(p) == (101, 512)
        for r in range(3):
            lo, hi = clipmi.shard_bounds(101, 3, r)
            part = clipmi.read_index(p, device="cpu", rows=(lo, hi))
            assert part.id_base == lo and part.n_file == 101 and np.array_equal(part.matrix().numpy(), x[lo:hi])
        with pytest.raises(ValueError):
            clipmi.read_index(p, device="cpu", rows=(50, 102))


class _FakeModel:
    """CPU stand-in for the encoder in tests of the build-side HOST logic (file sharding, store commits):
    features are a fixed function of the decoded pixels, so they do not depend on batch composition or rank."""

    class _V:
        input_resolution = 32

    def __init__(self):
        self.device = torch.device("cpu")
        self.visual = self._V()
        self.embed_dim = 512

    def encode_image(self, x, normalize=False):
        f = x.float().reshape(x.shape[0], -1)[:, :512] + 1.0
        return f / f.norm(dim=-1, keepdim=True) if normalize else f


def _make_photo_dir(d, n):
    from PIL import Image
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(17)
    for i in range(n):
        Image.fromarray(rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)).save(os.path.join(d, f"img_{i:05d}.png"))
    open(os.path.join(d, "broken.jpg"), "wb").write(b"not a jpeg")
    open(os.path.join(d, "notes.txt"), "w").write("ignored")


def _indexer_worker(rank, world, port, tmp, interrupt_rank=-1):
    sys.path.insert(0, ROOT)
    import clipmi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.chdir(tmp)
    ranks = clipmi.ranks.Ranks("cpu", rank=rank, world=world, local=rank).init()
    store = os.path.join(tmp, "sharded.store")
    db = clipmi.store.VectorStore(store, dim=512, backend="packed") if ranks.leader else None
    model = _FakeModel()
    if rank == interrupt_rank:
        # Ctrl-C reaches THIS rank in the middle of its second batch: SIGINT to itself from inside the encoder
        import signal
        calls = [0]
        enc = model.encode_image

        def encode_image(x, normalize=False):
            calls[0] += 1
            if calls[0] == 2:
                os.kill(os.getpid(), signal.SIGINT)
            return enc(x, normalize=normalize)
        model.encode_image = encode_image
    with clipmi.indexer.StopFlag() as stop:
        stopped = clipmi.indexer.encode_directories([os.path.join(tmp, "lib") + "/"], model, db, 4, 2, ranks, stop=stop,
                                                    store_path=store)
    assert stopped == (interrupt_rank >= 0)
    if ranks.leader:
        clipmi.indexer.finalise(db, "cpu", out=os.path.join(tmp, "sharded.index"))
        db.close()
        open(os.path.join(tmp, "leader.done"), "w").write("stopped" if stopped else "complete")
    ranks.close()


def test_sharded_build_gloo_world2(clipmi, tmp_path, capsys):
    """SURVEY.md §8e build side: the sorted file list is split contiguously over 2 ranks, each encodes its slice,
    rank 0 ingests every round's results. The store and index must equal the single-process build's, byte for byte."""
    import torch.multiprocessing as mp
    tmp = str(tmp_path)
    _make_photo_dir(os.path.join(tmp, "lib"), 23)
    db1 = clipmi.store.VectorStore(os.path.join(tmp, "single.store"), dim=512, backend="packed")
    clipmi.indexer.encode_directories([os.path.join(tmp, "lib") + "/"], _FakeModel(), db1, 4, 2)
    clipmi.indexer.finalise(db1, "cpu", out=os.path.join(tmp, "single.index"))
    mp.spawn(_indexer_worker, args=(2, 29500 + (os.getpid() + 7) % 2000, tmp), nprocs=2, join=True)
    db2 = clipmi.store.VectorStore(os.path.join(tmp, "sharded.store"), dim=512, backend="packed")
    assert db1.count() == db2.count() == 23
    for t in ("fn_db", "skip_db", "idx_db"):
        assert list(db1.b.items_sorted(t)) == list(db2.b.items_sorted(t)), t
    assert db2.is_skipped(os.path.join(tmp, "lib") + "/broken.jpg")
    assert open(os.path.join(tmp, "single.index"), "rb").read() == open(os.path.join(tmp, "sharded.index"), "rb").read()
    # a second sharded run finds nothing left to do (resume semantics, build-index.py:36-44)
    assert clipmi.indexer.candidates(os.path.join(tmp, "lib") + "/", db2) == []
    db1.close(); db2.close()
    # every rank wrote its own shard store and rank 0 ingested + removed them
    assert not [f for f in os.listdir(tmp) if ".shard-" in f]


def test_sharded_build_ctrl_c_on_one_rank_gloo_world2(clipmi, tmp_path):
    """ADVICE r02 (medium) / build-index.py:63-64: Ctrl-C that reaches ONE rank (rank 1, inside its second batch) must not
    hang the others in a collective: the ranks agree on the round to stop in, rank 0 ingests what the shards hold and still
    finalises. A second, uninterrupted run completes the library; the result equals the single-process build."""
    import torch.multiprocessing as mp
    tmp = str(tmp_path)
    _make_photo_dir(os.path.join(tmp, "lib"), 23)
    mp.spawn(_indexer_worker, args=(2, 29500 + (os.getpid() + 19) % 2000, tmp, 1), nprocs=2, join=True)
    assert open(os.path.join(tmp, "leader.done")).read() == "stopped"
    assert not [f for f in os.listdir(tmp) if ".shard-" in f]
    db = clipmi.store.VectorStore(os.path.join(tmp, "sharded.store"), dim=512, backend="packed")
    n_first = db.count()
    assert 0 < n_first < 23 and os.path.exists(os.path.join(tmp, "sharded.index"))
    assert clipmi.index.index_rows(os.path.join(tmp, "sharded.index")) == (n_first, 512)
    db.close()
    mp.spawn(_indexer_worker, args=(2, 29500 + (os.getpid() + 23) % 2000, tmp), nprocs=2, join=True)
    assert open(os.path.join(tmp, "leader.done")).read() == "complete"
    db1 = clipmi.store.VectorStore(os.path.join(tmp, "single.store"), dim=512, backend="packed")
    clipmi.indexer.encode_directories([os.path.join(tmp, "lib") + "/"], _FakeModel(), db1, 4, 2)
    clipmi.indexer.finalise(db1, "cpu", out=os.path.join(tmp, "single.index"))
    db2 = clipmi.store.VectorStore(os.path.join(tmp, "sharded.store"), dim=512, backend="packed")
    for t in ("fn_db", "skip_db", "idx_db"):
        assert list(db1.b.items_sorted(t)) == list(db2.b.items_sorted(t)), t
    assert open(os.path.join(tmp, "single.index"), "rb").read() == open(os.path.join(tmp, "sharded.index"), "rb").read()
    db1.close(); db2.close()


def test_leftover_shard_of_a_dead_run_is_ingested(clipmi, tmp_path):
    """A run that died between its last batch and rank 0's ingest leaves `<store>.shard-<rank>` behind: finished work. The
    next start ingests it before listing what is left to do."""
    tmp = str(tmp_path)
    store = os.path.join(tmp, "v.store")
    sh = clipmi.store.VectorStore(clipmi.indexer.shard_path(store, 3), dim=512, backend="packed")
    rng = np.random.default_rng(5)
    vec = unit_rows(rng, 2, 512)
    sh.put_vectors(["lib/a.png", "lib/b.png"], vec)
    sh.mark_skipped(["lib/bad.jpg"])
    sh.close()
    db = clipmi.store.VectorStore(store, dim=512, backend="packed")
    assert clipmi.indexer.ingest_shards(db, store) == 2
    assert np.array_equal(db.get_vector("lib/b.png")[0], vec[1]) and db.is_skipped("lib/bad.jpg") and db.count() == 2
    assert not os.path.exists(clipmi.indexer.shard_path(store, 3))
    db.close()


def _repl_worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import clipmi
    from conftest import TopkOracle
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    ranks = clipmi.ranks.Ranks("cpu", rank=rank, world=world, local=rank).init()
    path = os.path.join(tmp, "images.index")
    n, d = clipmi.index.index_rows(path)
    lo, hi = clipmi.shard_bounds(n, world, rank)
    rows = clipmi.read_index(path, device="cpu", rows=(lo, hi)).matrix().numpy()
    orc = TopkOracle()
    sharded = clipmi.repl.open_sharded(path, ranks, local_search=lambda q, k, base: orc.topk(rows, np.asarray(q), k, id_base=base))
    if not ranks.leader:
        clipmi.repl.follow(sharded, ranks)
    else:
        db = clipmi.store.VectorStore(os.path.join(tmp, "vectors.store"), dim=512, backend="packed")
        index = clipmi.repl.LeaderIndex(sharded, ranks)
        script = iter(["c 7", "i 5", "", "p 40", "i 999999", "i 2001", "q"])
        lines = []
        clipmi.repl.repl(None, index, db, inp=lambda p: next(script), out=lines.append)
        index.quit()
        db.close()
        open(os.path.join(tmp, "repl.out"), "w").write("\n".join(lines))
    ranks.close()


def test_sharded_query_repl_gloo_world2(clipmi, tmp_path, topk_oracle):
    """SURVEY.md §8e query side: rank 0 runs the prompt loop, both ranks search their row shard of images.index,
    one all-gather + merge; printed results equal the single-process exact answer."""
    import torch.multiprocessing as mp
    tmp = str(tmp_path)
    rng = np.random.default_rng(31)
    N = 3001
    x = unit_rows(rng, N, 512)
    x[2999] = x[5]                                           # a duplicate in the other shard
    keys = [f"/lib/img_{i:05d}.jpg" for i in range(N)]
    db = clipmi.store.VectorStore(os.path.join(tmp, "vectors.store"), dim=512, backend="packed")
    db.put_vectors(keys, x)
    db.assemble()
    db.close()
    idx = clipmi.IndexFlatIP(512, device="cpu")
    idx.add(x)
    clipmi.write_index(idx, os.path.join(tmp, "images.index"))
    mp.spawn(_repl_worker, args=(2, 29500 + (os.getpid() + 13) % 2000, tmp), nprocs=2, join=True)
    lines = open(os.path.join(tmp, "repl.out")).read().splitlines()
    res = [l for l in lines if l.count(" ") == 2 and l.split()[1].isdigit() and "/lib/img_" in l]
    # "c 7" -> K = 7 + 0 + 1 per query; the empty line pages only after a TEXT query (query-index.py:100-103)
    D, I = topk_oracle.topk(x, x[5:6], 8)
    D2, I2 = topk_oracle.topk(x, x[2001:2002], 8)
    want = [(f"{D[0][j]:.4f}", int(I[0][j])) for j in range(1, 8)] + [(f"{D2[0][j]:.4f}", int(I2[0][j])) for j in range(1, 8)]
    got = [(l.split()[0], int(l.split()[1])) for l in res]
    assert got == want
    assert I[0][0] == 5 and I[0][1] == 2999                   # the cross-shard duplicate ranks right behind the query row
    assert "Not found." in lines and "Set to probe 40 subsets." in lines
    assert all(l.split()[2] == f"/lib/img_{int(l.split()[1]):05d}.jpg" for l in res)


def test_bench_refuses_missing_gpus_and_stale_traffic(tmp_path):
    """bench.py on a box without the GPUs it is asked for: a JSON error record and a non-zero exit (no traceback, no GPU
    touched); and the roofline `traffic` figure is only taken from a profile summary whose recorded library digest equals the
    library in the tree."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--quick"], capture_output=True, text=True,
                       timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 2, r.stderr[-2000:]
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "error" in rec and rec["n_gpus"] == 64 and rec["value"] is None
    sys.path.insert(0, ROOT)
    import bench
    import clipmi
    val, src = bench.pmc_traffic("gemm_c_fc_bytes_per_launch")
    newest = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_traffic.json"))[-1]
    rec = json.load(open(os.path.join(ROOT, "profiles", newest)))
    if rec.get("lib_digest") == clipmi.build.source_digest():
        assert val == rec["gemm_c_fc_bytes_per_launch"] and newest in src
    else:
        assert val is None and "stale" in src
