"""CPU tests of the host side: shard arithmetic, index file round trip, and the N>1 search path
over gloo with world_size 2 (local scan supplied by the oracle; the collective + merge logic is
what is under test)."""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, unit_rows


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
    open(p, "wb").write(b"IwFl" + raw[4:])
    with pytest.raises(ValueError, match="rebuilt"):
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
