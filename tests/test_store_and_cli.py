"""CPU tests of the storage schema, the host transform and the CLI glue (no GPU: the model and the
search are stubbed; what is under test is keys, ordering, resume semantics, REPL arithmetic)."""
import importlib.util

import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, unit_rows


def _load_script(name):
    spec = importlib.util.spec_from_file_location(name.replace("-", "_"), os.path.join(ROOT, name))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_packed_store_schema_and_resume(clipmi, tmp_path):
    p = str(tmp_path / "vectors.lmdb")
    db = clipmi.store.VectorStore(p, dim=512, backend="packed")
    rng = np.random.default_rng(0)
    v = unit_rows(rng, 5, 512)
    keys = ["/d/b.jpg", "/d/a.jpg", "/d/Z.png", "/d/c.jpeg", "/d/\xe9.jpg"]
    db.put_vectors(keys[:3], v[:3])
    db.put_vectors(keys[3:], v[3:])
    db.mark_skipped(["/d/bad.jpg"])
    assert db.count() == 5 and db.has_vector("/d/a.jpg") and not db.has_vector("/d/bad.jpg")
    assert db.is_skipped("/d/bad.jpg") and not db.is_skipped("/d/a.jpg")
    # value bytes are exactly what build-index.py:51 stores
    assert db.b.get("fn_db", b"/d/a.jpg") == v[1].astype("float32").tobytes()
    got = db.get_vector("/d/a.jpg")
    assert got.shape == (1, 512) and np.array_equal(got[0], v[1])
    db.close()
    # torn tail from an interrupted write is ignored; everything committed before it survives
    with open(os.path.join(p, "fn_db.log"), "ab") as f:
        f.write(b"\x05\x00\x00\x00\x00\x08\x00\x00abc")
    db = clipmi.store.VectorStore(p, dim=512, backend="packed")
    assert db.count() == 5
    mat, paths = db.assemble()
    order = sorted(k.encode() for k in keys)             # LMDB key order = bytewise
    assert paths == order
    for i, k in enumerate(order):
        assert np.array_equal(mat[i], v[keys.index(k.decode())])
        assert db.idx_get(i) == k
    assert db.idx_get(99) is None
    db.close()


def test_packed_store_refuses_real_lmdb_dir(clipmi, tmp_path):
    d = tmp_path / "vectors.lmdb"
    d.mkdir()
    (d / "data.mdb").write_bytes(b"x")
    with pytest.raises(RuntimeError, match="lmdb"):
        clipmi.store.VectorStore(str(d), backend="packed")


def test_transform_matches_clip_preprocessing(clipmi):
    from PIL import Image
    rng = np.random.default_rng(1)
    tf = clipmi.make_transform(224)
    a = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)
    t = tf(Image.fromarray(a))
    mean = np.asarray(clipmi.model.CLIP_MEAN, np.float32).reshape(3, 1, 1)
    std = np.asarray(clipmi.model.CLIP_STD, np.float32).reshape(3, 1, 1)
    assert t.shape == (3, 224, 224) and t.dtype == torch.float32
    assert np.array_equal(t.numpy(), (a.transpose(2, 0, 1).astype(np.float32) / 255.0 - mean) / std)
    # shorter side -> 224 with Pillow bicubic, then centre crop; grey and RGBA inputs become RGB
    big = Image.fromarray(rng.integers(0, 256, (300, 500, 3), dtype=np.uint8))
    ref = big.resize((int(224 * 500 / 300), 224), Image.BICUBIC)
    left = int(round((ref.size[0] - 224) / 2.0))
    ref = np.asarray(ref.crop((left, 0, left + 224, 224)), np.float32).transpose(2, 0, 1) / 255.0
    assert np.array_equal(tf(big).numpy(), (ref - mean) / std)
    assert tf(Image.fromarray(a[:, :, 0])).shape == (3, 224, 224)


def test_pipeline_uint8_equals_transform_pixels(clipmi, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(2)
    p = str(tmp_path / "x.png")
    Image.fromarray(rng.integers(0, 256, (240, 260, 3), dtype=np.uint8)).save(p)
    u8 = clipmi.pipeline.load_uint8(p, 224)
    mean = np.asarray(clipmi.model.CLIP_MEAN, np.float32).reshape(3, 1, 1)
    std = np.asarray(clipmi.model.CLIP_STD, np.float32).reshape(3, 1, 1)
    assert np.array_equal(clipmi.make_transform(224)(Image.open(p)).numpy(), (u8.astype(np.float32) / 255.0 - mean) / std)


def test_decode_pool_matches_in_process_decode(clipmi, tmp_path):
    """Worker processes (pipeline.DecodePool -> decode_worker.py) return the pixels load_uint8 returns, in file order,
    report a file that does not decode instead of failing (build-index.py:55-58), and feed encode_files."""
    from PIL import Image
    rng = np.random.default_rng(5)
    paths = []
    for i, (h, w, ext) in enumerate([(224, 224, "jpg"), (300, 500, "png"), (640, 480, "jpg"), (224, 224, "png"), (100, 90, "jpg")]):
        p = str(tmp_path / f"img_{i}.{ext}")
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(p)
        paths.append(p)
    # a file NAME with a newline and a tab (legal on Linux): the request protocol must not split on it
    odd = str(tmp_path / "img 5 new\nline\ttab.png")
    Image.fromarray(rng.integers(0, 256, (64, 80, 3), dtype=np.uint8)).save(odd)
    paths.append(odd)
    bad = str(tmp_path / "broken.jpg")
    with open(bad, "wb") as f:
        f.write(b"not an image")
    mixed = paths[:2] + [bad] + paths[2:] + [str(tmp_path / "missing.jpg")]
    with clipmi.pipeline.DecodePool(3) as pool:
        arr, ok, failed = pool.decode(mixed, 224)
        assert ok == paths and failed == [bad, mixed[-1]]
        assert arr.shape == (6, 3, 224, 224) and arr.dtype == np.uint8
        for a, p in zip(arr, paths):
            assert np.array_equal(a, clipmi.pipeline.load_uint8(p, 224))
        arr2, ok2, _ = pool.decode(paths[::-1], 224)                       # the segment is reused; order follows the call
        assert ok2 == paths[::-1] and np.array_equal(arr2[0], arr[5])
        got = list(clipmi.pipeline.encode_files(_StubModel(), mixed, batch=3, workers=2, pool=pool))
        # a worker that dies (a file that crashes the decoder, an OOM kill) costs at most the file it was on: its share is
        # decoded in-process from then on, nothing is spawned
        pool.procs[1].kill()
        pool.procs[1].wait()
        arr3, ok3, failed3 = pool.decode(paths, 224)
        assert pool.procs[1] is None and len(ok3) >= len(paths) - 1 and set(ok3) | set(failed3) == set(paths)
        arr4, ok4, failed4 = pool.decode(paths, 224)
        assert ok4 == paths and failed4 == [] and np.array_equal(arr4, arr)
    ref = list(clipmi.pipeline.encode_files(_StubModel(), mixed, batch=3, workers=2))
    assert [g[0] for g in got] == [r[0] for r in ref] and [g[2] for g in got] == [r[2] for r in ref]
    for g, r in zip(got, ref):
        assert (g[1] is None and r[1] is None) or np.array_equal(g[1], r[1])


def test_ctrl_c_does_not_turn_good_files_into_skipped_files(clipmi, tmp_path, monkeypatch):
    """a terminal's Ctrl-C goes to the whole process group. The decode workers sit in their own
    session and ignore SIGINT, so it never reaches them; and a file whose worker DIED under it (here: killed) fails for
    this run only - it is kept out of skip_db and the next run encodes it."""
    import signal, time
    from PIL import Image
    bi = _load_script("build-index.py")
    d = tmp_path / "lib"
    d.mkdir()
    rng = np.random.default_rng(8)
    for i in range(6):
        Image.fromarray(rng.integers(0, 256, (40, 40, 3), dtype=np.uint8)).save(str(d / f"p{i}.png"))
    (d / "broken.jpg").write_bytes(b"not an image")
    base = str(d) + "/"
    monkeypatch.chdir(tmp_path)
    db = clipmi.store.VectorStore("vectors.lmdb", dim=512, backend="packed")
    with clipmi.pipeline.DecodePool(2) as pool:
        assert all(os.getpgid(p.pid) != os.getpgid(0) for p in pool.procs)     # not in the terminal's foreground group
        os.kill(pool.procs[0].pid, signal.SIGINT)                              # and a stray SIGINT is ignored
        time.sleep(0.2)
        assert pool.procs[0].poll() is None
        pool.procs[1].kill()                                                   # a worker that dies under its first file
        pool.procs[1].wait()
        bi.encode_directories([base], _StubModel(), db, batch=4, workers=2, pool=pool)
        assert len(pool.lost) == 1
        lost = next(iter(pool.lost))
    assert db.is_skipped(base + "broken.jpg") and not db.is_skipped(lost) and not db.has_vector(lost)
    assert bi.candidates(base, db) == [lost]                                   # retried by the next run
    assert clipmi.indexer.LostLog("vectors.lmdb").counts == {lost: 1}                      # ... and remembered in vectors.lmdb.lost
    # a file that costs a worker its life AGAIN (a decoder crash, a decompression bomb) is skipped from then on -
    # here the next run's worker dies on the same file (the pool reports it lost once more)
    with clipmi.pipeline.DecodePool(1) as pool2:
        pool2.procs[0].kill()
        pool2.procs[0].wait()
        bi.encode_directories([base], _StubModel(), db, batch=4, workers=1, pool=pool2)
        assert pool2.lost == {lost}
    assert db.is_skipped(lost) and bi.candidates(base, db) == [] and db.count() == 5
    db.close()


class _StubModel:
    """encode_image/encode_text that are deterministic functions of the input (CPU, test only)."""
    embed_dim, context_length = 512, 77

    class visual:
        input_resolution = 224

    device = torch.device("cpu")

    def encode_image(self, x, normalize=False):
        g = torch.Generator().manual_seed(7)
        proj = torch.randn(3 * 8 * 8, 512, generator=g)
        f = torch.nn.functional.avg_pool2d(x.float(), 28).reshape(x.shape[0], -1) @ proj
        return f / f.norm(dim=-1, keepdim=True) if normalize else f

    def encode_text(self, ids):
        g = torch.Generator().manual_seed(int(ids.sum()) % 1000)
        return torch.randn(ids.shape[0], 512, generator=g)


def test_build_index_glue_keys_skip_resume(clipmi, tmp_path, monkeypatch, capsys):
    from PIL import Image
    bi = _load_script("build-index.py")
    d = tmp_path / "photos"
    d.mkdir()
    rng = np.random.default_rng(3)
    for n in ("b.jpg", "a.JPG", "c.png", "notes.txt"):
        if n.endswith("txt"):
            (d / n).write_text("x")
        else:
            Image.fromarray(rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)).save(str(d / n))
    (d / "broken.jpeg").write_bytes(b"not an image")
    base = str(d) + "/"                                   # trailing slash: key = base + name
    monkeypatch.chdir(tmp_path)
    db = clipmi.store.VectorStore("vectors.lmdb", dim=512, backend="packed")
    with clipmi.pipeline.DecodePool(2) as pool:           # the CLI's default: decode in worker processes
        bi.encode_directories([base], _StubModel(), db, batch=2, workers=2, pool=pool)
    out = capsys.readouterr().out
    assert out.startswith(f"CLIPing {base}...") and out.count(".") >= 3 and out.count("#") == 1
    assert db.count() == 3 and db.is_skipped(base + "broken.jpeg") and db.has_vector(base + "a.JPG")
    assert bi.candidates(base, db) == []                  # resume: nothing left to do, failures not retried
    bi.finalise(db, "cpu")
    out = capsys.readouterr().out
    assert "Preparing index for 3 entries..." in out and "Generating (3, 512) matrix..." in out and "Saving index..." in out
    idx = clipmi.read_index("images.index", device="cpu")
    assert idx.ntotal == 3
    assert db.idx_get(0) == (base + "a.JPG").encode()     # bytewise key order: 'a.JPG' < 'b.jpg' < 'c.png'
    row0 = idx.matrix().numpy()[0]
    assert np.array_equal(row0, db.get_vector(base + "a.JPG")[0]) and abs(np.linalg.norm(row0) - 1) < 1e-5
    db.close()


def test_query_repl_paging_and_similarity(clipmi, tmp_path, monkeypatch, topk_oracle):
    qi = _load_script("query-index.py")
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(4)
    N = 300
    vecs = unit_rows(rng, N, 512)
    db = clipmi.store.VectorStore("vectors.lmdb", dim=512, backend="packed")
    keys = [f"/p/{i:04d}.jpg" for i in range(N)]
    db.put_vectors(keys, vecs)
    mat, _ = db.assemble()

    class Idx:
        nprobe = 32
        calls = []

        def search(self, f, K):
            self.calls.append(K)
            return topk_oracle.topk(mat, f, K)

    idx = Idx()
    script = iter(["h", "p 50", "p 500", "c 5", "i 7", "", "r 640x480", "r junk", "a", "i 99999", "c 0", "q"])
    lines = []
    qi.repl(_StubModel(), idx, db, inp=lambda prompt: next(script), out=lambda s: lines.append(s))
    assert idx.calls == [6]                               # K = k + offset + 1 with k = 5 (query-index.py:111);
    #                                                       "more results" is ignored until a TEXT query was made
    assert "Set to probe 50 subsets." in lines and "Invalid probe value." in lines and "Showing 5 results." in lines
    assert "Similar to /p/0007.jpg:" in lines and "Not found." in lines and "Reset number of results to 50." in lines
    assert "Set maximum resolution to 640x480." in lines and "Unset maximum resolution." in lines
    res = [l for l in lines if l.count(" ") == 2 and l.split()[1].isdigit() and l.split()[2].startswith("/p/")]
    assert len(res) == 5                                  # best hit (the image itself, j = 0) is dropped
    D, I = topk_oracle.topk(mat, vecs[7:8], 6)
    assert [int(l.split()[1]) for l in res] == list(I[0][1:]) and I[0][0] == 7
    assert res[0] == f"{D[0][1]:.4f} {I[0][1]} /p/{I[0][1]:04d}.jpg"
    db.close()


def _fake_reference_env(clipmi, path, n, rng, with_idx=True):
    """An environment laid out like the reference's vectors.lmdb (build-index.py:22-24,51,61,87): fn_db path ->
    2048-B vector (always on an overflow page), skip_db path -> b"1", idx_db decimal -> path."""
    from clipmi.lmdbfile import write_environment
    keys = sorted({f"photos/{rng.integers(0, 10 ** 9):09d}/img_{i:05d}.jpg".encode() for i in range(n)})
    vecs = {k: rng.standard_normal(512).astype("<f4") for k in keys}
    tables = {b"fn_db": [(k, vecs[k].tobytes()) for k in keys],
              b"skip_db": [(b"photos/broken_%d.jpg" % i, b"1") for i in range(7)]}
    if with_idx:
        tables[b"idx_db"] = [(str(i).encode(), k) for i, k in enumerate(keys)]
    write_environment(path, tables)
    return keys, vecs


@pytest.mark.parametrize("n", [1, 40, 3000])
def test_lmdb_file_reader_roundtrip(clipmi, tmp_path, n):
    """next-2: the own LMDB-format reader (used when py-lmdb is absent) against environments written from the
    same format description: point lookups through 1-3 tree levels, overflow-page values, key-order iteration,
    entry counts, missing keys. (PARITY UNPINNED: no file from the real library is available here.)"""
    from clipmi.lmdbfile import LmdbReader
    rng = np.random.default_rng(n)
    p = str(tmp_path / "vectors.lmdb")
    keys, vecs = _fake_reference_env(clipmi, p, n, rng)
    r = LmdbReader(p)
    fn, sk, ix = r.open_db(b"fn_db"), r.open_db(b"skip_db"), r.open_db(b"idx_db")
    assert r.entries(fn) == len(keys) and r.entries(sk) == 7 and r.entries(ix) == len(keys)
    assert (fn.depth >= 2) == (len(keys) > 100)
    for k in (keys[0], keys[-1], keys[len(keys) // 2]):
        assert r.get(fn, k) == vecs[k].tobytes()
    assert r.get(fn, b"photos/none.jpg") is None and r.get(fn, b"") is None and r.get(fn, b"zzzz") is None
    assert [k for k, _ in r.items(fn)] == keys
    assert all(v == vecs[k].tobytes() for k, v in r.items(fn))
    for i in (0, len(keys) - 1):
        assert r.get(ix, str(i).encode()) == keys[i]
    assert r.get(sk, b"photos/broken_3.jpg") == b"1"
    with pytest.raises(KeyError):
        r.open_db(b"nope")
    r.close()


def test_vector_store_opens_lmdb_env_read_only_without_pylmdb(clipmi, tmp_path):
    """An existing vectors.lmdb on a machine without py-lmdb: VectorStore picks the read-only format reader;
    the query side (get_vector, idx_get, assemble in key order) works; writes are refused; convert() copies it
    into a packed store that can be extended; export_lmdb() writes a store back out as an environment."""
    pytest.importorskip("numpy")
    try:
        import lmdb  # noqa: F401
        pytest.skip("py-lmdb is installed: the lmdb backend is used instead")
    except ImportError:
        pass
    rng = np.random.default_rng(5)
    p = str(tmp_path / "vectors.lmdb")
    keys, vecs = _fake_reference_env(clipmi, p, 300, rng)
    st = clipmi.store.VectorStore(p, dim=512)
    assert st.backend_name == "lmdbfile" and st.read_only and st.count() == 300
    assert np.array_equal(st.get_vector(keys[7])[0], vecs[keys[7]])
    assert st.idx_get(299) == keys[299] and st.is_skipped("photos/broken_0.jpg") and not st.is_skipped("x")
    mat, paths = st.assemble()
    assert paths == keys and np.array_equal(mat[5], vecs[keys[5]])
    with pytest.raises(clipmi.store.ReadOnlyStore):
        st.put_vectors(["a"], np.zeros((1, 512), np.float32))
    st.close()
    q = str(tmp_path / "vectors.packed")
    clipmi.store.convert(p, q)
    st2 = clipmi.store.VectorStore(q, dim=512)
    assert st2.backend_name == "packed" and st2.count() == 300 and st2.idx_get(0) == keys[0]
    st2.put_vectors(["zzz/new.jpg"], np.ones((1, 512), np.float32))
    assert st2.count() == 301
    back = str(tmp_path / "exported.lmdb")
    st2.export_lmdb(back)
    st2.close()
    st3 = clipmi.store.VectorStore(back, dim=512)
    assert st3.backend_name == "lmdbfile" and st3.count() == 301 and st3.has_vector("zzz/new.jpg")
    st3.close()


def test_lmdb_file_reader_deep_tree_small_pages(clipmi, tmp_path):
    """512-byte pages force a 3-level tree with a few thousand keys: branch descent, separators, multi-page
    overflow runs."""
    from clipmi.lmdbfile import LmdbReader, write_environment
    rng = np.random.default_rng(11)
    keys = sorted({b"k%07d" % rng.integers(0, 10 ** 7) for _ in range(4000)})
    vals = {k: bytes(rng.integers(0, 256, int(rng.integers(0, 1500)), dtype=np.uint8)) for k in keys}
    p = str(tmp_path / "deep.lmdb")
    write_environment(p, {b"t": [(k, vals[k]) for k in keys]}, psize=512)
    r = LmdbReader(p)
    t = r.open_db(b"t")
    assert r.psize == 512 and t.depth >= 3 and r.entries(t) == len(keys)
    for k in keys[::37]:
        assert r.get(t, k) == vals[k]
    assert r.get(t, b"k0000000x") is None and r.get(t, b"a") is None
    assert [k for k, _ in r.items(t)] == keys
    r.close()
