"""Vector store with the reference's schema.

The reference keeps three named LMDB tables in `vectors.lmdb` (map size 20 GiB, max_dbs 4:
build-index.py:12,22; query-index.py:25):
    fn_db    path bytes            -> 512 x f32 little-endian (2048 B)   build-index.py:23,51
    skip_db  path bytes            -> b"1"                               build-index.py:24,61
    idx_db   ASCII decimal row id  -> path bytes                         build-index.py:66,87-88
and derives the row order of the packed matrix from fn_db's KEY ORDER (bytewise-sorted paths,
build-index.py:75-89).

Backends:
  * `lmdb` (py-lmdb) when importable: the reference's exact environment / table names, so existing
    databases open unchanged. Commits are per BATCH, not per image (the reference fsyncs once per
    image, build-index.py:42, and once per row, :87).
  * `lmdbfile`: an existing LMDB environment (`data.mdb`) on a machine WITHOUT py-lmdb is opened READ-ONLY by
    the own format reader in lmdbfile.py: everything query-index.py does works (it only reads); adding images
    needs py-lmdb or `convert()` into a packed store first. Format restated from mdb.c: PARITY UNPINNED here.
  * otherwise an append-only packed directory with the same three tables.
`export_lmdb()` writes any store out as an LMDB environment for the reference's own tools.
"""
import os
import struct

import numpy as np

MAP_SIZE = 1024 * 1024 * 1024 * 20     # build-index.py:12


def _have_lmdb():
    try:
        import lmdb  # noqa: F401
        return True
    except ImportError:
        return False


class _LmdbBackend:
    def __init__(self, path, dim):
        import lmdb
        self.env = lmdb.open(path, map_size=MAP_SIZE, max_dbs=4)
        self.fn_db = self.env.open_db(b"fn_db")
        self.skip_db = self.env.open_db(b"skip_db")
        self.idx_db = self.env.open_db(b"idx_db")

    def get(self, table, key):
        with self.env.begin(db=getattr(self, table)) as txn:
            return txn.get(key)

    def put_many(self, table, items):
        with self.env.begin(db=getattr(self, table), write=True) as txn:
            for k, v in items:
                txn.put(k, v, dupdata=False, overwrite=True)

    def count(self, table):
        with self.env.begin(db=getattr(self, table)) as txn:
            return txn.stat()["entries"]

    def items_sorted(self, table):
        with self.env.begin(db=getattr(self, table)) as txn:
            for k, v in txn.cursor():
                yield bytes(k), bytes(v)

    def close(self):
        self.env.close()


class ReadOnlyStore(RuntimeError):
    pass


class _LmdbFileBackend:
    """Read-only: lmdbfile.LmdbReader over an existing environment (no py-lmdb on this machine)."""

    def __init__(self, path, dim):
        from .lmdbfile import LmdbReader
        self.path = path
        self.r = LmdbReader(path)
        self.dbs = {}
        for t in ("fn_db", "skip_db", "idx_db"):
            try:
                self.dbs[t] = self.r.open_db(t.encode())
            except KeyError:
                self.dbs[t] = None                     # table never created (e.g. idx_db before the first build)

    def get(self, table, key):
        db = self.dbs[table]
        return None if db is None else self.r.get(db, key)

    def put_many(self, table, items):
        raise ReadOnlyStore(f"{self.path} is an LMDB environment opened read-only (the `lmdb` module is not installed): "
                            "install py-lmdb, or convert it once with clipmi.store.convert(src, dst)")

    def count(self, table):
        db = self.dbs[table]
        return 0 if db is None else self.r.entries(db)

    def items_sorted(self, table):
        db = self.dbs[table]
        return iter(()) if db is None else self.r.items(db)

    def close(self):
        self.r.close()


class _PackedBackend:
    """Directory of append-only logs: <table>.log = repeated [u32 klen][u32 vlen][key][value]; the last
    record for a key wins. Loaded into dicts at open (1 M x 2 KiB vectors = 2 GiB: values of fn_db are
    kept as offsets into a memory map, not copies)."""

    TABLES = ("fn_db", "skip_db", "idx_db")

    def __init__(self, path, dim):
        if os.path.exists(os.path.join(path, "data.mdb")):
            raise RuntimeError(f"{path} is an LMDB environment: open it with backend 'lmdb' or 'lmdbfile'")
        os.makedirs(path, exist_ok=True)
        self.path = path
        self.index = {t: {} for t in self.TABLES}        # key -> (offset, length) of the value
        self.files = {}
        for t in self.TABLES:
            fn = os.path.join(path, t + ".log")
            f = open(fn, "a+b")
            self.files[t] = f
            self._scan(t)

    def _scan(self, t):
        f = self.files[t]
        f.seek(0, os.SEEK_END)
        end = f.tell()
        pos = 0
        idx = self.index[t]
        while pos + 8 <= end:
            f.seek(pos)
            klen, vlen = struct.unpack("<II", f.read(8))
            if pos + 8 + klen + vlen > end:
                break                                      # torn tail of an interrupted write: ignored
            key = f.read(klen)
            idx[key] = (pos + 8 + klen, vlen)
            pos += 8 + klen + vlen
        f.truncate(pos)

    def get(self, table, key):
        loc = self.index[table].get(key)
        if loc is None:
            return None
        f = self.files[table]
        f.seek(loc[0])
        return f.read(loc[1])

    def put_many(self, table, items):
        f = self.files[table]
        f.seek(0, os.SEEK_END)
        pos = f.tell()
        chunks = []
        for k, v in items:
            chunks.append(struct.pack("<II", len(k), len(v)) + k + v)
            self.index[table][k] = (pos + 8 + len(k), len(v))
            pos += 8 + len(k) + len(v)
        f.write(b"".join(chunks))
        f.flush()
        os.fsync(f.fileno())

    def count(self, table):
        return len(self.index[table])

    def items_sorted(self, table):
        for k in sorted(self.index[table]):                # bytewise order = LMDB's default key order
            yield k, self.get(table, k)

    def close(self):
        for f in self.files.values():
            f.close()


class VectorStore:
    def __init__(self, path="vectors.lmdb", dim=512, backend=None):
        self.dim = dim
        if backend is None:
            is_env = os.path.exists(os.path.join(path, "data.mdb"))
            backend = "lmdb" if _have_lmdb() else ("lmdbfile" if is_env else "packed")
        self.backend_name = backend
        self.read_only = backend == "lmdbfile"
        self.b = (_LmdbBackend(path, dim) if backend == "lmdb" else
                  _LmdbFileBackend(path, dim) if backend == "lmdbfile" else _PackedBackend(path, dim))

    # ---- fn_db ------------------------------------------------------------------------------
    def has_vector(self, key):
        return self.b.get("fn_db", key.encode()) is not None

    def get_vector(self, key):
        raw = self.b.get("fn_db", key if isinstance(key, bytes) else key.encode())
        if raw is None:
            return None
        return np.frombuffer(raw, dtype="<f4").reshape((1, self.dim))       # query-index.py:95

    def put_vectors(self, keys, vectors):
        """One commit for the whole batch. vectors: f32 [n, dim] (already normalised)."""
        v = np.ascontiguousarray(vectors, dtype="<f4")
        assert v.shape == (len(keys), self.dim)
        self.b.put_many("fn_db", [(k.encode(), v[i].tobytes()) for i, k in enumerate(keys)])

    def count(self):
        return self.b.count("fn_db")

    # ---- skip_db ----------------------------------------------------------------------------
    def is_skipped(self, key):
        return self.b.get("skip_db", key.encode()) is not None

    def mark_skipped(self, keys):
        if keys:
            self.b.put_many("skip_db", [(k.encode(), b"1") for k in keys])

    # ---- matrix assembly + idx_db (build-index.py:68-107) ---------------------------------------
    def assemble(self):
        """Rows in fn_db key order -> (f32 [n, dim] matrix, list of path bytes); writes idx_db[i] = path
        for every row in ONE commit. Values are read straight into f32 (the reference goes through a
        float64 scratch matrix, build-index.py:77, with bit-identical results)."""
        n = self.count()
        mat = np.empty((n, self.dim), dtype=np.float32)
        paths = []
        for i, (k, v) in enumerate(self.b.items_sorted("fn_db")):
            mat[i] = np.frombuffer(v, dtype="<f4")
            paths.append(k)
        if self.read_only:
            # a read-only environment keeps the idx_db its last build wrote; it must describe the same rows
            if self.b.count("idx_db") != n or any(self.idx_get(i) != p for i, p in ((0, paths[0]), (n - 1, paths[-1])) if n):
                raise ReadOnlyStore("idx_db of the read-only environment does not match fn_db's key order; "
                                    "convert the store (clipmi.store.convert) and re-run the build")
        else:
            self.b.put_many("idx_db", [(str(i).encode(), p) for i, p in enumerate(paths)])
        return mat, paths

    def idx_get(self, i):
        return self.b.get("idx_db", str(int(i)).encode())                    # query-index.py:92,117-118

    def export_lmdb(self, path):
        """Write this store out as an LMDB environment (data.mdb) with the reference's three tables, for the
        reference's own tools (lmdbfile.write_environment: one bulk transaction)."""
        from .lmdbfile import write_environment
        write_environment(path, {t.encode(): list(self.b.items_sorted(t)) for t in ("fn_db", "skip_db", "idx_db")},
                          mapsize=MAP_SIZE)

    def close(self):
        self.b.close()


def convert(src, dst, dim=512, dst_backend="packed"):
    """Copy every table of the store at `src` (any backend, e.g. an LMDB environment opened read-only) into a
    new store at `dst`."""
    a, b = VectorStore(src, dim=dim), VectorStore(dst, dim=dim, backend=dst_backend)
    try:
        for t in ("fn_db", "skip_db", "idx_db"):
            batch = []
            for kv in a.b.items_sorted(t):
                batch.append(kv)
                if len(batch) >= 4096:
                    b.b.put_many(t, batch)
                    batch = []
            if batch:
                b.b.put_many(t, batch)
    finally:
        a.close()
        b.close()
