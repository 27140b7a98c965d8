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
  * otherwise an append-only packed directory with the same three tables (py-lmdb and liblmdb are
    not installed in the build image; on-disk LMDB compatibility is therefore PARITY UNPINNED here).
An existing LMDB environment is never opened with the packed backend: that raises instead.
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


class _PackedBackend:
    """Directory of append-only logs: <table>.log = repeated [u32 klen][u32 vlen][key][value]; the last
    record for a key wins. Loaded into dicts at open (1 M x 2 KiB vectors = 2 GiB: values of fn_db are
    kept as offsets into a memory map, not copies)."""

    TABLES = ("fn_db", "skip_db", "idx_db")

    def __init__(self, path, dim):
        if os.path.exists(os.path.join(path, "data.mdb")):
            raise RuntimeError(f"{path} is an LMDB environment but the `lmdb` module is not installed; "
                               "install py-lmdb to open existing databases")
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
            backend = "lmdb" if _have_lmdb() else "packed"
        self.backend_name = backend
        self.b = _LmdbBackend(path, dim) if backend == "lmdb" else _PackedBackend(path, dim)

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
        self.b.put_many("idx_db", [(str(i).encode(), p) for i, p in enumerate(paths)])
        return mat, paths

    def idx_get(self, i):
        return self.b.get("idx_db", str(int(i)).encode())                    # query-index.py:92,117-118

    def close(self):
        self.b.close()
