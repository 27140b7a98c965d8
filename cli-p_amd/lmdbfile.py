"""Read-only reader (and a bulk writer for export / fixtures) of LMDB's on-disk format, for machines without
py-lmdb: `vectors.lmdb/data.mdb` written by the reference (build-index.py:22-24, lmdb.open(..., max_dbs=4) with
the named tables fn_db, skip_db, idx_db) can then still be QUERIED (query-index.py:25-27,92-95,117-118 only reads).

Format restated from LMDB's mdb.c (OpenLDAP, data version 1, 64-bit little-endian, no MDB_DEVEL):
  page header (16 B): pgno u64 | pad u16 | flags u16 | lower u16 | upper u16   (overflow pages: pages u32 in place
      of lower/upper); flags: BRANCH 0x01, LEAF 0x02, OVERFLOW 0x04, META 0x08, LEAF2 0x20, SUBP 0x40
  after the header: u16 offsets of the nodes in key order (count = (lower - 16) / 2); nodes sit at even offsets
      from `upper` to the end of the page
  node (8 B header): lo u16 | hi u16 | flags u16 | ksize u16 | key | data
      leaf:   data size = lo | hi << 16; flags: BIGDATA 0x01 (data = u64 first page of an overflow run whose payload
              starts 16 B into it), SUBDATA 0x02 (data = MDB_db record of a named table), DUPDATA 0x04 (unsupported)
      branch: child pgno = lo | hi << 16 | flags << 32; node 0's key is empty (= minus infinity)
  meta pages 0 and 1: page header, then magic u32 0xBEEFC0DE | version u32 1 | address u64 | mapsize u64 |
      MDB_db free | MDB_db main | last_pg u64 | txnid u64;  the meta with the larger txnid wins
  MDB_db (48 B): pad u32 (the FREE record's pad = page size) | flags u16 | depth u16 | branch_pages u64 |
      leaf_pages u64 | overflow_pages u64 | entries u64 | root u64 (all ones = empty)
  values larger than ((psize - 16) / 2 & ~1) - 2 - 8 - ksize bytes go to overflow pages: the reference's 2048-byte
      vectors always do (one 4-KiB page each)
  keys compare bytewise (memcmp, shorter first on a tie): the default comparator, which the reference uses.

PARITY UNPINNED: neither liblmdb nor py-lmdb exists in the build environment, so no file written by the real
library was available; tests round-trip through the writer below, which follows the same description. When
`lmdb` is importable, store.py uses it instead of this module.
"""
import mmap
import os
import struct

MAGIC = 0xBEEFC0DE
P_BRANCH, P_LEAF, P_OVERFLOW, P_META, P_LEAF2 = 0x01, 0x02, 0x04, 0x08, 0x20
F_BIGDATA, F_SUBDATA, F_DUPDATA = 0x01, 0x02, 0x04
P_INVALID = 0xFFFFFFFFFFFFFFFF
PAGEHDR = 16
_DB = struct.Struct("<IHHQQQQQ")          # MDB_db, 48 bytes


class LmdbFormatError(ValueError):
    pass


class _Db:
    def __init__(self, rec):
        (self.pad, self.flags, self.depth, self.branch_pages, self.leaf_pages, self.overflow_pages, self.entries,
         self.root) = rec


class LmdbReader:
    """Read-only view of an LMDB environment directory (or data file)."""

    def __init__(self, path):
        f = os.path.join(path, "data.mdb") if os.path.isdir(path) else path
        self._fh = open(f, "rb")
        size = os.fstat(self._fh.fileno()).st_size
        if size < 2 * 512:
            raise LmdbFormatError(f"{f}: too small for an LMDB environment")
        self._m = mmap.mmap(self._fh.fileno(), 0, access=mmap.ACCESS_READ)
        metas = []
        # the page size is only known from the meta itself: meta 1 starts at one page size; try the usual ones
        first = self._meta_at(0)
        if first is None:
            raise LmdbFormatError(f"{f}: bad magic / version in meta page 0")
        self.psize = first["psize"]
        metas.append(first)
        second = self._meta_at(self.psize)
        if second is not None:
            metas.append(second)
        meta = max(metas, key=lambda m: m["txnid"])
        self.main = meta["main"]
        self.last_pg = meta["last_pg"]
        if (self.last_pg + 1) * self.psize > size:
            raise LmdbFormatError(f"{f}: truncated (last page {self.last_pg}, file {size} bytes)")
        self._named = {}

    def close(self):
        self._m.close()
        self._fh.close()

    def _meta_at(self, off):
        if off + PAGEHDR + 136 > len(self._m):
            return None
        magic, version = struct.unpack_from("<II", self._m, off + PAGEHDR)
        if magic != MAGIC or version != 1:
            return None
        free = _Db(_DB.unpack_from(self._m, off + PAGEHDR + 24))
        main = _Db(_DB.unpack_from(self._m, off + PAGEHDR + 24 + 48))
        last_pg, txnid = struct.unpack_from("<QQ", self._m, off + PAGEHDR + 24 + 96)
        return {"psize": free.pad, "main": main, "last_pg": last_pg, "txnid": txnid}

    # ---- pages and nodes ------------------------------------------------------------------------
    def _page(self, pgno):
        off = pgno * self.psize
        if pgno > self.last_pg:
            raise LmdbFormatError(f"page {pgno} beyond the last page {self.last_pg}")
        _, _, flags, lower, upper = struct.unpack_from("<QHHHH", self._m, off)
        return off, flags, (lower - PAGEHDR) // 2

    def _node(self, off, i):
        (ptr,) = struct.unpack_from("<H", self._m, off + PAGEHDR + 2 * i)
        lo, hi, flags, ksize = struct.unpack_from("<HHHH", self._m, off + ptr)
        return off + ptr, lo, hi, flags, ksize

    def _key(self, noff, ksize):
        return self._m[noff + 8:noff + 8 + ksize]

    def _leaf_value(self, noff, lo, hi, flags, ksize):
        size = lo | (hi << 16)
        d = noff + 8 + ksize
        if flags & F_DUPDATA:
            raise LmdbFormatError("DUPSORT tables are not supported")
        if flags & F_BIGDATA:
            (pg,) = struct.unpack_from("<Q", self._m, d)
            o = pg * self.psize
            (_, _, pflags, _npages) = struct.unpack_from("<QHHI", self._m, o)
            if not pflags & P_OVERFLOW:
                raise LmdbFormatError(f"page {pg} is not an overflow page")
            return self._m[o + PAGEHDR:o + PAGEHDR + size]
        return self._m[d:d + size]

    def _search_leaf(self, db, key):
        if db.root == P_INVALID:
            return None
        pg = db.root
        for _ in range(64):
            off, flags, n = self._page(pg)
            if flags & P_LEAF2:
                raise LmdbFormatError("LEAF2 (DUPFIXED) pages are not supported")
            if flags & P_LEAF:
                return off, n
            if not flags & P_BRANCH or n < 1:
                raise LmdbFormatError(f"page {pg}: unexpected flags {flags:#x}")
            lo_i, hi_i = 1, n - 1                      # node 0 = minus infinity
            child = 0
            while lo_i <= hi_i:                        # last node whose key <= search key
                mid = (lo_i + hi_i) // 2
                noff, _, _, _, ks = self._node(off, mid)
                if self._key(noff, ks) <= key:
                    child = mid
                    lo_i = mid + 1
                else:
                    hi_i = mid - 1
            noff, lo, hi, fl, _ = self._node(off, child)
            pg = lo | (hi << 16) | (fl << 32)
        raise LmdbFormatError("tree deeper than 64 levels")

    def _get(self, db, key):
        hit = self._search_leaf(db, key)
        if hit is None:
            return None
        off, n = hit
        lo_i, hi_i = 0, n - 1
        while lo_i <= hi_i:
            mid = (lo_i + hi_i) // 2
            noff, lo, hi, fl, ks = self._node(off, mid)
            k = self._key(noff, ks)
            if k == key:
                return noff, lo, hi, fl, ks
            if k < key:
                lo_i = mid + 1
            else:
                hi_i = mid - 1
        return None

    def _walk(self, pg):
        off, flags, n = self._page(pg)
        if flags & P_LEAF:
            for i in range(n):
                yield self._node(off, i)
        elif flags & P_BRANCH:
            for i in range(n):
                _, lo, hi, fl, _ = self._node(off, i)
                yield from self._walk(lo | (hi << 16) | (fl << 32))
        else:
            raise LmdbFormatError(f"page {pg}: unexpected flags {flags:#x}")

    # ---- the interface store.py uses ----------------------------------------------------------------
    def open_db(self, name):
        """Named table (bytes name) -> handle; raises KeyError if the environment has no such table."""
        if name not in self._named:
            hit = self._get(self.main, name)
            if hit is None:
                raise KeyError(name)
            noff, lo, hi, fl, ks = hit
            if not fl & F_SUBDATA:
                raise LmdbFormatError(f"{name!r} is a plain record of the main table, not a named table")
            self._named[name] = _Db(_DB.unpack_from(self._m, noff + 8 + ks))
        return self._named[name]

    def get(self, db, key):
        hit = self._get(db, key)
        return None if hit is None else bytes(self._leaf_value(*hit))

    def entries(self, db):
        return db.entries

    def items(self, db):
        """(key, value) in key order."""
        if db.root == P_INVALID:
            return
        for noff, lo, hi, fl, ks in self._walk(db.root):
            yield bytes(self._key(noff, ks)), bytes(self._leaf_value(noff, lo, hi, fl, ks))


# =================================================================================================
# Bulk writer: a fresh environment from complete, sorted tables (export of a packed store for the reference's
# tools; fixtures for the reader's tests). One transaction, no free list, every page written once.
# =================================================================================================
def write_environment(path, tables, psize=4096, mapsize=20 * 1024 ** 3):
    """tables: {name bytes: iterable of (key bytes, value bytes)}; keys need not be sorted, must be unique.
    Writes <path>/data.mdb (and an empty lock.mdb, which LMDB recreates as needed)."""
    os.makedirs(path, exist_ok=True)
    nodemax = (((psize - PAGEHDR) // 2) & ~1) - 2
    pages = {}                                      # pgno -> bytes
    next_pg = [2]

    def alloc(n=1):
        pg = next_pg[0]
        next_pg[0] += n
        return pg

    def build_page(pgno, flags, nodes):
        """nodes: list of complete node byte strings in key order."""
        buf = bytearray(psize)
        upper = psize
        ptrs = []
        for nd in nodes:
            sz = (len(nd) + 1) & ~1
            upper -= sz
            buf[upper:upper + len(nd)] = nd
            ptrs.append(upper)
        lower = PAGEHDR + 2 * len(nodes)
        assert lower <= upper, "page overfull"
        struct.pack_into("<QHHHH", buf, 0, pgno, 0, flags, lower, upper)
        struct.pack_into(f"<{len(ptrs)}H", buf, PAGEHDR, *ptrs)
        pages[pgno] = bytes(buf)

    def build_tree(items):
        """items: sorted (key, leaf node flags, inline data bytes, full data size) -> MDB_db fields."""
        counts = {"branch": 0, "leaf": 0}
        level = []                                   # (first key, pgno) of the pages of the current level
        cur, room = [], psize - PAGEHDR

        def flush_leaf():
            pg = alloc()
            build_page(pg, P_LEAF, [n for _, n in cur])
            counts["leaf"] += 1
            level.append((cur[0][0], pg))
        for key, nflags, data, dsize in items:
            node = struct.pack("<HHHH", dsize & 0xFFFF, dsize >> 16, nflags, len(key)) + key + data
            need = ((len(node) + 1) & ~1) + 2
            if need > room and cur:
                flush_leaf()
                cur, room = [], psize - PAGEHDR
            cur.append((key, node))
            room -= need
        if not cur and not level:
            return (0, 0, 0, 0, 0, P_INVALID)
        if cur:
            flush_leaf()
        depth = 1
        while len(level) > 1:
            nxt, cur, room = [], [], psize - PAGEHDR
            for i, (key, pg) in enumerate(level):
                k = b"" if not cur else key          # first node of a branch page: empty key
                node = struct.pack("<HHHH", pg & 0xFFFF, (pg >> 16) & 0xFFFF, pg >> 32, len(k)) + k
                need = ((len(node) + 1) & ~1) + 2
                if need > room and len(cur) >= 2:
                    bp = alloc()
                    build_page(bp, P_BRANCH, [n for _, n in cur])
                    counts["branch"] += 1
                    nxt.append((cur[0][0], bp))
                    cur, room = [], psize - PAGEHDR
                    node = struct.pack("<HHHH", pg & 0xFFFF, (pg >> 16) & 0xFFFF, pg >> 32, 0)
                    need = ((len(node) + 1) & ~1) + 2
                cur.append((key, node))
                room -= need
            bp = alloc()
            build_page(bp, P_BRANCH, [n for _, n in cur])
            counts["branch"] += 1
            nxt.append((cur[0][0], bp))
            level = nxt
            depth += 1
        return (depth, counts["branch"], counts["leaf"], 0, 0, level[0][1])

    main_items = []
    for name in sorted(tables):
        recs = sorted(tables[name], key=lambda kv: kv[0])
        items, overflow = [], 0
        for i, (k, v) in enumerate(recs):
            if i and recs[i - 1][0] == k:
                raise ValueError(f"duplicate key {k!r} in table {name!r}")
            if not 0 < len(k) <= 511:
                raise ValueError(f"key length {len(k)} outside LMDB's 1..511")
            if 8 + len(k) + len(v) > nodemax:
                npg = (PAGEHDR - 1 + len(v)) // psize + 1
                pg = alloc(npg)
                buf = bytearray(npg * psize)
                struct.pack_into("<QHHI", buf, 0, pg, 0, P_OVERFLOW, npg)
                buf[PAGEHDR:PAGEHDR + len(v)] = v
                for j in range(npg):
                    pages[pg + j] = bytes(buf[j * psize:(j + 1) * psize])
                overflow += npg
                items.append((k, F_BIGDATA, struct.pack("<Q", pg), len(v)))
            else:
                items.append((k, 0, v, len(v)))
        depth, nb, nl, _, _, root = build_tree(items)
        rec = _DB.pack(0, 0, depth, nb, nl, overflow, len(recs), root)
        main_items.append((name, F_SUBDATA, rec, len(rec)))
    depth, nb, nl, _, _, root = build_tree(main_items)
    main_rec = _DB.pack(0, 0, depth, nb, nl, 0, len(main_items), root)
    free_rec = _DB.pack(psize, 0, 0, 0, 0, 0, 0, P_INVALID)
    last_pg = next_pg[0] - 1
    for i, txnid in ((0, 0), (1, 1)):
        buf = bytearray(psize)
        struct.pack_into("<QHHHH", buf, 0, i, 0, P_META, 0, 0)
        struct.pack_into("<IIQQ", buf, PAGEHDR, MAGIC, 1, 0, mapsize)
        if txnid:                                    # meta 0 (txn 0) describes the empty environment
            buf[PAGEHDR + 24:PAGEHDR + 72] = free_rec
            buf[PAGEHDR + 72:PAGEHDR + 120] = main_rec
            struct.pack_into("<QQ", buf, PAGEHDR + 120, last_pg, txnid)
        else:
            buf[PAGEHDR + 24:PAGEHDR + 72] = free_rec
            buf[PAGEHDR + 72:PAGEHDR + 120] = _DB.pack(0, 0, 0, 0, 0, 0, 0, P_INVALID)
            struct.pack_into("<QQ", buf, PAGEHDR + 120, 1, 0)
        pages[i] = bytes(buf)
    with open(os.path.join(path, "data.mdb"), "wb") as f:
        for pg in range(last_pg + 1):
            f.write(pages[pg])
    open(os.path.join(path, "lock.mdb"), "ab").close()
