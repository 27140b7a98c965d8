"""Flat inner-product index: the host-side mirror of the faiss objects the reference scripts use.

Reference call sites (paths under the reference tree):
  build-index.py:80-81   faiss.IndexFlatIP(512) / IndexIVFFlat(..., METRIC_INNER_PRODUCT)
  build-index.py:96,99   index.train(images) / index.add(images)
  build-index.py:109     faiss.write_index(index, "images.index")
  query-index.py:29-30   faiss.read_index("images.index"); index.nprobe = 32
  query-index.py:111     D, I = index.search(features, k + offset + 1)

Search here is EXACT (the reference's IVF list scan is approximate): `train` is a
no-op and `nprobe` is accepted and ignored. The packed [N][d] f32 matrix lives in HBM; search
calls clipmi_topk_ip through the C ABI. Multi-GPU: one process per GPU, rows split contiguously
by rank, per-rank top-K, ONE all-gather (RCCL when the tensors are on the GPU), then the same
K-way merge on every rank (clipmi_merge_topk) — SURVEY.md §8e.
"""
import struct

import os

import numpy as np
import torch

from . import _lib

MAGIC = b"CLIPMIIX"   # own packed file format, see write_index
METRIC_INNER_PRODUCT = 0


def shard_bounds(n_total, world_size, rank):
    """Contiguous row split: rank r owns [lo, hi). Every rank computes the same bounds."""
    base, rem = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def merge_lists_host(scores, ids, K):
    """Host restatement of the merge rule for the CPU (gloo) path of ShardedFlatIP tests:
    scores f32 [R,Q,K], ids i64 [R,Q,K] -> (f32 [Q,K], i64 [Q,K]); score desc, id asc, -1 = empty."""
    R, Q, _ = scores.shape
    out_s = np.full((Q, K), -np.finfo(np.float32).max, dtype=np.float32)
    out_i = np.full((Q, K), -1, dtype=np.int64)
    for q in range(Q):
        s = scores[:, q, :].reshape(-1)
        i = ids[:, q, :].reshape(-1)
        ok = (i >= 0) & ~np.isnan(s)
        s, i = s[ok], i[ok]
        order = np.lexsort((i, -s.astype(np.float64)))[:K]
        out_s[q, :len(order)] = s[order]
        out_i[q, :len(order)] = i[order]
    return out_s, out_i


class IndexFlatIP:
    """Exact inner-product index over f32 vectors resident in HBM."""

    def __init__(self, d, device="cuda:0", coarse=None):
        """coarse="bf16": keep a bf16 copy of the matrix beside the f32 one (+50 % HBM) and search through
        clipmi_topk_ip_coarse — a bf16-MFMA scan that keeps a provable superset, then exact f32
        re-scoring: same bit-exact results, half the bytes per pass, 64 queries per pass.
        coarse="int8": the same with a per-row-scaled int8 copy (+25 % HBM + 8 B per row): a quarter of the
        f32 bytes per pass; the superset bound uses each row's exact quantisation-error norm, so rows with
        one dominant component loosen it (more survivors, or the exact fallback) but never change results."""
        if coarse not in (None, "bf16", "int8"):
            raise ValueError("IndexFlatIP: coarse must be None, 'bf16' or 'int8'")
        self.coarse = coarse
        self._dbh = None
        self._rmax = None
        self._db8 = None
        if d not in (512, 768):
            raise ValueError("IndexFlatIP: d must be 512 (ViT-B/32) or 768 (ViT-L/14)")
        self.d = int(d)
        self.device = torch.device(device)
        self.nprobe = 1          # accepted for drop-in compatibility (query-index.py:30,51); ignored
        self.is_trained = True
        self.metric_type = METRIC_INNER_PRODUCT
        self.id_base = 0         # global id of local row 0 (non-zero on shards)
        self._chunks = []
        self._db = None
        self._ws = {}            # workspace per HIP stream: searches in flight on different streams never share one

    # -- build side ---------------------------------------------------------------------------
    def train(self, x):
        """No-op: exact search needs no k-means (build-index.py:96,105)."""
        return None

    def add(self, x):
        """Append rows (numpy f32 [n,d] as in build-index.py:99,107, or a torch tensor)."""
        if not isinstance(x, torch.Tensor):
            x = np.asarray(x)
            if not x.flags.writeable:
                x = x.copy()                # torch warns on (and may alias) read-only buffers, e.g. np.frombuffer views
            t = torch.as_tensor(x)
        else:
            t = x
        if t.dim() != 2 or t.shape[1] != self.d:
            raise ValueError(f"add: expected [n,{self.d}], got {tuple(t.shape)}")
        t = t.to(device=self.device, dtype=torch.float32).contiguous()
        self._chunks.append(t)
        self._db = None
        self._dbh = None
        self._db8 = None
        self._rmax = None

    @property
    def ntotal(self):
        return sum(c.shape[0] for c in self._chunks)

    def matrix(self):
        """The packed [N][d] f32 device matrix (rows in add order = LMDB key order, build-index.py:75-89)."""
        if self._db is None:
            if not self._chunks:
                self._db = torch.empty((0, self.d), dtype=torch.float32, device=self.device)
            elif len(self._chunks) == 1:
                self._db = self._chunks[0]
            else:
                self._db = torch.cat(self._chunks, dim=0)
                self._chunks = [self._db]
        return self._db

    def matrix_bf16(self):
        """(bf16 copy [N][d], upper bound of the largest row norm) for the coarse path; built once (clipmi_rows_to_bf16)."""
        if self._dbh is None:
            db = self.matrix()
            dbh = torch.empty(db.shape, dtype=torch.bfloat16, device=db.device)
            if db.shape[0]:
                _lib.check(_lib.lib().clipmi_rows_to_bf16(db.data_ptr(), db.shape[0], self.d, dbh.data_ptr(),
                                                          _lib.stream_ptr(self.device)), "clipmi_rows_to_bf16")
            self._dbh = dbh
        return self._dbh, self._row_norm_max()

    def _stats(self, meta=None):
        """(largest row norm, largest error norm of the int8 copy's meta) by clipmi_rows_stats, each with a 1e-6 margin."""
        db = self.matrix()
        if db.shape[0] == 0:
            return 0.0, 0.0
        if self.device.type != "cuda":
            raise _lib.ClipmiError("the coarse copies need the HIP path (device is not a GPU); no CPU fallback")
        out = torch.empty(2, dtype=torch.float32, device=db.device)
        _lib.check(_lib.lib().clipmi_rows_stats(db.data_ptr(), db.shape[0], self.d, meta.data_ptr() if meta is not None else None,
                                                out.data_ptr(), _lib.stream_ptr(self.device)), "clipmi_rows_stats")
        rmax, amax = (float(v) for v in out.cpu())
        return rmax * (1.0 + 1e-6), amax * (1.0 + 1e-6)

    def _row_norm_max(self):
        if self._rmax is None:
            self._rmax = self._stats()[0]
        return self._rmax

    def matrix_i8(self):
        """(int8 copy in 32-row blocks of [d / 32][64][16 B] - include/clipmi.h clipmi_quantize_rows_i8 -, f32 meta:
        (block scale, error norm) per row padded to 32 rows (+32) then (scale, largest error norm) per block, largest
        error norm, largest row norm) for the int8 coarse path; built once by clipmi_quantize_rows_i8 + clipmi_rows_stats."""
        if self._db8 is None:
            L = _lib.lib()
            db = self.matrix()
            N = db.shape[0]
            q8 = torch.empty(L.clipmi_i8_copy_bytes(N, self.d), dtype=torch.int8, device=db.device)
            meta = torch.zeros(L.clipmi_i8_meta_bytes(N) // 4, dtype=torch.float32, device=db.device)
            # The copy holds the rows ORDERED BY THEIR LARGEST |COMPONENT| (round 4): the 32 rows of a block then share a scale
            # that is nearly each row's own (error norms x 1.265 -> x 1.000 on unit rows, ~12 % fewer exactly re-scored rows);
            # the library maps surviving slots back to row ids, results are row ids in add order as ever. The order is built
            # once per index by the library itself (clipmi_rows_order_by_absmax: row maxima + a stable radix sort on the device;
            # round 5 - a framework sort did this before). CLIPMI_I8_SORT=0: rows in add order.
            perm = ws = None
            if os.environ.get("CLIPMI_I8_SORT", "1") != "0" and N > 32:
                perm = torch.empty(N, dtype=torch.int32, device=db.device)
                ws = torch.empty(L.clipmi_rows_order_workspace_bytes(N), dtype=torch.uint8, device=db.device)
                _lib.check(L.clipmi_rows_order_by_absmax(db.data_ptr(), N, self.d, perm.data_ptr(), ws.data_ptr(), ws.numel(),
                                                         _lib.stream_ptr(self.device)), "clipmi_rows_order_by_absmax")
            _lib.check(L.clipmi_quantize_rows_i8(db.data_ptr(), N, self.d, perm.data_ptr() if perm is not None else None,
                                                 q8.data_ptr(), q8.numel(), meta.data_ptr(), meta.numel() * 4,
                                                 _lib.stream_ptr(self.device)), "clipmi_quantize_rows_i8")
            if perm is not None:
                torch.cuda.current_stream(self.device).synchronize()      # perm / ws are read by the kernels just enqueued
            del perm, ws
            self._rmax, amax = self._stats(meta)
            self._db8 = (q8, meta, amax)
        return self._db8 + (self._row_norm_max(),)

    def uses_coarse(self):
        """True when searches go through a coarse-then-exact path (coarse copy requested, d = 512, N >= 65536)."""
        return self.coarse in ("bf16", "int8") and self.d == 512 and self.ntotal >= 65536

    # -- query side ---------------------------------------------------------------------------
    def search_device(self, q, K, out=None, _one_pass=False):
        """q: f32 [Q,d] device tensor -> (scores f32 [Q,K], ids i64 [Q,K]) device tensors. Async.
        `out` = (scores, ids) views to write into (the packed all-gather record of ShardedFlatIP)."""
        L = _lib.lib()
        db = self.matrix()
        if self.device.type != "cuda":
            raise _lib.ClipmiError("IndexFlatIP.search needs the HIP path (device is not a GPU); no CPU fallback")
        q = q.to(device=self.device, dtype=torch.float32).contiguous()
        Q = q.shape[0]
        N = db.shape[0]
        coarse = self.uses_coarse()
        if coarse and self.coarse == "int8":
            db8, meta, amax, rmax = self.matrix_i8()
            coarse = rmax > 0.0 and np.isfinite(rmax) and np.isfinite(amax)
        elif coarse:
            dbh, rmax = self.matrix_bf16()
            coarse = rmax > 0.0 and np.isfinite(rmax)
        # more than 64 queries: the int8 path takes the whole search as wide passes inside the library (one stream of the
        # copy per <= 1024 queries, csrc/topk.hip "Wide coarse pass"); the bf16 path pipelines its 64-query passes here
        if coarse and self.coarse != "int8" and Q > self.PASS_Q and self.batches_in_flight > 1 and not _one_pass:
            return self._search_pipelined(q, K, out, self.PASS_Q)
        # int8, more than one wide chunk (1024 queries): the chunks alternate between two streams the same way - one chunk's
        # re-scoring and selects run beside the other's matrix-bound scan (10 M rows, 2 x 1024 queries: 6.37 -> 6.07 ms per chunk,
        # 160.9 -> 168.8 k q/s; a single chunk cut in two halves gains nothing: 162.4 k)
        if coarse and self.coarse == "int8" and Q > self.WIDE_Q and self.batches_in_flight > 1 and not _one_pass:
            return self._search_pipelined(q, K, out, self.WIDE_Q)
        need = (L.clipmi_topk_ip_coarse_workspace_bytes if coarse else L.clipmi_topk_ip_workspace_bytes)(N, self.d, Q, K)
        if need == 0:
            raise _lib.ClipmiError("topk_ip: " + _lib.last_error())
        # the workspace belongs to the stream the call is enqueued on (torch's current stream): two batches in flight on
        # two streams each get their own, and the caching allocator hands a block back only to the stream that owned it
        skey = torch.cuda.current_stream(self.device).cuda_stream
        ws = self._ws.get(skey)
        if ws is None or ws.numel() < need:
            ws = self._ws[skey] = torch.empty(need, dtype=torch.uint8, device=self.device)
        if out is None:
            out_s = torch.empty((Q, K), dtype=torch.float32, device=self.device)
            out_i = torch.empty((Q, K), dtype=torch.int64, device=self.device)
        else:
            out_s, out_i = out
        if coarse and self.coarse == "int8":
            rc = L.clipmi_topk_ip_coarse_i8(db.data_ptr(), db8.data_ptr(), meta.data_ptr(), amax, N, self.d, rmax, q.data_ptr(),
                                            Q, K, self.id_base, out_s.data_ptr(), out_i.data_ptr(), ws.data_ptr(),
                                            ws.numel(), _lib.stream_ptr(self.device))
            _lib.check(rc, "clipmi_topk_ip_coarse_i8")
            return out_s, out_i
        if coarse:
            rc = L.clipmi_topk_ip_coarse(db.data_ptr(), dbh.data_ptr(), N, self.d, rmax, q.data_ptr(), Q, K, self.id_base,
                                         out_s.data_ptr(), out_i.data_ptr(), ws.data_ptr(), ws.numel(),
                                         _lib.stream_ptr(self.device))
            _lib.check(rc, "clipmi_topk_ip_coarse")
            return out_s, out_i
        rc = L.clipmi_topk_ip(db.data_ptr(), _lib.F32, N, self.d, q.data_ptr(), Q, K, self.id_base,
                              out_s.data_ptr(), out_i.data_ptr(), ws.data_ptr(), ws.numel(),
                              _lib.stream_ptr(self.device))
        _lib.check(rc, "clipmi_topk_ip")
        return out_s, out_i

    PASS_Q = 64                   # queries of one coarse pass (csrc/topk.hip COARSE_Q)
    WIDE_Q = 1024                 # queries of one wide pass of the int8 copy (csrc/topk.hip WIDE_MAX_Q)
    batches_in_flight = 2         # 64-query passes of ONE large search kept in flight on internal streams (1 = off)

    def _search_pipelined(self, q, K, out, chunk):
        """A search of more than 64 queries on the bf16 coarse path (`chunk` = 64), or of more than 1024 on the int8 path (`chunk`
        = 1024 = one wide pass): its passes alternate between the caller's stream and an
        internal HIP stream (each with its own workspace), so one pass's latency-bound side kernels run beside the other's
        HBM-bound scan -
        what bench.py measures as "two batches in flight" (0.97-1.03 vs 1.09-1.10 ms per pass at 10 M rows). Same calls,
        same results; the caller's stream waits for the side stream before anything after the search runs."""
        Q = q.shape[0]
        if out is None:
            out_s = torch.empty((Q, K), dtype=torch.float32, device=self.device)
            out_i = torch.empty((Q, K), dtype=torch.int64, device=self.device)
        else:
            out_s, out_i = out
        # lanes = the caller's stream + ONE side stream, the process-wide one that CLIP.encode_image's second kernel sequence uses
        # too (_lib.side_stream: this ROCm gives only the first three streams of a process their own hardware queue and every
        # later one shares the fourth - two side streams created after a caller's own two ran on ONE queue, back to back:
        # rocprofv3 Queue_Id; a side stream per model and index did the same to bench.py's search leg in round 5)
        cur, side1 = _lib.side_stream(self.device)
        side = [side1]
        lanes = [cur] + side
        for s_ in side:
            s_.wait_stream(cur)                       # q, the outputs and the index copies are ready
        for gi, lo in enumerate(range(0, Q, chunk)):
            hi = min(Q, lo + chunk)
            with torch.cuda.stream(lanes[gi % len(lanes)]):
                self.search_device(q[lo:hi], K, out=(out_s[lo:hi], out_i[lo:hi]), _one_pass=True)
        for s_ in side:
            cur.wait_stream(s_)
        return out_s, out_i

    def search(self, x, K):
        """D, I = index.search(features, K) (query-index.py:111): numpy in, numpy out."""
        q = torch.from_numpy(np.array(x, dtype=np.float32, order="C")) if not isinstance(x, torch.Tensor) else x
        if q.dim() != 2 or q.shape[1] != self.d:
            raise ValueError(f"search: expected [Q,{self.d}], got {tuple(q.shape)}")
        s, i = self.search_device(q, int(K))
        return s.cpu().numpy(), i.cpu().numpy()


def IndexIVFFlat(quantizer, d, nlist=100, metric=METRIC_INNER_PRODUCT):
    """faiss.IndexIVFFlat(quantizer, 512, 100, faiss.METRIC_INNER_PRODUCT) (build-index.py:81) — returns
    the EXACT flat index: the inverted lists exist to avoid a full scan, which this build performs
    at HBM speed instead. `nlist` is accepted and ignored; only inner product is supported."""
    if metric != METRIC_INNER_PRODUCT:
        raise ValueError("IndexIVFFlat: only METRIC_INNER_PRODUCT is supported")
    dev = getattr(quantizer, "device", "cuda:0")
    return IndexFlatIP(d, device=dev)


class ShardedFlatIP:
    """Rank-local shard + the one all-gather merge. One process per GPU.

    Each rank holds rows [lo, hi) of the global matrix (`shard_bounds`), searches them with
    global ids, all-gathers the [Q,K] partials and merges. On GPUs the collective is RCCL
    (`backend="nccl"`) over xGMI and the merge is clipmi_merge_topk; on the gloo/CPU path (tests
    of the host logic, no GPU) `local_search` must be supplied and the merge is merge_lists_host.
    """

    def __init__(self, local_index, n_total, group=None, local_search=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.local = local_index
        self.n_total = int(n_total)
        self.lo, self.hi = shard_bounds(n_total, self.world, self.rank)
        if local_index is not None:
            local_index.id_base = self.lo
        self._local_search = local_search
        self.nprobe = 1

    def _record(self, Q, K, device):
        """Per-rank packed record [scores f32 Q*K | pad | ids i64 Q*K] and the gather buffer, cached per (Q, K, stream)."""
        skey = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
        key = (Q, K, str(device), skey)
        cache = self.__dict__.setdefault("_rec_cache", {})
        ent = cache.get(key)
        if ent is None:
            ids_off = (Q * K * 4 + 7) // 8 * 8
            nbytes = ids_off + Q * K * 8
            rec = torch.empty(nbytes, dtype=torch.uint8, device=device)
            gath = torch.empty(self.world * nbytes, dtype=torch.uint8, device=device)
            ent = cache[key] = (rec, gath, rec[:Q * K * 4].view(torch.float32).view(Q, K), rec[ids_off:].view(torch.int64).view(Q, K))
        return ent

    def search_device(self, q, K, out=None):
        """-> (scores f32 [Q,K], ids i64 [Q,K]) of the WHOLE index, on every rank. `out` = (scores, ids) to write into;
        otherwise fresh tensors per call (a caller holding two results keeps both: the record and gather buffers are
        reused, the results are not)."""
        dist = self.dist
        Q = q.shape[0]
        if self._local_search is None:
            # GPU path: local top-K written straight into the packed record, ONE all-gather (RCCL),
            # merge kernel on the gathered buffer — three enqueues, no host round trip
            dev = self.local.device
            rec, gath, rec_s, rec_i = self._record(Q, K, dev)
            self.local.search_device(q, K, out=(rec_s, rec_i))
            dist.all_gather_into_tensor(gath, rec, group=self.group)
            L = _lib.lib()
            if out is None:
                out_s = torch.empty((Q, K), dtype=torch.float32, device=dev)
                out_i = torch.empty((Q, K), dtype=torch.int64, device=dev)
            else:
                out_s, out_i = out
            rc = L.clipmi_merge_topk_packed(gath.data_ptr(), rec.numel(), self.world, Q, K, out_s.data_ptr(),
                                            out_i.data_ptr(), _lib.stream_ptr(dev))
            _lib.check(rc, "clipmi_merge_topk_packed")
            return out_s, out_i
        # CPU / gloo path (tests of the host logic): same record, host merge
        s, i = self._local_search(q, K, self.lo)
        s = torch.as_tensor(np.ascontiguousarray(s, dtype=np.float32))
        i = torch.as_tensor(np.ascontiguousarray(i, dtype=np.int64))
        rec, gath, rec_s, rec_i = self._record(Q, K, torch.device("cpu"))
        rec_s.copy_(s)
        rec_i.copy_(i)
        dist.all_gather_into_tensor(gath, rec, group=self.group)
        n = rec.numel()
        ids_off = (Q * K * 4 + 7) // 8 * 8
        S = np.stack([gath[r * n:r * n + Q * K * 4].view(torch.float32).view(Q, K).numpy() for r in range(self.world)])
        I = np.stack([gath[r * n + ids_off:(r + 1) * n].view(torch.int64).view(Q, K).numpy() for r in range(self.world)])
        ms, mi = merge_lists_host(S, I, K)
        if out is not None:
            out[0].copy_(torch.from_numpy(ms))
            out[1].copy_(torch.from_numpy(mi))
            return out
        return torch.from_numpy(ms), torch.from_numpy(mi)

    def search(self, x, K):
        q = torch.from_numpy(np.array(x, dtype=np.float32, order="C")) if not isinstance(x, torch.Tensor) else x
        s, i = self.search_device(q, int(K))
        return s.cpu().numpy(), i.cpu().numpy()


FAISS_FOURCC_FLAT_IP = b"IxFI"


def write_index(index, path, format="clipmi"):
    """faiss.write_index stand-in (build-index.py:109).
    format="clipmi": own packed format, little-endian: 8-byte magic, u32 version, u32 d, u64 ntotal, then
        ntotal*d f32 rows.
    format="faiss": the serialisation of a faiss IndexFlatIP as published in faiss/impl/index_write.cpp
        (fourcc "IxFI"; header d:i32, ntotal:i64, two i64 placeholders = 1<<20, is_trained:u8,
        metric_type:i32 = 0; then u64 count of floats + the f32 rows), so that the ORIGINAL
        query-index.py:29 (`faiss.read_index("images.index")`) can open an index built here
        (SURVEY.md §8f next-3). faiss is not available offline: this layout is restated from the
        upstream source, PARITY UNPINNED until checked against a real faiss build."""
    db = np.ascontiguousarray(index.matrix().cpu().numpy(), dtype="<f4")
    with open(path, "wb") as f:
        if format == "faiss":
            f.write(FAISS_FOURCC_FLAT_IP)
            f.write(struct.pack("<iqqq", index.d, db.shape[0], 1 << 20, 1 << 20))
            f.write(struct.pack("<Bi", 1, METRIC_INNER_PRODUCT))
            f.write(struct.pack("<Q", db.size))
        elif format == "clipmi":
            f.write(MAGIC + struct.pack("<IIQ", 1, index.d, db.shape[0]))
        else:
            raise ValueError(f"write_index: unknown format {format!r}")
        f.write(db.tobytes())


def _read_faiss_header(f):
    """faiss write_index_header: d i32, ntotal i64, two dummy i64, is_trained u8, metric i32."""
    d, n, _, _ = struct.unpack("<iqqq", f.read(28))
    trained, metric = struct.unpack("<Bi", f.read(5))
    return d, n, metric


def _read_faiss_ivfflat(f, path):
    """The file the REFERENCE writes (build-index.py:80-81,109: IndexIVFFlat(IndexFlatIP(512), 512, 100,
    METRIC_INNER_PRODUCT) -> faiss.write_index): fourcc "IwFl", index header, nlist u64, nprobe u64, the
    quantizer as a nested index ("IxFI" + centroids), the direct map (u8 type, i64 vector[, hashtable]),
    then ArrayInvertedLists: fourcc "ilar", nlist u64, code_size u64, list sizes ("full": u64 vector of nlist
    sizes; "sprs": u64 vector of (list, size) pairs), then per non-empty list its codes (n * code_size bytes =
    n rows of d f32) and its ids (n i64). Layout as in faiss impl/index_write.cpp (write_ivf_header,
    write_direct_map, write_InvertedLists). Returns the rows in ID order: exact search over them returns a
    superset-quality answer of what the IVF index (nprobe of 100 lists) would.
    PARITY UNPINNED: no faiss build exists in this environment to produce or check a real file; the test
    fixture is written by tests/ from the same layout."""
    d, n, metric = _read_faiss_header(f)
    if metric != METRIC_INNER_PRODUCT:
        raise ValueError(f"{path}: faiss IVF index with metric {metric}; only inner product is supported")
    nlist, _nprobe = struct.unpack("<QQ", f.read(16))
    q4 = f.read(4)
    if q4 not in (b"IxFI", b"IxF2", b"IxFl"):
        raise ValueError(f"{path}: IVF quantizer {q4!r} is not a flat index")
    qd, qn, _ = _read_faiss_header(f)
    (cnt,) = struct.unpack("<Q", f.read(8))
    if qd != d or cnt != qn * qd:
        raise ValueError(f"{path}: malformed IVF quantizer")
    f.seek(cnt * 4, 1)                                   # centroids: not needed for exact search
    (dm_type,) = struct.unpack("<B", f.read(1))
    (dm_n,) = struct.unpack("<Q", f.read(8))
    f.seek(dm_n * 8, 1)
    if dm_type == 2:                                     # hashtable: vector of (key, value) pairs
        (hn,) = struct.unpack("<Q", f.read(8))
        f.seek(hn * 16, 1)
    if f.read(4) != b"ilar":
        raise ValueError(f"{path}: only ArrayInvertedLists ('ilar') are supported")
    il_nlist, code_size = struct.unpack("<QQ", f.read(16))
    if il_nlist != nlist or code_size != 4 * d:
        raise ValueError(f"{path}: inverted lists nlist {il_nlist} / code size {code_size} do not match IVFFlat d={d}")
    kind = f.read(4)
    (sz_n,) = struct.unpack("<Q", f.read(8))
    raw = np.fromfile(f, dtype="<u8", count=sz_n)
    sizes = np.zeros(nlist, dtype=np.int64)
    if kind == b"full":
        if sz_n != nlist:
            raise ValueError(f"{path}: 'full' list sizes vector has {sz_n} entries for {nlist} lists")
        sizes[:] = raw
    elif kind == b"sprs":
        sizes[raw[0::2].astype(np.int64)] = raw[1::2]
    else:
        raise ValueError(f"{path}: unknown inverted-list size encoding {kind!r}")
    if int(sizes.sum()) != n:
        raise ValueError(f"{path}: inverted lists hold {int(sizes.sum())} vectors, header says {n}")
    mat = np.empty((n, d), dtype=np.float32)
    seen = np.zeros(n, dtype=bool)
    for ln in sizes:
        ln = int(ln)
        if ln == 0:
            continue
        codes = np.fromfile(f, dtype="<f4", count=ln * d)
        ids = np.fromfile(f, dtype="<i8", count=ln)
        if codes.size != ln * d or ids.size != ln:
            raise ValueError(f"{path}: truncated inverted list")
        if ids.min() < 0 or ids.max() >= n or seen[ids].any():
            raise ValueError(f"{path}: ids are not a permutation of 0..{n - 1} (custom ids are not supported)")
        seen[ids] = True
        mat[ids] = codes.reshape(ln, d)
    return d, mat


def read_index(path, device="cuda:0", rows=None, coarse=None):
    """faiss.read_index stand-in (query-index.py:29): reads both formats write_index produces and the
    IndexIVFFlat file the reference's build-index.py writes (rows come back in id order and are searched
    exactly; `nprobe` is accepted and ignored).
    rows=(lo, hi): load only that contiguous row range (one rank's shard) — the flat formats
    seek straight to it; the returned index has `id_base = lo` and `n_file` = the file's row count."""
    with open(path, "rb") as f:
        head = f.read(8)
        if head[:4] == FAISS_FOURCC_FLAT_IP or head == MAGIC:
            if head == MAGIC:
                ver, d, n = struct.unpack("<IIQ", f.read(16))
                if ver != 1:
                    raise ValueError(f"{path}: unsupported version {ver}")
            else:
                f.seek(4)
                d, n, metric = _read_faiss_header(f)
                if metric != METRIC_INNER_PRODUCT:
                    raise ValueError(f"{path}: faiss flat index with metric {metric}; only inner product is supported")
                (count,) = struct.unpack("<Q", f.read(8))
                if count != n * d:
                    raise ValueError(f"{path}: vector count {count} != ntotal*d {n * d}")
            lo, hi = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
            if not (0 <= lo <= hi <= n):
                raise ValueError(f"{path}: rows {rows} outside 0..{n}")
            f.seek((lo * d) * 4, 1)
            data = np.fromfile(f, dtype="<f4", count=(hi - lo) * d)
        elif head[:4] == b"IwFl":
            f.seek(4)
            d, mat = _read_faiss_ivfflat(f, path)
            n = mat.shape[0]
            lo, hi = (0, n) if rows is None else (int(rows[0]), int(rows[1]))
            if not (0 <= lo <= hi <= n):
                raise ValueError(f"{path}: rows {rows} outside 0..{n}")
            data = np.ascontiguousarray(mat[lo:hi]).reshape(-1)
        else:
            raise ValueError(f"{path}: not a clipmi, faiss IndexFlatIP or faiss IndexIVFFlat file")
    if data.size != (hi - lo) * d:
        raise ValueError(f"{path}: truncated ({data.size} of {(hi - lo) * d} floats)")
    idx = IndexFlatIP(d, device=device, coarse=coarse)
    if hi > lo:
        idx.add(data.reshape(hi - lo, d))
    idx.id_base = lo
    idx.n_file = n
    return idx


def index_rows(path):
    """(ntotal, d) of an index file without loading its rows (flat formats: header only)."""
    with open(path, "rb") as f:
        head = f.read(8)
        if head == MAGIC:
            _, d, n = struct.unpack("<IIQ", f.read(16))
            return n, d
        if head[:4] in (FAISS_FOURCC_FLAT_IP, b"IwFl"):
            f.seek(4)
            d, n, _ = _read_faiss_header(f)
            return n, d
    raise ValueError(f"{path}: not a clipmi, faiss IndexFlatIP or faiss IndexIVFFlat file")
