"""Indexer behind `build-index.py DIR/ [DIR/ ...]` — the build side of the drop-in CLI.

Same command line, console output and on-disk schema as the reference's build-index.py (keys are
`DIR + filename` with no separator inserted, build-index.py:31 — pass directories with a trailing
slash; .jpg/.jpeg/.png only, build-index.py:32-34; already-indexed and previously-failed files are
skipped, :36-44; a failing file prints '#' and is recorded, :55-61; Ctrl-C still finalises the index,
:63-64). What differs: images are encoded in batches by the HIP kernels, the store commits once per
batch, and the index is an exact flat inner-product matrix instead of a trained IVF file.

Multi-GPU (the reference is single-device, build-index.py:17): launched as
`python -m torch.distributed.run --nproc-per-node N build-index.py DIR/ ...`, rank 0 owns the store. Per
directory it lists the files still to encode, SORTS them and broadcasts the list; every rank takes the
contiguous slice `shard_bounds(len(todo), world, rank)` and encodes it on its own GPU — images are
independent, so the encode path has no collective. Every rank commits its batches to its OWN shard of the
store (`vectors.lmdb.shard-<rank>`, the packed append-only form); a round of one batch per rank ends with one
small gloo all-gather of three integers per rank (files done, files failed, "I was asked to stop") — progress
for rank 0's console and the agreement that lets every rank leave the loop in the SAME round when Ctrl-C
reached any of them. After the loop rank 0 ingests the shards in rank order (= sorted-list order) into the
store, assembles the matrix and writes `images.index` ("each rank emits its rows ... rank-0
ingest of shard outputs"). Shards left behind by a run that died are ingested at the next start.

Weights: $CLIPMI_WEIGHTS (a local ViT-B-32.pt / state-dict), or CLIPMI_RANDOM_WEIGHTS=<seed> for a
synthetic model. Knobs: CLIPMI_BATCH (default 870 = two kernel sequences of 435 images, whole rounds of GEMM tiles on 256 CUs, and a
JPEG decode launch whose serial chains are paid once per 870 files),
CLIPMI_DEVICE_JPEG_KB (default 8192: baseline JPEG files of up to that size are decoded on the device - csrc/jpeg.hip, Pillow's bytes -
instead of by Pillow in the workers; 0 = off),
CLIPMI_WORKERS (decode workers per rank, default min(16, CPUs / ranks on the node)), CLIPMI_DECODE (`procs`, the default: worker processes started before
the GPU is touched — 23 k images/s end to end from 224 x 224 JPEGs on 16 workers against 4 k on threads, which the GIL
binds; `threads`: the old form).
"""
import os

from . import pipeline, store as vstore
from .index import IndexFlatIP, shard_bounds, write_index
from .model import load
from .ranks import Ranks

EXTS = (".jpg", ".jpeg", ".png")


def candidates(base_path, db):
    """Files of one directory that still need encoding, as store keys (base_path + name)."""
    todo = []
    for name in os.listdir(base_path):
        if os.path.splitext(name)[1].lower() not in EXTS:
            continue
        key = base_path + name
        if db.is_skipped(key) or db.has_vector(key):
            continue
        todo.append(key)
    return todo


class StopFlag:
    """Ctrl-C under N ranks: SIGINT sets a flag instead of raising inside whatever call it lands in (a collective, the
    encoder); the build loop reads the flag once per round and the ranks agree on the round to stop in. Main thread only
    (signal handlers cannot be installed elsewhere); elsewhere the flag can still be set by hand."""

    def __init__(self):
        self.set = False
        self._old = None

    def _handler(self, signum, frame):
        self.set = True

    def __enter__(self):
        import signal
        import threading
        if threading.current_thread() is threading.main_thread():
            self._old = signal.signal(signal.SIGINT, self._handler)
        return self

    def __exit__(self, *a):
        import signal
        if self._old is not None:
            signal.signal(signal.SIGINT, self._old)
            self._old = None


def shard_path(store_path, rank):
    return f"{store_path}.shard-{int(rank)}"


def ingest_shards(db, store_path):
    """Rank 0: move every shard store beside `store_path` into `db`, in rank order, then remove it. Returns the number of
    vectors ingested. Also run at start-up: a shard that a dead run left behind holds finished work."""
    import glob
    import shutil
    n = 0
    found = []
    for p in glob.glob(glob.escape(store_path) + ".shard-*"):
        try:
            found.append((int(p.rsplit("-", 1)[1]), p))
        except ValueError:
            pass
    for _, p in sorted(found):
        sh = vstore.VectorStore(p, dim=db.dim, backend="packed")
        try:
            keys, rows = [], []
            for k, v in sh.b.items_sorted("fn_db"):
                keys.append(k.decode("utf-8", "surrogateescape"))
                rows.append(v)
                if len(keys) >= 4096:
                    db.b.put_many("fn_db", [(k_.encode("utf-8", "surrogateescape"), r_) for k_, r_ in zip(keys, rows)])
                    n += len(keys)
                    keys, rows = [], []
            if keys:
                db.b.put_many("fn_db", [(k_.encode("utf-8", "surrogateescape"), r_) for k_, r_ in zip(keys, rows)])
                n += len(keys)
            skipped = [kv for kv in sh.b.items_sorted("skip_db")]
            if skipped:
                db.b.put_many("skip_db", skipped)
        finally:
            sh.close()
        shutil.rmtree(p, ignore_errors=True)
    return n


class LostLog:
    """Files a decode worker DIED on, counted across runs in `<store>.lost` (one hex-encoded path per line, appended by
    whichever rank lost it: O_APPEND keeps short lines whole). A worker's death proves nothing about the file the first time
    (an external kill, an out-of-memory kill of a neighbour): the file stays a candidate. The second death on the SAME file
    does: it then goes to skip_db like a file the decoder rejected - otherwise a file that deterministically crashes its
    decoder (or a decompression bomb that gets the worker killed) would cost every later run one worker."""

    def __init__(self, store_path):
        self.path = str(store_path).rstrip("/") + ".lost"
        self.counts = {}
        try:
            with open(self.path, "rb") as f:
                for ln in f:
                    try:
                        k = bytes.fromhex(ln.strip().decode("ascii")).decode("utf-8", "surrogateescape")
                    except ValueError:
                        continue
                    self.counts[k] = self.counts.get(k, 0) + 1
        except OSError:
            pass

    def record(self, key):
        """Note one more worker death on `key`; returns how many there have been (this one included)."""
        self.counts[key] = self.counts.get(key, 0) + 1
        try:
            fd = os.open(self.path, os.O_WRONLY | os.O_APPEND | os.O_CREAT, 0o644)
            try:
                os.write(fd, key.encode("utf-8", "surrogateescape").hex().encode("ascii") + b"\n")
            finally:
                os.close(fd)
        except OSError:
            pass                      # a read-only directory: the count still holds for this run
        return self.counts[key]


def _undecodable(bad, pool, lost_log=None):
    """The failed files that may go to skip_db: those a decoder REJECTED, and those that have now cost a decode worker its
    life TWICE (LostLog). A file whose worker died under it for the first time (killed, out of memory) failed for this
    run only and is retried by the next one. (The dead worker's slot decodes in-process for the rest of the run: no program
    is started once the GPU may be initialised - pipeline.DecodePool.)"""
    lost = getattr(pool, "lost", None)
    if not lost:
        return bad
    out = []
    for b in bad:
        if b not in lost:
            out.append(b)
        elif lost_log is not None and b not in getattr(pool, "_lost_counted", set()):
            pool.__dict__.setdefault("_lost_counted", set()).add(b)
            if lost_log.record(b) >= 2:
                out.append(b)
    return out


def encode_directories(dirs, model, db, batch, workers, ranks=None, pool=None, stop=None, store_path="vectors.lmdb"):
    """ranks=None or world 1: the single-GPU loop. Otherwise `db` is only used on rank 0 (others pass None), every rank
    writes `shard_path(store_path, rank)` and rank 0 ingests the shards at the end; returns True when the ranks agreed to
    stop early (`stop`: a StopFlag). pool: a pipeline.DecodePool (decode in worker processes) or None (decode on `workers`
    threads)."""
    lost_log = LostLog(store_path) if pool is not None else None
    if ranks is None or ranks.world == 1:
        for base_path in dirs:
            print(f"CLIPing {base_path}...")
            todo = candidates(base_path, db)
            for ok, feats, bad in pipeline.encode_files(model, todo, batch=batch, workers=workers, pool=pool):
                if ok:
                    db.put_vectors(ok, feats)
                db.mark_skipped(_undecodable(bad, pool, lost_log))
                print("." * len(ok) + "#" * len(bad), end="", flush=True)
            print(flush=True)
        return False
    if ranks.leader:
        ingest_shards(db, store_path)                 # finished work of a run that died before its ingest
    ranks.barrier()
    shard = vstore.VectorStore(shard_path(store_path, ranks.rank), dim=model.embed_dim, backend="packed")
    stopped = False
    seen = set()
    try:
        for base_path in dirs:
            if ranks.leader:
                print(f"CLIPing {base_path}...")
            # the same sorted list on every rank (rank 0 is the only one that can see what is already stored)
            todo = ranks.bcast(sorted(k for k in candidates(base_path, db) if k not in seen) if ranks.leader else None)
            seen.update(todo)
            lo, hi = shard_bounds(len(todo), ranks.world, ranks.rank)
            mine = pipeline.encode_files(model, todo[lo:hi], batch=batch, workers=workers, pool=pool)
            # every rank walks the same number of rounds: the largest slice decides (slices differ by <= 1 file)
            rounds = (shard_bounds(len(todo), ranks.world, 0)[1] + batch - 1) // batch
            for _ in range(rounds):
                ok, feats, bad = [], None, []
                if not (stop is not None and stop.set):
                    try:
                        ok, feats, bad = next(mine, ([], None, []))
                    except KeyboardInterrupt:          # no StopFlag handler installed (not the main thread): same meaning
                        if stop is not None:
                            stop.set = True
                        ok, feats, bad = [], None, []
                    if stop is not None and stop.set:
                        # the signal arrived while this round was being decoded / encoded: nothing of it is committed
                        # (a terminal's Ctrl-C goes to the whole process group - a decode worker it killed would report
                        # its current file as failed); the files stay candidates for the next run
                        ok, feats, bad = [], None, []
                if ok:
                    shard.put_vectors(ok, feats)       # this rank's own shard: one commit per batch
                shard.mark_skipped(_undecodable(bad, pool, lost_log))
                flags = ranks.all_gather_ints([len(ok), len(bad), 1 if (stop is not None and stop.set) else 0])
                if ranks.leader:
                    print("".join("." * f[0] + "#" * f[1] for f in flags), end="", flush=True)   # rank order = list order
                if any(f[2] for f in flags):
                    stopped = True                     # every rank sees the same flags: all leave in this round
                    break
            mine.close()
            if ranks.leader:
                print(flush=True)
            if stopped:
                break
    finally:
        shard.close()
    ranks.barrier()                                    # every shard is closed and on disk
    if ranks.leader:
        ingest_shards(db, store_path)
    return stopped


def finalise(db, device, out="images.index"):
    n = db.count()
    if n == 0:
        return
    print(f"Preparing index for {n} entries...")
    print(f"Generating {(n, db.dim)} matrix...")
    matrix, _ = db.assemble()
    index = IndexFlatIP(db.dim, device=device)
    print("Adding to index...")
    index.add(matrix)
    print("Saving index...")
    write_index(index, out)


def default_workers():
    """Decode workers per rank when CLIPMI_WORKERS is not set: the CPUs this process may run on, shared between the
    ranks of the node, at most 16 (16 workers: 23 k images/s from 224 x 224 JPEGs, 8: 14 k)."""
    try:
        cpus = len(os.sched_getaffinity(0))
    except AttributeError:
        cpus = os.cpu_count() or 8
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", "1")))
    return max(1, min(16, cpus // local_world))


def main(argv):
    workers = int(os.environ.get("CLIPMI_WORKERS", "0")) or default_workers()
    # decode workers first: they are child programs, and nothing in this process has touched the GPU yet
    pool = None
    if os.environ.get("CLIPMI_DECODE", "procs") == "procs":
        try:
            pool = pipeline.DecodePool(workers)
        except OSError as e:                          # no room for child processes: decode on threads, as before
            print(f"(decode workers unavailable: {e}; decoding on {workers} threads)")
    try:
        ranks = Ranks("cuda").init()
        device = str(ranks.device)
        model, _ = load(os.environ.get("CLIPMI_WEIGHTS", "ViT-B/32"), device=device, jit=False)
        model.eval()
        db = vstore.VectorStore("vectors.lmdb", dim=model.embed_dim) if ranks.leader else None
        batch = int(os.environ.get("CLIPMI_BATCH", "870"))
        if ranks.world > 1:
            # Ctrl-C reaches every rank of the launcher's process group at its own moment: the flag + the loop's per-round
            # agreement make all of them stop in the same round; rank 0 still finalises (build-index.py:63-64)
            with StopFlag() as stop:
                interrupted = encode_directories(argv, model, db, batch, workers, ranks, pool, stop=stop)
        else:
            interrupted = False
            try:
                encode_directories(argv, model, db, batch, workers, ranks, pool)
            except KeyboardInterrupt:
                interrupted = True
        if interrupted and ranks.leader:
            print("Interrupted!")
        if ranks.leader:
            finalise(db, device)
            print("Done!")
            db.close()
        ranks.close()
    finally:
        if pool is not None:
            pool.close()
