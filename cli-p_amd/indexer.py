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
independent, so the encode path has no collective. After every round of one batch per rank, the ranks'
(keys, vectors, failures) are gathered to rank 0 over gloo (host-side ingest I/O, not the data path) and
committed in rank order; rank 0 alone assembles the matrix and writes `images.index`.

Weights: $CLIPMI_WEIGHTS (a local ViT-B-32.pt / state-dict), or CLIPMI_RANDOM_WEIGHTS=<seed> for a
synthetic model. Knobs: CLIPMI_BATCH (default 435 = whole rounds of GEMM tiles on 256 CUs),
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


def encode_directories(dirs, model, db, batch, workers, ranks=None, pool=None):
    """ranks=None or world 1: the single-GPU loop. Otherwise `db` is only used on rank 0 (others pass None).
    pool: a pipeline.DecodePool (decode in worker processes) or None (decode on `workers` threads)."""
    if ranks is None or ranks.world == 1:
        for base_path in dirs:
            print(f"CLIPing {base_path}...")
            todo = candidates(base_path, db)
            for ok, feats, bad in pipeline.encode_files(model, todo, batch=batch, workers=workers, pool=pool):
                if ok:
                    db.put_vectors(ok, feats)
                db.mark_skipped(bad)
                print("." * len(ok) + "#" * len(bad), end="", flush=True)
            print(flush=True)
        return
    for base_path in dirs:
        if ranks.leader:
            print(f"CLIPing {base_path}...")
        # the same sorted list on every rank (rank 0 is the only one that can see what is already stored)
        todo = ranks.bcast(sorted(candidates(base_path, db)) if ranks.leader else None)
        lo, hi = shard_bounds(len(todo), ranks.world, ranks.rank)
        mine = pipeline.encode_files(model, todo[lo:hi], batch=batch, workers=workers, pool=pool)
        # every rank walks the same number of rounds: the largest slice decides (slices differ by <= 1 file)
        rounds = (shard_bounds(len(todo), ranks.world, 0)[1] + batch - 1) // batch
        for _ in range(rounds):
            got = next(mine, ([], None, []))
            parts = ranks.gather(got)
            if ranks.leader:
                for ok, feats, bad in parts:           # rank order = sorted-list order: deterministic commits
                    if ok:
                        db.put_vectors(ok, feats)
                    db.mark_skipped(bad)
                    print("." * len(ok) + "#" * len(bad), end="", flush=True)
        if ranks.leader:
            print(flush=True)


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
        try:
            encode_directories(argv, model, db, int(os.environ.get("CLIPMI_BATCH", "435")), workers, ranks, pool)
        except KeyboardInterrupt:
            print("Interrupted!")
        if ranks.leader:
            finalise(db, device)
            print("Done!")
            db.close()
        ranks.close()
    finally:
        if pool is not None:
            pool.close()
