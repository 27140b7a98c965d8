"""Indexer behind `build-index.py DIR/ [DIR/ ...]` — the build side of the drop-in CLI.

Same command line, console output and on-disk schema as the reference's build-index.py (keys are
`DIR + filename` with no separator inserted, build-index.py:31 — pass directories with a trailing
slash; .jpg/.jpeg/.png only, build-index.py:32-34; already-indexed and previously-failed files are
skipped, :36-44; a failing file prints '#' and is recorded, :55-61; Ctrl-C still finalises the index,
:63-64). What differs: images are encoded in batches by the HIP kernels, the store commits once per
batch, and the index is an exact flat inner-product matrix instead of a trained IVF file.

Weights: $CLIPMI_WEIGHTS (a local ViT-B-32.pt / state-dict), or CLIPMI_RANDOM_WEIGHTS=<seed> for a
synthetic model. Knobs: CLIPMI_BATCH (default 435 = whole rounds of GEMM tiles on 256 CUs),
CLIPMI_WORKERS (decode threads, default 8).
"""
import os

from . import pipeline, store as vstore
from .index import IndexFlatIP, write_index
from .model import load

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


def encode_directories(dirs, model, db, batch, workers):
    for base_path in dirs:
        print(f"CLIPing {base_path}...")
        todo = candidates(base_path, db)
        for ok, feats, bad in pipeline.encode_files(model, todo, batch=batch, workers=workers):
            if ok:
                db.put_vectors(ok, feats)
            db.mark_skipped(bad)
            print("." * len(ok) + "#" * len(bad), end="", flush=True)
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


def main(argv):
    device = "cuda:0"
    model, _ = load(os.environ.get("CLIPMI_WEIGHTS", "ViT-B/32"), device=device, jit=False)
    model.eval()
    db = vstore.VectorStore("vectors.lmdb", dim=model.embed_dim)
    try:
        encode_directories(argv, model, db, int(os.environ.get("CLIPMI_BATCH", "435")),
                           int(os.environ.get("CLIPMI_WORKERS", "8")))
    except KeyboardInterrupt:
        print("Interrupted!")
    finalise(db, device)
    print("Done!")
    db.close()

