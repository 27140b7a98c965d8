"""Interactive prompt behind `query-index.py` — the query side of the drop-in CLI.

Same prompt and commands as the reference's query-index.py (q, h, i ID, r WxH, a, c NUM, p NUM, empty
line = more results; result lines "score id path"; the best hit is dropped and K = k + offset + 1,
query-index.py:111,115-116). Text is encoded by the HIP text tower and searched EXACTLY over the flat
matrix; `p NUM` is accepted and has no effect. Images are shown only if OpenCV is installed.

Multi-GPU (the reference is single-process, query-index.py:20): launched as
`python -m torch.distributed.run --nproc-per-node N query-index.py`, every rank loads rows
`shard_bounds(ntotal, world, rank)` of images.index into its own GPU. Rank 0 owns the terminal, the store and
the text tower and runs the unchanged prompt loop over a `LeaderIndex`: each search first broadcasts
(features, K) to the followers (a 2-KB host object over gloo), then all ranks run `ShardedFlatIP.search` —
per-shard exact top-K, ONE all-gather (RCCL), the same merge everywhere. Followers sit in `follow()`.

Weights as for the indexer; the BPE merge table comes from $CLIPMI_BPE_PATH.
"""
import os
import sys
import time

import numpy as np

from . import store as vstore, tokenizer
from .index import ShardedFlatIP, index_rows, read_index, shard_bounds
from .model import load
from .ranks import Ranks

HELP = ("Enter a search query and you will receive a list of best matching\nimages. The first number is the "
        "difference score, the second the\nimage ID followed by the filename.\n\nPress q to stop viewing image "
        "and space for the next image.\n\nJust press enter for more results.\n\nCommands:\nq\tQuit\n"
        "i ID\tFind images similar to ID\nr [RES]\tSet maximum resolution (e.g. 1280x720)\n"
        "a\tToggle align window position\nc NUM\tSet default number of results to NUM\n"
        "p NUM\tSet number of subsets to probe (1-100, 32 default)\nh\tShow this help")


def normalize(v):
    n = np.linalg.norm(v)
    return v if n < 0.000000001 else v / n


class Viewer:
    """Optional OpenCV window (query-index.py:120-151); a no-op without cv2."""

    def __init__(self):
        try:
            import cv2
            self.cv2 = cv2
        except ImportError:
            self.cv2 = None
        self.max_res = None
        self.align = False

    def show(self, path):
        """Returns False when the user pressed q (stop showing this result list)."""
        cv2 = self.cv2
        if cv2 is None:
            return True
        img = cv2.imread(path, cv2.IMREAD_COLOR)
        if img is None or img.shape[0] < 2:
            return True
        h, w = img.shape[:2]
        if self.max_res is not None:
            scale = min(1.0, self.max_res[0] / w, self.max_res[1] / h)
            if scale < 1.0:
                img = cv2.resize(img, (int(w * scale + 0.5), int(h * scale + 0.5)), interpolation=cv2.INTER_LANCZOS4)
        cv2.imshow("Image", img)
        if self.align:
            cv2.moveWindow("Image", 0, 0)
        while True:
            key = cv2.waitKey(0) & 0xFF
            if key == ord(" "):
                return True
            if key == ord("q"):
                return False

    def close(self):
        if self.cv2 is not None:
            self.cv2.destroyAllWindows()


def repl(model, index, db, inp=input, out=print):
    k, offset, last_j = 50, 0, 0
    features, have_text = None, False
    viewer = Viewer()
    while True:
        line = inp("[h,q,i,r,a,c,p] >>> ").strip()
        if line == "q":
            break
        if line == "h":
            out(HELP)
            continue
        if line.startswith("p "):
            probe = int(line[2:])
            if 0 < probe < 101:
                index.nprobe = probe
                out(f"Set to probe {probe} subsets.")
            else:
                out("Invalid probe value.")
            continue
        if line == "a":
            viewer.align = not viewer.align
            out("Aligning window position." if viewer.align else "Not aligning window position.")
            continue
        if line.startswith("r "):
            try:
                x, y = (int(t) for t in line[2:].split("x"))
                if x > 0 and y > 0:
                    viewer.max_res = (x, y)
                    out(f"Set maximum resolution to {x}x{y}.")
                    continue
            except ValueError:
                pass
            viewer.max_res = None
            out("Unset maximum resolution.")
            continue
        if line.startswith("c "):
            k = int(line[2:])
            if k < 1:
                k = 50
                out("Reset number of results to 50.")
            else:
                out(f"Showing {k} results.")
            continue
        if line.startswith("i "):
            offset = last_j = 0
            path = db.idx_get(int(line[2:]))
            vec = db.get_vector(path) if path is not None else None
            if vec is None:
                out("Not found.")
                continue
            features = vec
            out(f"Similar to {path.decode()}:")
        elif line == "":
            offset = last_j
            if not have_text:
                continue
        else:
            offset = last_j = 0
            try:
                tokens = tokenizer.tokenize([line], context_length=model.context_length)
            except (RuntimeError, FileNotFoundError) as e:
                out(str(e))
                continue
            have_text = True
            features = normalize(model.encode_text(tokens).cpu().numpy().astype("float32"))

        t0 = time.perf_counter()
        D, I = index.search(features, k + offset + 1)
        out(f"Search time: {time.perf_counter() - t0:.4f}s")
        for j, i in enumerate(I[0]):
            if j <= offset or i < 0:
                continue
            path = db.idx_get(i).decode()
            out(f"{D[0][j]:.4f} {i} {path}")
            last_j = j
            if not viewer.show(path):
                break
        viewer.close()


class LeaderIndex:
    """What rank 0's prompt loop sees as `index` when the rows are sharded over the ranks: `.search` tells the
    followers what to search for, then takes part in the collective search itself."""

    def __init__(self, sharded, ranks):
        self.sharded, self.ranks = sharded, ranks
        self.nprobe = 1                      # accepted and ignored, as on the flat index

    def search(self, features, K):
        f = np.ascontiguousarray(features, dtype=np.float32)
        self.ranks.bcast(("search", f, int(K)))
        return self.sharded.search(f, int(K))

    def quit(self):
        self.ranks.bcast(("quit", None, 0))


def follow(sharded, ranks):
    """Ranks > 0: wait for rank 0's next search, join it, until told to quit."""
    while True:
        what, f, K = ranks.bcast(None)
        if what == "quit":
            return
        sharded.search(f, K)


def open_sharded(path, ranks, coarse=None, local_search=None):
    """This rank's shard of the index file + the all-gather merge around it. `local_search(q, K, lo)` replaces the
    GPU search on the CPU/gloo path (tests of the host logic; the product path has no CPU search)."""
    n, d = index_rows(path)
    lo, hi = shard_bounds(n, ranks.world, ranks.rank)
    if local_search is not None:
        return ShardedFlatIP(None, n, group=ranks.data, local_search=local_search)
    local = read_index(path, device=str(ranks.device), rows=(lo, hi))
    if coarse in ("int8", "bf16") and local.d == 512 and local.ntotal >= 65536:
        local.coarse = coarse
    return ShardedFlatIP(local, n, group=ranks.data)


def main():
    ranks = Ranks("cuda").init()
    device = str(ranks.device)
    # large libraries: scan a coarse copy first (identical results; CLIPMI_COARSE = int8 | bf16 | none)
    coarse = os.environ.get("CLIPMI_COARSE", "int8")
    if ranks.world > 1:
        sharded = open_sharded("images.index", ranks, coarse=coarse)
        if not ranks.leader:
            follow(sharded, ranks)
            ranks.close()
            sys.exit(0)
        index = LeaderIndex(sharded, ranks)
    else:
        index = read_index("images.index", device=device)
        if coarse in ("int8", "bf16") and index.d == 512 and index.ntotal >= 65536:
            index.coarse = coarse
    model, _ = load(os.environ.get("CLIPMI_WEIGHTS", "ViT-B/32"), device=device, jit=False)
    model.eval()
    db = vstore.VectorStore("vectors.lmdb", dim=model.embed_dim)
    index.nprobe = 32
    try:
        repl(model, index, db)
    except (EOFError, KeyboardInterrupt):
        print("Interrupted.")
    if ranks.world > 1:
        index.quit()
        ranks.close()
    db.close()
    sys.exit(0)
