"""cli-p_amd — MI355X (gfx950) native CLIP index-and-search hot path.

Drop-in for the calls the reference scripts make on `clip` / `faiss` objects:
    model, transform = load("ViT-B/32", device)       # build-index.py:18, query-index.py:21
    model.encode_image(x) / model.encode_text(ids)     # build-index.py:49, query-index.py:108
    index.search(features, K)                          # query-index.py:111
All arithmetic runs in hand-written HIP kernels inside libclipmi.so (C ABI: include/clipmi.h),
reached through ctypes; torch is used for device memory, streams and torch.distributed only.
The directory name carries a hyphen, so import it with importlib.import_module("cli-p_amd")
or through the `clipmi` shim module at the repo root.
"""
from . import build
from . import _lib
from ._lib import ClipmiError
from .index import (IndexFlatIP, IndexIVFFlat, ShardedFlatIP, METRIC_INNER_PRODUCT, read_index, write_index,
                    shard_bounds, merge_lists_host)

__all__ = ["_lib", "ClipmiError", "IndexFlatIP", "ShardedFlatIP", "METRIC_INNER_PRODUCT",
           "read_index", "write_index", "shard_bounds", "merge_lists_host"]
from . import weights  # noqa: E402
from . import model  # noqa: E402
from .model import CLIP, load, make_transform, available_models  # noqa: E402
from . import tokenizer, store, pipeline, resize, jpeg, jpeg_parse  # noqa: E402
from .tokenizer import tokenize  # noqa: E402
from . import ranks  # noqa: E402
from . import indexer, repl  # noqa: E402
