"""Generates tests/golden/topk_ties.npz with oracle/topk_oracle.c (run from the repo root):
    python tests/golden/make_topk_golden.py
The reference holds no golden vectors for this path, so this fixture pins the
oracle's own output (and a float64 re-derivation of the ranking, checked in tests/test_oracle.py)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, HERE)
from conftest import TopkOracle  # noqa: E402
import topk_case  # noqa: E402

db, q = topk_case.build()
D, I = TopkOracle().topk(db, q, topk_case.K)
np.savez_compressed(os.path.join(HERE, "topk_ties.npz"), D=D, I=I, q=q,
                    db_checksum=np.float64(db.astype(np.float64).sum()))
print("wrote topk_ties.npz", D[0, :4], I[0, :4])
