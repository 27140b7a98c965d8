import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    import torch
    return torch.cuda.is_available()


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must not silently pass: fail loudly instead of skipping.
    pass


class TopkOracle:
    """ctypes view of oracle/_build/libtopk_oracle.so (test infrastructure only)."""

    def __init__(self):
        so = os.path.join(ROOT, "oracle", "_build", "libtopk_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        self.lib = C.CDLL(so)
        self.lib.topk_oracle.restype = C.c_int
        self.lib.topk_oracle_mt.restype = C.c_int
        try:
            self.threads = max(1, min(16, len(os.sched_getaffinity(0))))
        except AttributeError:
            self.threads = 4
        self.lib.topk_oracle_merge.restype = C.c_int

    def topk(self, db, q, K, id_base=0):
        db = np.ascontiguousarray(db, dtype=np.float32)
        q = np.ascontiguousarray(q, dtype=np.float32)
        N, E = db.shape if db.ndim == 2 else (0, q.shape[1])
        Q = q.shape[0]
        out_s = np.empty((Q, K), dtype=np.float32)
        out_i = np.empty((Q, K), dtype=np.int64)
        # (the queries dealt over the host's cores: independent, so the same results as the one-thread entry point -
        # test_oracle_topk.py::test_threaded_oracle_equals_the_plain_one)
        rc = self.lib.topk_oracle_mt(db.ctypes.data_as(C.c_void_p), C.c_int64(N), C.c_int(E),
                                     q.ctypes.data_as(C.c_void_p), C.c_int(Q), C.c_int(K), C.c_int64(id_base),
                                     out_s.ctypes.data_as(C.c_void_p), out_i.ctypes.data_as(C.c_void_p), C.c_int(self.threads))
        assert rc == 0
        return out_s, out_i

    def topk_one_thread(self, db, q, K, id_base=0):
        db = np.ascontiguousarray(db, dtype=np.float32)
        q = np.ascontiguousarray(q, dtype=np.float32)
        N, E = db.shape if db.ndim == 2 else (0, q.shape[1])
        Q = q.shape[0]
        out_s = np.empty((Q, K), dtype=np.float32)
        out_i = np.empty((Q, K), dtype=np.int64)
        rc = self.lib.topk_oracle(db.ctypes.data_as(C.c_void_p), C.c_int64(N), C.c_int(E),
                                  q.ctypes.data_as(C.c_void_p), C.c_int(Q), C.c_int(K), C.c_int64(id_base),
                                  out_s.ctypes.data_as(C.c_void_p), out_i.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return out_s, out_i

    def scores(self, db, qrow):
        db = np.ascontiguousarray(db, dtype=np.float32)
        qrow = np.ascontiguousarray(qrow, dtype=np.float32)
        out = np.empty(db.shape[0], dtype=np.float32)
        self.lib.topk_oracle_scores(db.ctypes.data_as(C.c_void_p), C.c_int64(db.shape[0]), C.c_int(db.shape[1]),
                                    qrow.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
        return out

    def merge(self, scores, ids, K):
        scores = np.ascontiguousarray(scores, dtype=np.float32)
        ids = np.ascontiguousarray(ids, dtype=np.int64)
        R, Q, K_ = scores.shape
        assert K_ == K
        out_s = np.empty((Q, K), dtype=np.float32)
        out_i = np.empty((Q, K), dtype=np.int64)
        rc = self.lib.topk_oracle_merge(scores.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p),
                                        C.c_int(R), C.c_int(Q), C.c_int(K),
                                        out_s.ctypes.data_as(C.c_void_p), out_i.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return out_s, out_i


@pytest.fixture(scope="session")
def topk_oracle():
    return TopkOracle()


@pytest.fixture(scope="session")
def clipmi():
    import clipmi as m
    return m


@pytest.fixture(scope="session")
def gpu():
    """GPU tests call through the C ABI; there is no CPU fallback, so a missing GPU or a missing
    libclipmi.so is a failure, not a skip."""
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    import clipmi as m
    m._lib.lib()
    return torch.device("cuda:0")


def unit_rows(rng, n, d):
    x = rng.standard_normal((n, d), dtype=np.float32)
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)
