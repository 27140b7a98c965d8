"""CPU checks of oracle/topk_oracle.c (the checker itself): against the committed golden
vectors, against an independent float64 ranking, and of its stated score order."""
import os
import sys

import numpy as np

from conftest import unit_rows

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import topk_case  # noqa: E402


def test_oracle_reproduces_golden(topk_oracle):
    g = np.load(os.path.join(GOLD, "topk_ties.npz"))
    db, q = topk_case.build()
    assert float(db.astype(np.float64).sum()) == float(g["db_checksum"])
    D, I = topk_oracle.topk(db, q, topk_case.K)
    assert np.array_equal(I, g["I"]) and np.array_equal(D.view(np.uint32), g["D"].view(np.uint32))
    # planted structure: duplicates of the best hit come back as a tie group in ascending id order
    assert list(I[0, :3]) == sorted(I[0, :3]) and D[0, 0] == D[0, 1] == D[0, 2]
    assert {100, 4000}.issubset(set(I[0, :3]))
    assert list(I[1, :2]) == [17, 18]


def test_oracle_score_order_is_the_documented_fmaf_chain(topk_oracle):
    rng = np.random.default_rng(3)
    db = unit_rows(rng, 8, 512)
    q = unit_rows(rng, 1, 512)[0]
    got = topk_oracle.scores(db, q)
    # restate with exact rational arithmetic per step: fmaf == round_f32(a*b + acc) in float64?
    # a*b of two f32 is exact in f64 (48-bit product); adding an f32 acc may round in f64, so use
    # Python fractions for the single rounding.
    from fractions import Fraction
    for r in range(db.shape[0]):
        acc = np.float32(0)
        for t in range(32):
            for c in range(4):
                for g in range(4):
                    k = 16 * t + 4 * g + c
                    exact = Fraction(float(db[r, k])) * Fraction(float(q[k])) + Fraction(float(acc))
                    acc = np.float32(_round_to_f32(exact))
        assert acc.view(np.uint32) == got[r].view(np.uint32)


def _round_to_f32(fr):
    # round-to-nearest-even of an exact rational to binary32, via float64 candidates
    from fractions import Fraction
    d = float(fr)                      # correctly rounded to f64
    f = np.float32(d)
    # double rounding guard: compare neighbours exactly
    cands = [f, np.nextafter(f, np.float32(np.inf)), np.nextafter(f, np.float32(-np.inf))]
    best = min(cands, key=lambda c: (abs(Fraction(float(c)) - fr), int(np.float32(c).view(np.uint32)) & 1))
    return best


def test_oracle_ranking_agrees_with_float64(topk_oracle):
    rng = np.random.default_rng(4)
    db = unit_rows(rng, 3000, 512)
    q = unit_rows(rng, 4, 512)
    D, I = topk_oracle.topk(db, q, 20)
    ref = db.astype(np.float64) @ q.astype(np.float64).T
    for j in range(4):
        want = np.argsort(-ref[:, j], kind="stable")[:20]
        # f32 accumulation can swap near-ties: demand set agreement on all but the boundary
        assert len(set(want[:15]) - set(I[j])) == 0
        assert np.allclose(D[j], ref[I[j], j], atol=2e-6)
        assert np.all(np.diff(D[j]) <= 0)


def test_oracle_padding_and_merge(topk_oracle, clipmi):
    rng = np.random.default_rng(5)
    db = unit_rows(rng, 40, 512)
    q = unit_rows(rng, 2, 512)
    D, I = topk_oracle.topk(db, q, 51, id_base=7)
    assert (I[:, 40:] == -1).all() and (I[:, :40] >= 7).all()
    # sharded + merge == whole (oracle merge and the host merge used on the gloo path)
    K = 10
    parts = [topk_oracle.topk(db[lo:hi], q, K, id_base=lo) for lo, hi in
             (clipmi.shard_bounds(40, 3, r) for r in range(3))]
    S = np.stack([p[0] for p in parts])
    Ii = np.stack([p[1] for p in parts])
    Dw, Iw = topk_oracle.topk(db, q, K)
    Dm, Im = topk_oracle.merge(S, Ii, K)
    assert np.array_equal(Im, Iw) and np.array_equal(Dm, Dw)
    Dh, Ih = clipmi.merge_lists_host(S, Ii, K)
    assert np.array_equal(Ih, Iw) and np.array_equal(Dh, Dw)


def test_threaded_oracle_equals_the_plain_one(topk_oracle):
    """The suites check through topk_oracle_mt (the queries dealt over the host's cores); queries are independent, so its
    results are those of the one-thread entry point, NaN rows, ties and padding included."""
    rng = np.random.default_rng(9)
    db = rng.standard_normal((5000, 64)).astype(np.float32)
    db[17] = db[4711]
    db[100, 3] = np.nan
    q = rng.standard_normal((37, 64)).astype(np.float32)
    a = topk_oracle.topk(db, q, 51, id_base=1000)
    b = topk_oracle.topk_one_thread(db, q, 51, id_base=1000)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
    a = topk_oracle.topk(db[:20], q[:3], 51)
    b = topk_oracle.topk_one_thread(db[:20], q[:3], 51)
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
