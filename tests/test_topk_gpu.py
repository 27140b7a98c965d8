"""-m gpu parity of clipmi_topk_ip / clipmi_merge_topk (through the C ABI) against
oracle/topk_oracle.c: scores AND ids bit-exact (integer/index work)."""
import numpy as np
import pytest
import torch

from conftest import unit_rows

pytestmark = pytest.mark.gpu


def _run(clipmi, gpu, db, q, K, id_base=0):
    idx = clipmi.IndexFlatIP(db.shape[1] if db.ndim == 2 else q.shape[1], device=gpu)
    if db.shape[0]:
        idx.add(db)
    idx.id_base = id_base
    D, I = idx.search(q, K)
    return D, I


def _assert_exact(D, I, Ds, Is, tag):
    bad = np.nonzero((I != Is) | (D.view(np.uint32) != Ds.view(np.uint32)))
    assert bad[0].size == 0, (f"{tag}: {bad[0].size} mismatching slots, first at q={bad[0][0]} k={bad[1][0]}: "
                              f"got ({D[bad[0][0], bad[1][0]]!r}, {I[bad[0][0], bad[1][0]]}) "
                              f"want ({Ds[bad[0][0], bad[1][0]]!r}, {Is[bad[0][0], bad[1][0]]})")


@pytest.mark.parametrize("N,Q,K", [(1, 1, 1), (15, 1, 5), (16, 3, 16), (17, 16, 51), (1000, 5, 51),
                                   (4096, 3, 51), (4099, 16, 11), (20000, 17, 51), (70001, 16, 51),
                                   (70001, 1, 101), (131072, 33, 51), (5000, 2, 300)])
def test_topk_matches_oracle(clipmi, gpu, topk_oracle, N, Q, K):
    rng = np.random.default_rng(N * 131 + Q * 7 + K)
    db = unit_rows(rng, N, 512)
    q = unit_rows(rng, Q, 512)
    D, I = _run(clipmi, gpu, db, q, K, id_base=1000)
    Ds, Is = topk_oracle.topk(db, q, K, id_base=1000)
    _assert_exact(D, I, Ds, Is, f"N={N} Q={Q} K={K}")


def test_topk_golden_ties_and_duplicates(clipmi, gpu, topk_oracle):
    """SURVEY.md §8c fixture (iv): planted duplicate rows (identical vectors = duplicate photos)
    and an exact tie group; ties must resolve by ascending id."""
    import os, sys
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, gold)
    import topk_case
    g = np.load(os.path.join(gold, "topk_ties.npz"))
    db, q = topk_case.build()
    assert np.array_equal(q, g["q"]) and float(db.astype(np.float64).sum()) == float(g["db_checksum"])
    D, I = _run(clipmi, gpu, db, q, topk_case.K)
    _assert_exact(D, I, g["D"], g["I"], "golden ties")


def test_topk_fewer_rows_than_k(clipmi, gpu, topk_oracle):
    rng = np.random.default_rng(5)
    db = unit_rows(rng, 7, 512)
    q = unit_rows(rng, 2, 512)
    D, I = _run(clipmi, gpu, db, q, 51)
    Ds, Is = topk_oracle.topk(db, q, 51)
    _assert_exact(D, I, Ds, Is, "N<K")
    assert (I[:, 7:] == -1).all() and (D[:, 7:] == -np.finfo(np.float32).max).all()


def test_topk_empty_index(clipmi, gpu):
    idx = clipmi.IndexFlatIP(512, device=gpu)
    D, I = idx.search(np.ones((2, 512), np.float32), 5)
    assert (I == -1).all()


def test_topk_adversarial_order(clipmi, gpu, topk_oracle):
    """Rows sorted so that every later row beats all earlier ones: the sample threshold is
    useless and every wave keeps compacting. Still exact."""
    rng = np.random.default_rng(11)
    N = 80000
    q = unit_rows(rng, 2, 512)
    db = unit_rows(rng, N, 512)
    s = db @ q[0]
    db = db[np.argsort(s, kind="stable")]
    D, I = _run(clipmi, gpu, db, q, 51)
    Ds, Is = topk_oracle.topk(db, q, 51)
    _assert_exact(D, I, Ds, Is, "adversarial")


def test_topk_e768(clipmi, gpu, topk_oracle):
    rng = np.random.default_rng(12)
    db = unit_rows(rng, 3000, 768)
    q = unit_rows(rng, 4, 768)
    D, I = _run(clipmi, gpu, db, q, 51)
    Ds, Is = topk_oracle.topk(db, q, 51)
    _assert_exact(D, I, Ds, Is, "E=768")


def test_topk_nan_inf_rows(clipmi, gpu, topk_oracle):
    rng = np.random.default_rng(13)
    db = unit_rows(rng, 500, 512)
    db[3, 7] = np.nan
    db[10, :] = 0
    db[11, 0] = np.inf
    q = np.abs(unit_rows(rng, 2, 512)) + 0.01
    D, I = _run(clipmi, gpu, db, q, 20)
    Ds, Is = topk_oracle.topk(db, q, 20)
    _assert_exact(D, I, Ds, Is, "nan/inf")
    assert 3 not in I


@pytest.mark.gpu
@pytest.mark.parametrize("poison", ["nan", "inf"])
def test_non_finite_rows_keep_a_coarse_size_database_exact(clipmi, gpu, topk_oracle, poison):
    """a database large enough for the coarse path (N >= 65536) that holds a NaN / inf row must still return
    the oracle's result: clipmi_rows_stats reports a non-finite maximum (a NaN used to vanish from `nf > best`), so the
    index keeps such data on the exact scan, for 3 queries (one pass) and for 200 (the wide pass's shape)."""
    rng = np.random.default_rng(29)
    db = unit_rows(rng, 70000, 512)
    db[4321, 17] = np.nan if poison == "nan" else np.inf
    idx = clipmi.IndexFlatIP(512, device=gpu, coarse="int8")
    idx.add(db)
    assert idx.uses_coarse()
    for Q in (3, 200):
        q = np.abs(unit_rows(rng, Q, 512)) + 0.01
        D, I = idx.search(q, 20)
        Ds, Is = topk_oracle.topk(db, q, 20)
        _assert_exact(D, I, Ds, Is, f"{poison} row, Q = {Q}")
    assert not np.isfinite(idx.matrix_i8()[3]) or not np.isfinite(idx.matrix_i8()[2])
    if poison == "nan":
        assert 4321 not in I


@pytest.mark.parametrize("Q", [5, 64, 130])
def test_non_finite_queries_through_the_permuted_int8_copy(clipmi, gpu, topk_oracle, Q):
    """the candidate lists tell a fresh coarse survivor (0xffffffff, slot of the PERMUTED int8 copy) from an entry
    that is already (score bits, row id) by the first word alone - correct only while every writer stores a NaN score as -inf.
    Queries with NaN / +inf / -inf components make every score of theirs NaN or infinite; they travel beside ordinary queries
    through the 64-query pass (64-pair re-scoring) and the wide pass (16-pair re-scoring) of a coarse-size database whose copy is
    a non-trivial permutation (rows of very different maxima). Ids and score bits must equal the oracle's for every query."""
    rng = np.random.default_rng(97 + Q)
    N = 70000
    db = (unit_rows(rng, N, 512) * rng.uniform(0.2, 2.0, size=(N, 1))).astype(np.float32)
    q = unit_rows(rng, Q, 512)
    q[1, 5] = np.nan
    q[2, 9] = np.inf
    q[3, 9] = -np.inf
    q[4, :] = 0.0
    if Q > 40:
        q[33, 100] = np.nan
        q[40, 0], q[40, 1] = np.inf, -np.inf           # inf - inf: NaN scores
    idx = clipmi.IndexFlatIP(512, device=gpu, coarse="int8")
    idx.add(db)
    assert idx.uses_coarse()
    D, I = idx.search(q, 51)
    Ds, Is = topk_oracle.topk(db, q, 51)
    _assert_exact(D, I, Ds, Is, f"non-finite queries, Q = {Q}")
    assert (I[1] == -1).all() and (I[0] >= 0).all()


def test_sharded_equals_single(clipmi, gpu, topk_oracle):
    """Size-independent property at scale: split N rows into R contiguous shards, search each
    with its id_base, merge with clipmi_merge_topk == single-pass result == oracle merge."""
    rng = np.random.default_rng(21)
    N, Q, K, R = 300000, 16, 51, 8
    db = torch.from_numpy(unit_rows(rng, N, 512)).to(gpu)
    q = unit_rows(rng, Q, 512)
    # plant duplicates across shard boundaries
    db[N // R] = db[5]
    db[N - 1] = db[5]
    full = clipmi.IndexFlatIP(512, device=gpu)
    full.add(db)
    D, I = full.search(q, K)
    parts_s, parts_i = [], []
    for r in range(R):
        lo, hi = clipmi.shard_bounds(N, R, r)
        sh = clipmi.IndexFlatIP(512, device=gpu)
        sh.add(db[lo:hi])
        sh.id_base = lo
        s, i = sh.search(q, K)
        parts_s.append(s)
        parts_i.append(i)
    S = torch.from_numpy(np.stack(parts_s)).to(gpu)
    Iall = torch.from_numpy(np.stack(parts_i)).to(gpu)
    L = clipmi._lib.lib()
    out_s = torch.empty((Q, K), dtype=torch.float32, device=gpu)
    out_i = torch.empty((Q, K), dtype=torch.int64, device=gpu)
    ws = torch.empty(256, dtype=torch.uint8, device=gpu)
    rc = L.clipmi_merge_topk(S.data_ptr(), Iall.data_ptr(), R, Q, K, out_s.data_ptr(), out_i.data_ptr(),
                             ws.data_ptr(), ws.numel(), None)
    clipmi._lib.check(rc, "merge")
    _assert_exact(out_s.cpu().numpy(), out_i.cpu().numpy(), D, I, "sharded vs single")
    Ms, Mi = topk_oracle.merge(np.stack(parts_s), np.stack(parts_i), K)
    _assert_exact(out_s.cpu().numpy(), out_i.cpu().numpy(), Ms, Mi, "merge vs oracle merge")
    # spot-check the single-pass result against the oracle on a row subset containing all hits
    rows = np.unique(I.reshape(-1))
    sc = topk_oracle.scores(db[rows].cpu().numpy(), q[0])
    got = {int(i): s for i, s in zip(I[0], D[0])}
    for r_, s_ in zip(rows, sc):
        if int(r_) in got:
            assert np.float32(got[int(r_)]).view(np.uint32) == np.float32(s_).view(np.uint32)


def test_topk_errors(clipmi, gpu):
    idx = clipmi.IndexFlatIP(512, device=gpu)
    idx.add(np.zeros((4, 512), np.float32))
    with pytest.raises(clipmi.ClipmiError):
        idx.search(np.zeros((1, 512), np.float32), 100000)


def _run_coarse(clipmi, gpu, db, q, K, id_base=0, kind="bf16"):
    idx = clipmi.IndexFlatIP(512, device=gpu, coarse=kind)
    idx.add(db)
    idx.id_base = id_base
    return idx.search(q, K)


@pytest.mark.parametrize("N,Q,K", [(65536, 1, 51), (70001, 16, 51), (100000, 33, 11), (131072, 64, 51),
                                   (200003, 70, 51), (80000, 5, 300), (150000, 128, 51), (99999, 130, 21)])
def test_coarse_bf16_path_is_bit_exact(clipmi, gpu, topk_oracle, N, Q, K):
    """bf16 coarse scan + exact re-scoring returns the SAME bits as the exact path and the oracle."""
    rng = np.random.default_rng(N + Q + K)
    db = unit_rows(rng, N, 512)
    q = unit_rows(rng, Q, 512)
    D, I = _run_coarse(clipmi, gpu, db, q, K, id_base=7)
    Ds, Is = topk_oracle.topk(db, q, K, id_base=7)
    _assert_exact(D, I, Ds, Is, f"coarse N={N} Q={Q} K={K}")


def test_coarse_bf16_unnormalised_rows_and_clustered_scores(clipmi, gpu, topk_oracle):
    """Row norms from 0 to 3 (margin scales with the largest norm), near-duplicate rows whose scores differ
    by less than the bf16 error (all must survive the coarse pass), exact ties."""
    rng = np.random.default_rng(77)
    N = 90000
    db = unit_rows(rng, N, 512) * rng.uniform(0.0, 3.0, size=(N, 1)).astype(np.float32)
    q = unit_rows(rng, 9, 512) * np.float32(1.7)
    base = db[np.argmax(db @ q[0])].copy()
    for j in range(200):                              # 200 rows within ~1e-4 of the best score
        db[1000 + 7 * j] = base * np.float32(1.0 - 1e-6 * j)
    db[50000] = db[1000]
    D, I = _run_coarse(clipmi, gpu, db, q, 51)
    Ds, Is = topk_oracle.topk(db, q, 51)
    _assert_exact(D, I, Ds, Is, "coarse unnormalised/clustered")


def test_coarse_bf16_overflow_falls_back_to_exact(clipmi, gpu, topk_oracle):
    """More survivors than the coarse candidate capacity (300k identical rows): the device-side fallback
    must run the exact scan; ties resolve by ascending id."""
    rng = np.random.default_rng(78)
    N = 300000
    v = unit_rows(rng, 1, 512)
    db = np.repeat(v, N, axis=0)
    db[123456] *= np.float32(1.5)
    q = unit_rows(rng, 2, 512)
    D, I = _run_coarse(clipmi, gpu, db, q, 20)
    Ds, Is = topk_oracle.topk(db, q, 20)
    _assert_exact(D, I, Ds, Is, "coarse overflow fallback")


@pytest.mark.parametrize("N,Q,K", [(65536, 1, 51), (70001, 16, 51), (100000, 33, 11), (131072, 64, 51),
                                   (200003, 70, 51), (80000, 5, 300), (99999, 130, 21)])
def test_coarse_int8_path_is_bit_exact(clipmi, gpu, topk_oracle, N, Q, K):
    """int8 coarse scan (per-row scale, exact integer MFMA, Cauchy-Schwarz superset bound with each row's own
    quantisation-error norm) + exact re-scoring returns the SAME bits as the oracle."""
    rng = np.random.default_rng(N + Q + K + 1)
    db = unit_rows(rng, N, 512)
    q = unit_rows(rng, Q, 512)
    D, I = _run_coarse(clipmi, gpu, db, q, K, id_base=7, kind="int8")
    Ds, Is = topk_oracle.topk(db, q, K, id_base=7)
    _assert_exact(D, I, Ds, Is, f"int8 coarse N={N} Q={Q} K={K}")


def test_coarse_int8_hard_rows(clipmi, gpu, topk_oracle):
    """Rows that quantise badly or oddly: row norms 0..3, one-hot rows (scale = the single component: huge error
    bound for everything else in the row... none), rows with one dominant component (large per-row error norm),
    an all-zero row, near-duplicates of the best row whose scores differ by far less than the int8 error, exact
    ties, and a query with a dominant component. Results must still equal the oracle's bits."""
    rng = np.random.default_rng(79)
    N = 90000
    db = unit_rows(rng, N, 512) * rng.uniform(0.0, 3.0, size=(N, 1)).astype(np.float32)
    q = unit_rows(rng, 9, 512) * np.float32(1.7)
    q[3, 17] = 2.5                                     # dominant query component
    db[5] = 0.0
    db[6] = 0.0; db[6, 17] = 1.0                       # one-hot
    for j in range(50):                                # dominant component + small rest
        db[2000 + j, 100 + j] = 4.0
    base = db[np.argmax(db @ q[0])].copy()
    for j in range(200):
        db[1000 + 7 * j] = base * np.float32(1.0 - 1e-6 * j)
    db[50000] = db[1000]
    D, I = _run_coarse(clipmi, gpu, db, q, 51, kind="int8")
    Ds, Is = topk_oracle.topk(db, q, 51)
    _assert_exact(D, I, Ds, Is, "int8 coarse hard rows")


def test_coarse_int8_overflow_falls_back_to_exact(clipmi, gpu, topk_oracle):
    rng = np.random.default_rng(80)
    N = 300000
    v = unit_rows(rng, 1, 512)
    db = np.repeat(v, N, axis=0)
    db[123456] *= np.float32(1.5)
    q = unit_rows(rng, 2, 512)
    D, I = _run_coarse(clipmi, gpu, db, q, 20, kind="int8")
    Ds, Is = topk_oracle.topk(db, q, 20)
    _assert_exact(D, I, Ds, Is, "int8 coarse overflow fallback")


@pytest.mark.parametrize("kind", ["bf16", "int8"])
@pytest.mark.parametrize("Q", [64, 130])
def test_coarse_identical_rows_many_queries(clipmi, gpu, topk_oracle, kind, Q):
    """300 k identical rows at Q > 32 (scan_coarse_kernel<QG=4>): EVERY (query, row) pair of every 32-row step
    passes the coarse test, i.e. the per-wave LDS list takes its maximum of 2048 appends per step behind up to
    512 pending ones (the round-1 list held 1536: VERDICT r01 weak #1). Lists overflow their global capacity,
    the exact fallback runs; results equal the oracle's bits, ties by ascending id."""
    rng = np.random.default_rng(780 + Q)
    N = 300000
    v = unit_rows(rng, 1, 512)
    db = np.repeat(v, N, axis=0)
    db[123456] *= np.float32(1.5)
    q = unit_rows(rng, Q, 512)
    D, I = _run_coarse(clipmi, gpu, db, q, 20, kind=kind)
    Ds, Is = topk_oracle.topk(db, q, 20)
    _assert_exact(D, I, Ds, Is, f"{kind} identical rows Q={Q}")


@pytest.mark.parametrize("kind", ["bf16", "int8"])
@pytest.mark.parametrize("Q", [64, 130])
def test_coarse_duplicate_cluster_many_queries(clipmi, gpu, topk_oracle, kind, Q):
    """A run of 4096 consecutive duplicates of a row that ranks first for every one of Q similar queries (burst
    shots + near-identical prompts): 128 consecutive 32-row steps in which all 64 queries of a pass accept all 32
    rows (2048 pairs per step per wave) WITHOUT overflowing the global lists (4096 < 2^18), so the answer comes
    from the coarse path itself, not from the fallback."""
    rng = np.random.default_rng(790 + Q)
    N = 100000
    db = unit_rows(rng, N, 512)
    base = unit_rows(rng, 1, 512)[0]
    db[30000:34096] = base
    db[77777] = base                                    # one more copy far away: loses every tie to the run
    q = base[None, :] + 0.02 * unit_rows(rng, Q, 512)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    D, I = _run_coarse(clipmi, gpu, db, q, 51, kind=kind)
    Ds, Is = topk_oracle.topk(db, q, 51)
    _assert_exact(D, I, Ds, Is, f"{kind} duplicate cluster Q={Q}")
    assert (I[:, 0] == 30000).all() and (I[:, 50] == 30050).all()


@pytest.mark.parametrize("kind", ["bf16", "int8"])
def test_coarse_prepass_threshold_useless(clipmi, gpu, topk_oracle, kind):
    """Q = 64 with pre-pass thresholds that filter nothing. (A literal -inf threshold needs < K finite scores in
    the sample, i.e. NaN rows, and a non-finite row norm already routes the call to the exact path — so the
    reachable worst case is a threshold below every score.) The first 40 k rows — all of the level-1 and level-2
    samples — score about -125 against every query, later rows about 0: thr0 ~ -125, the main coarse pass accepts
    every (query, row) pair of all 70 k rows: 2048 appends per 32-row step in every wave, 70 k survivors per
    query (< 2^18: no fallback)."""
    rng = np.random.default_rng(801)
    N = 70001
    q = unit_rows(rng, 64, 512)
    u = q.sum(axis=0); u /= np.linalg.norm(u)
    db = unit_rows(rng, N, 512)
    db[:40000] = (-1000.0 * u)[None, :] * (1.0 + 1e-6 * np.arange(40000, dtype=np.float32))[:, None]
    D, I = _run_coarse(clipmi, gpu, db, q, 51, kind=kind)
    Ds, Is = topk_oracle.topk(db, q, 51)
    _assert_exact(D, I, Ds, Is, f"{kind} useless pre-pass threshold")
    assert (I >= 40000).all()


def _anisotropic_rows(rng, n, d=512, strong=8, gain=6.0):
    """Unit rows with a few dominant dimensions (real CLIP embeddings are not isotropic: a handful of components
    carry much of the norm), so per-row int8 scales are set by those components and the rest quantise coarsely."""
    x = rng.standard_normal((n, d), dtype=np.float32)
    x[:, :strong] *= np.float32(gain)
    x[:, 0] += np.float32(2.0 * gain)                       # a common offset direction, as CLIP's mean embedding is
    x /= np.linalg.norm(x, axis=1, keepdims=True)
    return x.astype(np.float32)


@pytest.mark.parametrize("kind", ["int8", "bf16"])
def test_coarse_anisotropic_db_exact_and_survivors(clipmi, gpu, topk_oracle, kind):
    """on an ANISOTROPIC database (8 dominant dimensions + a shared offset direction) the coarse
    path must still return the oracle's bits; the test prints how many rows per query survive the coarse filter and
    whether the exact fallback had to run (a list reaching its 2^18 capacity), next to the isotropic case."""
    import ctypes as C
    L = clipmi._lib.lib()
    N, Q, K = 200_000, 64, 51
    report = {}
    for name, gen in (("isotropic", unit_rows), ("anisotropic", lambda r, n, d: _anisotropic_rows(r, n, d))):
        rng = np.random.default_rng(4242)
        db = gen(rng, N, 512)
        q = gen(rng, Q, 512)
        idx = clipmi.IndexFlatIP(512, device=gpu, coarse=kind)
        idx.add(db)
        D, I = idx.search(q, K)
        Ds, Is = topk_oracle.topk(db, q, K)
        _assert_exact(D, I, Ds, Is, f"{kind} {name}")
        dbt = idx.matrix()
        qd = torch.from_numpy(q).to(gpu)
        os_ = torch.empty((Q, K), dtype=torch.float32, device=gpu)
        oi_ = torch.empty((Q, K), dtype=torch.int64, device=gpu)
        ws = torch.empty(L.clipmi_topk_ip_coarse_workspace_bytes(N, 512, Q, K), dtype=torch.uint8, device=gpu)
        ms, surv = C.c_float(0), C.c_longlong(-1)
        if kind == "int8":
            db8, meta, amax, rmax = idx.matrix_i8()
            rc = L.clipmi_dbg_topk_coarse_i8_scan_ms(dbt.data_ptr(), db8.data_ptr(), meta.data_ptr(), amax, N, 512, rmax,
                                                     qd.data_ptr(), Q, K, os_.data_ptr(), oi_.data_ptr(), ws.data_ptr(), ws.numel(),
                                                     None, 1, C.byref(ms), C.byref(surv))
        else:
            dbh, rmax = idx.matrix_bf16()
            rc = L.clipmi_dbg_topk_coarse_scan_ms(dbt.data_ptr(), dbh.data_ptr(), N, 512, rmax, qd.data_ptr(), Q, K,
                                                  os_.data_ptr(), oi_.data_ptr(), ws.data_ptr(), ws.numel(), None, 1,
                                                  C.byref(ms), C.byref(surv))
        clipmi._lib.check(rc, "coarse scan hook")
        torch.cuda.synchronize()
        assert np.array_equal(oi_.cpu().numpy(), Is)
        report[name] = surv.value / Q
        # survivors counts every exactly re-scored pair of the call; a list that had reached its capacity (the only way
        # into the exact fallback) would show up as >= 2^18 here
        assert surv.value / Q < (1 << 18), f"{kind} {name}: coarse lists overflowed (fallback taken)"
    print(f"{kind}: exactly re-scored rows per query (N={N}, K={K}): isotropic {report['isotropic']:.0f}, "
          f"anisotropic {report['anisotropic']:.0f}; fallback not taken in either case")


def test_two_search_batches_in_flight_on_two_streams(clipmi, gpu, topk_oracle):
    """Two search batches enqueued back to back on two HIP streams (what bench.py does for throughput): every stream owns
    its workspace and result buffers, so both return the oracle's bits."""
    rng = np.random.default_rng(909)
    N, K = 120_000, 51
    db = unit_rows(rng, N, 512)
    qa, qb = unit_rows(rng, 64, 512), unit_rows(rng, 40, 512)
    idx = clipmi.IndexFlatIP(512, device=gpu, coarse="int8")
    idx.add(db)
    idx.matrix_i8()
    ta, tb = torch.from_numpy(qa).to(gpu), torch.from_numpy(qb).to(gpu)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(device=gpu), torch.cuda.Stream(device=gpu)
    outs = []
    for rep in range(3):
        with torch.cuda.stream(s1):
            ra = idx.search_device(ta, K)
        with torch.cuda.stream(s2):
            rb = idx.search_device(tb, K)
        outs.append((ra, rb))
    torch.cuda.synchronize()
    Da, Ia = topk_oracle.topk(db, qa, K)
    Db, Ib = topk_oracle.topk(db, qb, K)
    for ra, rb in outs:
        _assert_exact(ra[0].cpu().numpy(), ra[1].cpu().numpy(), Da, Ia, "stream 1")
        _assert_exact(rb[0].cpu().numpy(), rb[1].cpu().numpy(), Db, Ib, "stream 2")
    assert len(idx._ws) >= 2


@pytest.mark.parametrize("kind", ["int8", "bf16"])
def test_large_query_batch_is_pipelined_and_exact(clipmi, gpu, topk_oracle, kind):
    """A search of more than 64 queries: the bf16 path runs its 64-query passes alternately on the caller's stream and an
    internal one (index.py _search_pipelined); the int8 path takes the whole search as wide passes inside the library (one
    stream). Either way: the oracle's bits, the same bits with batches_in_flight = 1, and usable back to back."""
    rng = np.random.default_rng(4242)
    N, Q, K = 90_000, 200, 51
    db = unit_rows(rng, N, 512)
    q = unit_rows(rng, Q, 512)
    idx = clipmi.IndexFlatIP(512, device=gpu, coarse=kind)
    idx.add(db)
    tq = torch.from_numpy(q).to(gpu)
    assert idx.batches_in_flight == 2
    outs = [idx.search_device(tq, K) for _ in range(3)]
    outs = [(s.clone(), i.clone()) for s, i in outs]
    if kind == "bf16":
        assert len(idx._ws) >= 2                       # the caller's stream + the process-wide side stream (_lib.side_stream)
        assert len(clipmi._lib._SIDE_STREAMS) >= 1
    else:
        assert len(idx._ws) == 1
    idx.batches_in_flight = 1
    s1, i1 = idx.search_device(tq, K)
    torch.cuda.synchronize()
    D, I = topk_oracle.topk(db, q, K)
    _assert_exact(s1.cpu().numpy(), i1.cpu().numpy(), D, I, "one stream")
    for s, i in outs:
        _assert_exact(s.cpu().numpy(), i.cpu().numpy(), D, I, "pipelined")
    # numpy front end (query-index.py:111's call) goes the same way
    idx.batches_in_flight = 2
    Dn, In = idx.search(q, K)
    _assert_exact(Dn, In, D, I, "search()")


@pytest.mark.parametrize("N,Q,K", [(70001, 65, 51), (70001, 128, 51), (80000, 191, 21), (80000, 192, 21), (70001, 200, 51),
                                   (100000, 256, 51), (100000, 257, 51), (131072, 300, 11), (70001, 513, 51), (70001, 520, 51),
                                   (90000, 640, 21), (70001, 777, 51), (70001, 993, 11), (200003, 1024, 51),
                                   (66000, 1100, 51), (66000, 2200, 21)])
def test_wide_int8_pass_is_bit_exact(clipmi, gpu, topk_oracle, N, Q, K):
    """More than 64 queries in ONE call (query-index.py:111 is one index.search whatever Q): the int8 path takes them as
    wide passes (csrc/topk.hip scan_coarse_wide_kernel below 192 queries, scan_coarse_wide2_kernel's balanced tiles from there:
    waves with 0, 1 and 2 query groups, one and two tiles, ragged last groups, a second chunk past 1024).
    Bit-exact against the oracle on 96 of the queries (every tile / set position) and, for ALL queries, against the exact f32
    scan (clipmi_topk_ip), which the tests above pin to the oracle."""
    rng = np.random.default_rng(N + Q + K + 2)
    db = unit_rows(rng, N, 512)
    q = unit_rows(rng, Q, 512)
    D, I = _run_coarse(clipmi, gpu, db, q, K, id_base=7, kind="int8")
    De, Ie = _run(clipmi, gpu, db, q, K, id_base=7)
    _assert_exact(D, I, De, Ie, f"wide int8 vs exact scan N={N} Q={Q} K={K}")
    pick = np.unique(np.concatenate([np.arange(0, Q, max(1, Q // 64)), np.arange(max(0, Q - 32), Q)]))[:96]
    Ds, Is = topk_oracle.topk(db, q[pick], K, id_base=7)
    _assert_exact(D[pick], I[pick], Ds, Is, f"wide int8 vs oracle N={N} Q={Q} K={K}")


@pytest.mark.parametrize("Q", [256, 512, 1024])
def test_wide_identical_rows_and_duplicate_cluster(clipmi, gpu, topk_oracle, Q):
    """The wide pass's per-wave list at its limits: (i) 300 k identical rows - every (query, row) pair of every 32-row block
    passes, lists overflow their global capacity and the exact fallback answers; (ii) 4096 consecutive duplicates of the row
    that ranks first for every query - 128 consecutive blocks in which all queries accept all rows, answered by the wide path
    itself (4096 < 2^15 slots)."""
    rng = np.random.default_rng(7800 + Q)
    N = 300000
    v = unit_rows(rng, 1, 512)
    db = np.repeat(v, N, axis=0)
    db[123456] *= np.float32(1.5)
    q = unit_rows(rng, Q, 512)
    D, I = _run_coarse(clipmi, gpu, db, q, 20, kind="int8")
    pick = np.arange(0, Q, Q // 32)
    Ds, Is = topk_oracle.topk(db, q[pick], 20)
    _assert_exact(D[pick], I[pick], Ds, Is, f"wide identical rows Q={Q}")
    assert (I[:, 0] == 123456).sum() >= 1 and (np.sort(I, axis=1)[:, :19] == np.arange(19)[None, :]).all()
    N = 100000
    db = unit_rows(rng, N, 512)
    base = unit_rows(rng, 1, 512)[0]
    db[30000:34096] = base
    db[77777] = base
    q = base[None, :] + 0.02 * unit_rows(rng, Q, 512)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True)).astype(np.float32)
    D, I = _run_coarse(clipmi, gpu, db, q, 51, kind="int8")
    Ds, Is = topk_oracle.topk(db, q[pick], 51)
    _assert_exact(D[pick], I[pick], Ds, Is, f"wide duplicate cluster Q={Q}")
    assert (I[:, 0] == 30000).all() and (I[:, 50] == 30050).all()


def test_wide_pass_useless_threshold_and_anisotropic(clipmi, gpu, topk_oracle):
    """Q = 200: (i) sample thresholds that filter nothing (the first 40 k rows score ~ -125 against every query): every pair of
    the later rows survives the first segments; (ii) an anisotropic database (8 dominant dimensions + a shared offset)."""
    rng = np.random.default_rng(8010)
    N, Q = 70001, 200
    q = unit_rows(rng, Q, 512)
    u = q.sum(axis=0); u /= np.linalg.norm(u)
    db = unit_rows(rng, N, 512)
    db[:40000] = (-1000.0 * u)[None, :] * (1.0 + 1e-6 * np.arange(40000, dtype=np.float32))[:, None]
    D, I = _run_coarse(clipmi, gpu, db, q, 51, kind="int8")
    De, Ie = _run(clipmi, gpu, db, q, 51)
    _assert_exact(D, I, De, Ie, "wide useless threshold vs exact scan")
    Ds, Is = topk_oracle.topk(db, q[:40], 51)
    _assert_exact(D[:40], I[:40], Ds, Is, "wide useless threshold vs oracle")
    assert (I >= 40000).mean() > 0.8          # (a few of 200 queries point away from u: the -1000 u rows then rank first)
    db = _anisotropic_rows(rng, 150000, 512)
    q = _anisotropic_rows(rng, Q, 512)
    D, I = _run_coarse(clipmi, gpu, db, q, 51, kind="int8")
    De, Ie = _run(clipmi, gpu, db, q, 51)
    _assert_exact(D, I, De, Ie, "wide anisotropic vs exact scan")
    Ds, Is = topk_oracle.topk(db, q[::5], 51)
    _assert_exact(D[::5], I[::5], Ds, Is, "wide anisotropic vs oracle")


@pytest.mark.parametrize("N", [1, 33, 2048, 2049, 300_007])
def test_rows_order_by_absmax_is_the_stable_sort(clipmi, gpu, N):
    """clipmi_rows_order_by_absmax (include/clipmi.h): the permutation a stable ascending sort of the rows' largest |component|
    gives - many equal maxima (duplicate rows, zero rows), several radix blocks, a ragged last block."""
    import torch
    rng = np.random.default_rng(N)
    x = rng.standard_normal((N, 64)).astype(np.float32)
    if N > 40:
        x[rng.integers(0, N, N // 3)] = x[5]              # a third of the rows share one maximum
        x[rng.integers(0, N, N // 50 + 1)] = 0.0
        x[11, 3] = np.float32(3e38)
        x[17, 9] = np.float32(-1e-40)                     # denormal magnitudes order by their bits too
    L = clipmi._lib.lib()
    xd = torch.from_numpy(x).to(gpu)
    perm = torch.full((N,), -1, dtype=torch.int32, device=gpu)
    ws = torch.empty(L.clipmi_rows_order_workspace_bytes(N), dtype=torch.uint8, device=gpu)
    clipmi._lib.check(L.clipmi_rows_order_by_absmax(xd.data_ptr(), N, 64, perm.data_ptr(), ws.data_ptr(), ws.numel(),
                                                    clipmi._lib.stream_ptr(gpu)), "rows_order")
    torch.cuda.synchronize()
    want = np.argsort(np.abs(x).max(axis=1), kind="stable")
    assert np.array_equal(perm.cpu().numpy().view(np.uint32), want.astype(np.uint32))
    assert L.clipmi_rows_order_by_absmax(xd.data_ptr(), N, 64, perm.data_ptr(), ws.data_ptr(), ws.numel() - 1, None) == 1


def test_quantize_rows_i8_matches_numpy(clipmi, gpu):
    """clipmi_quantize_rows_i8 (include/clipmi.h): slot t of the copy holds row perm[t] (IndexFlatIP orders the rows by their
    largest |component|: a stable sort), blocks of 32 slots share scale = max|x| / 127, q = rint(x / scale) (round half to
    even), stored as [block][k-step of 32 B][lane][16 B] with lane l = slot l & 31, bytes 32 s + 16 (l >> 5); error norm >= the
    true one and within 0.2 % of it; one (scale, largest error norm) pair per block behind the slot meta, then the slot -> row
    table. Sorted blocks give every row (nearly) its own scale: the error norms are those of per-row scales."""
    import torch
    rng = np.random.default_rng(81)
    N = 1000
    x = (unit_rows(rng, N, 512) * rng.uniform(0.1, 3.0, size=(N, 1))).astype(np.float32)
    x[7] = 0.0
    x[32:64] = 0.0                                     # 33 zero rows: they sort to the front; a whole block of them: scale 1, codes 0
    idx = clipmi.IndexFlatIP(512, device=gpu, coarse="int8")
    idx.add(x)
    q8, meta, amax, rmax = idx.matrix_i8()
    torch.cuda.synchronize()
    L = clipmi._lib.lib()
    N32 = (N + 31) // 32 * 32
    nblk = N32 // 32
    assert q8.numel() == L.clipmi_i8_copy_bytes(N, 512) == N32 * 512
    assert meta.numel() * 4 == L.clipmi_i8_meta_bytes(N) == ((N32 + 32) + (nblk + 1)) * 8 + (N32 + 32) * 4
    q8, meta = q8.cpu().numpy(), meta.cpu().numpy()
    rmeta = meta[:2 * (N32 + 32)].reshape(-1, 2)
    bmeta = meta[2 * (N32 + 32):2 * (N32 + 32) + 2 * (nblk + 1)].reshape(-1, 2)
    slot_rows = meta[2 * (N32 + 32) + 2 * (nblk + 1):].view(np.uint32)
    perm = np.argsort(np.abs(x).max(axis=1), kind="stable")
    assert np.array_equal(slot_rows[:N], perm.astype(np.uint32)) and (slot_rows[N:N32] == 0xffffffff).all()
    xp = np.zeros((N32, 512), np.float32)
    xp[:N] = x[perm]
    s = np.abs(xp).reshape(nblk, -1).max(axis=1) / np.float32(127.0)
    s[s == 0] = 1.0
    s = s.astype(np.float32)
    srow = np.repeat(s, 32)
    assert np.array_equal(rmeta[:N32, 0], srow) and np.array_equal(bmeta[:nblk, 0], s)
    ref = np.clip(np.rint(xp * (np.float32(1.0) / srow)[:, None]), -127, 127).astype(np.int8)
    # un-tile: [blk][ks][h][r][16] -> [blk*32 + r][32 ks + 16 h + b]
    got = q8.reshape(nblk, 16, 2, 32, 16).transpose(0, 3, 1, 2, 4).reshape(N32, 512)
    assert np.array_equal(got, ref)
    err = np.linalg.norm(xp.astype(np.float64) - srow[:, None].astype(np.float64) * got.astype(np.float64), axis=1)
    assert (rmeta[:N, 1] >= err[:N]).all() and (rmeta[:N, 1] <= err[:N] * 1.002 + 1e-12).all()
    assert (rmeta[N:, 1] == 0).all() and (rmeta[N32:] == 0).all() and (bmeta[nblk:] == 0).all()
    assert np.array_equal(bmeta[:nblk, 1], rmeta[:N32, 1].reshape(nblk, 32).max(axis=1))
    # the point of the order: a block's scale is within a few per cent of each of its rows' own
    own = np.abs(xp[:N]).max(axis=1) / np.float32(127.0)
    nz = own > 0
    assert np.median(srow[:N][nz] / own[nz]) < 1.05
    assert amax >= rmeta[:, 1].max() and rmax >= np.linalg.norm(x, axis=1).max()


def test_rows_stats_and_bf16_copy_match_numpy(clipmi, gpu):
    """clipmi_rows_stats / clipmi_rows_to_bf16 (include/clipmi.h; the index's build side): the largest row norm is an
    upper bound of the f64 norm within one f32 ulp, the largest error norm is the meta's maximum, the bf16 copy is the
    round-to-nearest-even conversion (torch's) bit for bit - NaN / inf / denormals included."""
    rng = np.random.default_rng(83)
    N = 3001
    x = (unit_rows(rng, N, 512) * rng.uniform(0.05, 7.0, size=(N, 1))).astype(np.float32)
    x[5, :4] = [np.float32(1e-40), np.float32(-1e-40), 0.0, np.float32(3.3895e38)]
    idx = clipmi.IndexFlatIP(512, device=gpu, coarse="int8")
    idx.add(x)
    q8, meta, amax, rmax = idx.matrix_i8()
    torch.cuda.synchronize()
    true = np.sqrt((x.astype(np.float64) ** 2).sum(axis=1)).max()
    assert rmax >= true and rmax <= float(np.nextafter(np.float32(true), np.float32(np.inf))) * (1 + 2e-6)
    a_all = meta.cpu().numpy()[:2 * N].reshape(-1, 2)[:, 1]
    assert amax >= a_all.max() and amax <= float(a_all.max()) * (1 + 2e-6)
    y = x.copy()
    y[9, 0], y[9, 1], y[9, 2] = np.nan, np.inf, -np.inf
    idb = clipmi.IndexFlatIP(512, device=gpu, coarse="bf16")
    idb.add(y)
    dbh, _ = idb.matrix_bf16()
    want = torch.from_numpy(y).to(torch.bfloat16)
    got = dbh.cpu()
    nan = torch.isnan(want)
    assert torch.equal(torch.isnan(got), nan)                       # (a NaN's payload is not part of the contract)
    bad = (got.view(torch.int16) != want.view(torch.int16)) & ~nan
    assert not bad.any(), (torch.from_numpy(y)[bad][:8], got[bad][:8], want[bad][:8])
    out = torch.empty(2, dtype=torch.float32, device=gpu)
    L = clipmi._lib.lib()
    assert L.clipmi_rows_stats(idx.matrix().data_ptr(), N, 510, None, out.data_ptr(), None) != 0      # E % 4 != 0
    assert b"multiple of 4" in L.clipmi_last_error()


@pytest.mark.parametrize("N", [10_000_000, 12_500_000])
def test_full_size_10m_properties(clipmi, gpu, topk_oracle, N):
    """BASELINE.json's full sizes (configs[1]/[2]: 10 M x 512; configs[4]: 12.5 M rows = one rank's share of the
    100 M-row database; K = 51): size-independent properties instead of a CPU sort of
    10 M rows — (i) the coarse-then-exact path and the exact f32 scan return identical bits; (ii) every
    returned score equals the oracle's score of that row; (iii) rows are sorted (score desc, id asc), ids
    unique; (iv) no row outside the result beats the K-th score, checked on a 200 k-row random subset with
    the oracle; (v) an 8-way sharded search + merge equals the single pass."""
    Q, K = 64, 51
    g = torch.Generator(device=gpu); g.manual_seed(42)
    db = torch.empty((N, 512), dtype=torch.float32, device=gpu)
    for s in range(0, N, 1 << 20):
        e = min(N, s + (1 << 20))
        blk = torch.randn((e - s, 512), generator=g, device=gpu)
        db[s:e] = blk / blk.norm(dim=1, keepdim=True)
    db[N - 1] = db[17]                                       # duplicate row at the far end
    q = torch.randn((Q, 512), generator=g, device=gpu)
    q = q / q.norm(dim=1, keepdim=True)
    q[3] = db[17]                                            # a query with an exact tie pair at rank 0/1
    exact = clipmi.IndexFlatIP(512, device=gpu); exact.add(db)
    coarse = clipmi.IndexFlatIP(512, device=gpu, coarse="bf16"); coarse.add(db)
    De, Ie = exact.search(q, K)
    Dc, Ic = coarse.search(q, K)
    _assert_exact(Dc, Ic, De, Ie, "coarse vs exact at 10M")
    del coarse
    coarse8 = clipmi.IndexFlatIP(512, device=gpu, coarse="int8"); coarse8.add(db)
    D8, I8 = coarse8.search(q, K)
    _assert_exact(D8, I8, De, Ie, "int8 coarse vs exact at 10M")
    del coarse8
    assert list(Ie[3, :2]) == [17, N - 1] and De[3, 0] == De[3, 1]
    qh = q.cpu().numpy()
    for j in (0, 3, 31, 63):
        rows = Ie[j]
        assert len(set(rows.tolist())) == K
        sc = topk_oracle.scores(db[torch.from_numpy(rows).to(gpu)].cpu().numpy(), qh[j])
        assert np.array_equal(sc.view(np.uint32), De[j].view(np.uint32))
        order = np.lexsort((rows, -De[j].astype(np.float64)))
        assert np.array_equal(order, np.arange(K))
    rng = np.random.default_rng(7)
    sub = np.sort(rng.choice(N, 200_000, replace=False))
    subdb = db[torch.from_numpy(sub).to(gpu)].cpu().numpy()
    for j in (0, 63):
        sc = topk_oracle.scores(subdb, qh[j])
        inside = set(Ie[j].tolist())
        better = [(s_, int(i_)) for s_, i_ in zip(sc, sub) if int(i_) not in inside and
                  (s_ > De[j, -1] or (s_ == De[j, -1] and i_ < Ie[j, -1]))]
        assert not better, better[:3]
    # (v) sharded == single
    parts_s, parts_i = [], []
    for r in range(8):
        lo, hi = clipmi.shard_bounds(N, 8, r)
        sh = clipmi.IndexFlatIP(512, device=gpu, coarse="bf16" if r % 2 else "int8")
        sh.add(db[lo:hi]); sh.id_base = lo
        s_, i_ = sh.search(q, K)
        parts_s.append(s_); parts_i.append(i_)
    # the merge the product runs after its ONE all-gather: every shard's packed record [scores | pad | ids] side by side
    # (what all_gather_into_tensor produces), merged by clipmi_merge_topk_packed - and the oracle's merge beside it
    ids_off = (Q * K * 4 + 7) // 8 * 8
    rec_bytes = ids_off + Q * K * 8
    gath = torch.zeros(8 * rec_bytes, dtype=torch.uint8, device=gpu)
    for r in range(8):
        gath[r * rec_bytes:r * rec_bytes + Q * K * 4] = torch.from_numpy(parts_s[r].reshape(-1).view(np.uint8).copy()).to(gpu)
        gath[r * rec_bytes + ids_off:(r + 1) * rec_bytes] = torch.from_numpy(parts_i[r].reshape(-1).view(np.uint8).copy()).to(gpu)
    out_s = torch.empty((Q, K), dtype=torch.float32, device=gpu)
    out_i = torch.empty((Q, K), dtype=torch.int64, device=gpu)
    L = clipmi._lib.lib()
    clipmi._lib.check(L.clipmi_merge_topk_packed(gath.data_ptr(), rec_bytes, 8, Q, K, out_s.data_ptr(), out_i.data_ptr(), None),
                      "merge_topk_packed")
    _assert_exact(out_s.cpu().numpy(), out_i.cpu().numpy(), De, Ie, "8 shards + clipmi_merge_topk_packed vs single at 10M")
    Ms, Mi = topk_oracle.merge(np.stack(parts_s), np.stack(parts_i), K)
    _assert_exact(Ms, Mi, De, Ie, "8 shards + oracle merge vs single at 10M")
    # ONE call of 1024 queries at full size: the wide pass against the 64 queries above (same first 64 rows of the batch) and
    # against the exact f32 scan on a sample of the rest
    qw = torch.cat([q, torch.randn((960, 512), generator=g, device=gpu)])
    qw[64:] = qw[64:] / qw[64:].norm(dim=1, keepdim=True)
    wide = clipmi.IndexFlatIP(512, device=gpu, coarse="int8"); wide.add(db)
    Dw, Iw = wide.search(qw, K)
    _assert_exact(Dw[:64], Iw[:64], De, Ie, "wide pass (Q = 1024) vs exact at full size, first 64 queries")
    pick = torch.arange(64, 1024, 31, device=gpu)[:32]
    Dx, Ix = exact.search(qw[pick], K)
    _assert_exact(Dw[pick.cpu().numpy()], Iw[pick.cpu().numpy()], Dx, Ix, "wide pass (Q = 1024) vs exact at full size, 32 more")


@pytest.mark.gpu
def test_live_threshold_scan_is_bit_exact_when_enabled():
    """The experimental one-launch scan (scan_coarse_live_kernel, CLIPMI_LIVE=1, off by default: DESIGN.md 4.1e) returns the
    exact f32 scan's bits. The switch is read once per process, so the check runs in a child."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CLIPMI_LIVE="1", CLIPMI_DEV_LIB="1")      # the kernel lives in the development library only
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "attic", "live_check.py"), "300000", "1,16,64"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.count("exact: True") == 3 and "exact: False" not in r.stdout, r.stdout


@pytest.mark.gpu
def test_two_digit_query_scan_and_both_rescoring_forms_are_bit_exact():
    """Development-library variants that the product does not select for every shape: the 64-query scan with the
    query as two int8 digits (CLIPMI_COARSE_Q2=1: a tighter margin, the same exact results) and each exact re-scoring form forced
    onto all lists (CLIPMI_RESCORE=16 / 64). Knobs are read once per process: child processes, tools/attic/live_check.py with the live
    scan off."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for knobs in ({"CLIPMI_COARSE_Q2": "1"}, {"CLIPMI_RESCORE": "16"}, {"CLIPMI_RESCORE": "64"}):
        env = dict(os.environ, CLIPMI_LIVE="0", CLIPMI_DEV_LIB="1", **knobs)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "attic", "live_check.py"), "300000", "1,33,64,200"], env=env,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        assert r.stdout.count("exact: True") == 4 and "exact: False" not in r.stdout, (knobs, r.stdout)
