"""Parity of the HIP flat inner-product top-k (through the C ABI, via the FAISSIndex drop-in)
against the CPU oracle.  Tolerances: tests/cases.py."""
import numpy as np
import pytest
import torch

import oracle
from amdrec import synth
from tests import cases

pytestmark = pytest.mark.gpu

_PREFILTER = "bf16"


@pytest.fixture(params=["bf16", "fp32"], autouse=True)
def prefilter(request):
    """Every case runs through both engines: amdrec_flat_search_mixed (bf16 MFMA filter + fp32 re-score +
    certificate, the default) and amdrec_flat_search (fp32 MFMA filter)."""
    global _PREFILTER
    _PREFILTER = request.param
    yield request.param
    _PREFILTER = "bf16"


def _mk(n, d, seed):
    return synth.unit_corpus(n, d, seed=seed)


def _both(xb, xq, k, ad_ids=None):
    from amdrec.index import FAISSIndex
    idx = FAISSIndex(xb.shape[1], index_type="Flat", prefilter=_PREFILTER)
    assert idx._mixed == (_PREFILTER == "bf16" and xb.shape[1] % 8 == 0)
    ora = oracle.search.FlatIndex(xb.shape[1])
    idx.add(xb, ad_ids)
    ora.add(xb, ad_ids)
    ids, D = idx.search(xq, k)
    rids, rD = ora.search(xq, k)
    assert ids.dtype == np.int64 and D.dtype == np.float32 and ids.shape == (len(xq), k) == D.shape
    return (ids, D), (rids, rD), ora


def _check(got, ref, ora, xq, tau=cases.TOPK_TAU):
    ids, D = got
    rids, rD = ref
    qn = oracle.search.normalize_l2(xq)
    id2pos = {v: i for i, v in enumerate(ora.id_map)}

    def scores_of(q, which):
        rows = np.array([id2pos[int(i)] for i in which])
        return (ora.xb[rows].astype(np.float64) @ qn[q].astype(np.float64)).astype(np.float32)

    oracle.search.check_topk(rD, rids, D, ids, tau=tau, score_tol=cases.SCORE_ATOL, scores_of=scores_of)


@pytest.mark.parametrize("n,nq,k", [(1000, 1, 10), (1000, 5, 500), (777, 33, 100), (4096, 70, 500),
                                    (8192, 130, 37), (3000, 200, 1)])
def test_small_corpus_all_candidates(n, nq, k):
    xb, xq = _mk(n, 256, 1), _mk(nq, 256, 2)
    got, ref, ora = _both(xb, xq, k)
    _check(got, ref, ora, xq)


@pytest.mark.parametrize("n,nq,k,d", [(50_000, 37, 500, 256), (120_001, 512, 500, 256), (30_000, 64, 2048, 128),
                                      (20_000, 3, 10, 64), (9_000, 257, 100, 32),
                                      (70_000, 1100, 50, 128),      # three 512-query groups in the streaming filter
                                      (25_000, 40, 100, 96)])       # dim outside {32,64,128,256}: generic bf16 tiles
def test_sampled_threshold_path(n, nq, k, d):
    xb, xq = _mk(n, d, 3), _mk(nq, d, 4)
    got, ref, ora = _both(xb, xq, k)
    _check(got, ref, ora, xq)


def test_unnormalised_inputs_are_renormalised_and_not_mutated():
    rng = np.random.default_rng(0)
    xb = (rng.standard_normal((5000, 256)) * 3).astype(np.float32)
    xq = (rng.standard_normal((9, 256)) * 0.1).astype(np.float64)      # non-fp32 input
    xb0, xq0 = xb.copy(), xq.copy()
    got, ref, ora = _both(xb, xq, 50)
    assert np.array_equal(xb, xb0) and np.array_equal(xq, xq0)
    _check(got, ref, ora, xq)
    assert np.all(got[1] <= 1.0 + 1e-5)


def test_k_larger_than_corpus_pads_like_faiss():
    xb, xq = _mk(100, 256, 5), _mk(4, 256, 6)
    (ids, D), (rids, rD), _ = _both(xb, xq, 500)
    assert np.all(np.isneginf(D[:, 100:])) and np.all(np.isneginf(rD[:, 100:]))
    # unfilled slots: position -1 -> id_map[-1] (faiss_retrieval.py:159)
    assert np.all(ids[:, 100:] == 99) and np.all(rids[:, 100:] == 99)
    assert np.array_equal(ids[:, :100], rids[:, :100])


def test_custom_ids_are_remapped():
    xb, xq = _mk(3000, 256, 7), _mk(6, 256, 8)
    ad_ids = (np.arange(3000) * 7 + 1000).tolist()
    got, ref, ora = _both(xb, xq, 20, ad_ids)
    _check(got, ref, ora, xq)
    assert got[0].min() >= 1000


def test_exact_ties_lower_position_wins():
    """Duplicated rows (training_pipeline.py:523 indexes every interaction row, so duplicates
    are expected): equal scores must come back lowest position first, bit-exact."""
    base = _mk(2500, 256, 9)
    xb = np.concatenate([base, base, base, base], axis=0)          # 10000 rows, 4 copies each
    xq = _mk(16, 256, 10)
    (ids, D), (rids, rD), _ = _both(xb, xq, 100)
    assert np.array_equal(ids, rids)
    assert np.abs(D - rD).max() <= cases.SCORE_ATOL
    # within each group of equal scores ids ascend
    for q in range(len(xq)):
        same = D[q, 1:] == D[q, :-1]
        assert np.all(ids[q, 1:][same] > ids[q, :-1][same])


def test_massive_ties_take_the_exact_fixup_path():
    """All rows identical: every score ties, the candidate list overflows -> fix-up scan."""
    row = _mk(1, 256, 11)
    xb = np.repeat(row, 20_000, axis=0)
    xq = _mk(3, 256, 12)
    (ids, D), (rids, rD), _ = _both(xb, xq, 500)
    assert np.array_equal(ids, np.tile(np.arange(500), (3, 1)))
    assert np.array_equal(ids, rids)
    assert np.abs(D - rD).max() <= cases.SCORE_ATOL


def test_adversarial_order_underflow_takes_fixup_path():
    """Corpus sorted so that the best rows sit in one unsampled stretch and most sampled rows
    score high for query 0: the sampled threshold is useless, the result must still be exact."""
    rng = np.random.default_rng(13)
    n, d = 40_000, 64
    xq = _mk(2, d, 14)
    xb = _mk(n, d, 15)
    s = xb @ xq[0]
    xb = xb[np.argsort(-s)]                                        # descending by query-0 score
    got, ref, ora = _both(xb, xq, 500)
    _check(got, ref, ora, xq)
    assert np.array_equal(np.sort(got[0][0]), np.arange(500))


def test_padded_leading_dimension_and_device_api():
    from amdrec.index import flat_search
    n, nq, k, d = 20_000, 50, 64, 96
    xb, xq = _mk(n, d, 16), _mk(nq, d, 17)
    big = torch.zeros((n, 128), dtype=torch.float32, device="cuda")
    big[:, :d] = torch.from_numpy(xb).cuda()
    qd = torch.from_numpy(xq).cuda()
    D = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    I = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    nfix = torch.zeros(1, dtype=torch.int32, device="cuda")
    flat_search(big[:, :d], n, qd, k, D, I, pos_offset=1_000_000, n_fixup=nfix)
    rD, rI = oracle.search.flat_ip_search(xb, xq, k)
    oracle.search.check_topk(rD, rI + 1_000_000, D.cpu().numpy(), I.cpu().numpy(), tau=cases.TOPK_TAU,
                             score_tol=cases.SCORE_ATOL)
    assert int(nfix.item()) == 0


def test_bf16_shadow_rows_and_max_norm():
    """amdrec_bf16_rows == round-to-nearest-even bf16 (torch's cast), max_norm == largest row norm."""
    from amdrec.index import FAISSIndex
    rng = np.random.default_rng(21)
    xb = (rng.standard_normal((5000, 72)) * rng.uniform(0.1, 3.0, (5000, 1))).astype(np.float32)
    idx = FAISSIndex(72, index_type="Flat")
    idx.add(xb[:3000])
    idx.add(xb[3000:])                                              # growth + append keep the shadow in step
    x = idx._xb[:5000]
    assert torch.equal(idx._xb16[:5000].view(torch.int16), x.to(torch.bfloat16).view(torch.int16))
    assert abs(idx._maxnorm[0].item() - x.norm(dim=1).max().item()) <= 1e-6
    dmax = (x - x.to(torch.bfloat16).float()).norm(dim=1).max().item()               # largest rounding-error norm of a row
    assert dmax <= idx._maxnorm[1].item() <= dmax * 1.001


def test_mixed_search_is_exact_on_unnormalised_rows_and_clustered_scores():
    """The C entry point itself (no renormalisation): rows of very different norms, and a cluster of rows whose
    exact scores differ by ~1e-5 around the k-th place - far below what bf16 resolves - must still come back as the
    exact fp32 top-k (pruning margin + certificate), without the fix-up path on the well-spread case."""
    from amdrec.index import flat_search_mixed
    from amdrec import _lib
    rng = np.random.default_rng(22)
    n, nq, k, d = 60_000, 70, 100, 128
    xb = (_mk(n, d, 23) * rng.uniform(0.2, 4.0, (n, 1))).astype(np.float32)
    xq = (_mk(nq, d, 24) * rng.uniform(0.5, 2.0, (nq, 1))).astype(np.float32)
    # 300 near-duplicates of query 0's direction: scores 1e-5 apart around ranks 1..300
    xb[1000:1300] = xq[0] / np.linalg.norm(xq[0]) * 3.0 + rng.standard_normal((300, d)).astype(np.float32) * 1e-5
    X = torch.from_numpy(xb).cuda()
    X16 = torch.empty((n, d), dtype=torch.bfloat16, device="cuda")
    mx = torch.zeros(2, dtype=torch.float32, device="cuda")
    lib = _lib.load()
    _lib.check(lib.amdrec_bf16_rows(_lib.ptr(X), n, d, d, _lib.ptr(X16), d, _lib.ptr(mx), _lib.stream_ptr(X.device)))
    Q = torch.from_numpy(xq).cuda()
    D = torch.empty((nq, k), dtype=torch.float32, device="cuda")
    I = torch.empty((nq, k), dtype=torch.int64, device="cuda")
    nfix = torch.zeros(1, dtype=torch.int32, device="cuda")
    flat_search_mixed(X, X16, mx, n, Q, k, D, I, n_fixup=nfix)
    rD, rI = oracle.search.flat_ip_search(xb, xq, k, dtype=np.float64)
    # scores up to 8 in magnitude here: scale the absolute tolerances by the largest |score|
    scale = float(np.abs(rD).max())
    oracle.search.check_topk(rD, rI, D.cpu().numpy(), I.cpu().numpy(), tau=cases.TOPK_TAU * scale,
                             score_tol=cases.SCORE_ATOL * scale)
    assert int(nfix.item()) <= 1                                    # at most the clustered query


def test_empty_and_error_paths():
    from amdrec import _lib
    from amdrec.index import FAISSIndex
    idx = FAISSIndex(256, index_type="Flat")
    ids, D = idx.search(_mk(2, 256, 1), 5)                          # empty index
    assert np.all(np.isneginf(D)) and ids.shape == (2, 5)
    idx.add(_mk(10, 256, 2))
    ids, D = idx.search(np.zeros((0, 256), np.float32), 5)         # no queries
    assert ids.shape == (0, 5)
    with pytest.raises(_lib.AmdrecError):
        idx.search(_mk(1, 256, 3), 5000)                            # k > AMDREC_MAX_K
    with pytest.raises(ValueError):
        FAISSIndex(256, index_type="Annoy")


def test_full_size_1m_against_oracle():
    """BASELINE config 2 size: 1M x 256 corpus, 512 queries, k=500 (oracle on 24 of them)."""
    from amdrec.index import FAISSIndex
    xb = _mk(1_000_000, 256, 1234)
    user, ad, nnum = synth.demo_dims()
    xq = _mk(512, 256, 99)
    idx = FAISSIndex(256, index_type="Flat", prefilter=_PREFILTER)       # both engines at full size
    assert idx._mixed == (_PREFILTER == "bf16")
    idx.add(xb)
    nfix = torch.zeros(1, dtype=torch.int32, device="cuda")
    ids, D = idx.search(xq, 500)
    # size-independent properties on all 512 queries
    assert np.all(np.diff(D, axis=1) <= 0)
    assert all(len(set(r.tolist())) == 500 for r in ids)
    recomputed = np.einsum("qkd,qd->qk", xb[ids[:64]].astype(np.float64), xq[:64].astype(np.float64))
    assert np.abs(recomputed - D[:64]).max() <= cases.SCORE_ATOL
    # oracle on a subset
    sub = np.arange(0, 512, 22)[:24]
    rD, rI = oracle.search.flat_ip_search(xb, xq[sub], 500)
    oracle.search.check_topk(rD, rI, D[sub], ids[sub], tau=cases.TOPK_TAU, score_tol=cases.SCORE_ATOL)


def test_batch_search_chunks_like_the_reference():
    """FAISSIndex.batch_search (faiss_retrieval.py:168-194): query-chunked search + vstack == one search, for a
    batch size that does not divide the query count, with custom ids; shapes and dtypes as search()."""
    from amdrec.index import FAISSIndex
    xb, xq = _mk(30_000, 128, 31), _mk(2_345, 128, 32)
    ad_ids = (np.arange(30_000)[::-1] * 3 + 7).tolist()
    idx = FAISSIndex(128, index_type="Flat", prefilter=_PREFILTER)
    ora = oracle.search.FlatIndex(128)
    idx.add(xb, ad_ids)
    ora.add(xb, ad_ids)
    ids, D = idx.batch_search(xq, k=50, batch_size=1000)                 # chunks of 1000, 1000, 345
    one_ids, one_D = idx.search(xq, 50)
    assert ids.shape == (2_345, 50) == D.shape and ids.dtype == np.int64 and D.dtype == np.float32
    assert np.array_equal(ids, one_ids) and np.array_equal(D, one_D)      # chunking changes nothing (bit-exact)
    sub = np.arange(0, 2_345, 97)
    rids, rD = ora.search(xq[sub], 50)
    _check((ids[sub], D[sub]), (rids, rD), ora, xq[sub])


def test_mixed_worst_case_rounding():
    """The certificate of amdrec_flat_search_mixed must hold when bf16 rounding errors of query AND row add up
    (eps used u = 2^-9; the true unit roundoff of bf16 is 2^-8, bound 2u + u^2 = 0.00783).
    Query and 50 'victim' rows have every component just below a bf16 midpoint (both round down: approx = 1.0,
    exact = 1.00773).  150 'strong' rows are exactly representable with exact score 1.00738 (approx 1.0035) and a
    medium population fixes the sampled threshold near 1.0025: with the round-1 bound the k-th re-scored value
    (1.00738) clears tau + eps (1.0065), the certificate passes and the victims - outside the candidate list,
    true rank 1..50 - are lost.  With the sound bound the query takes the exact fix-up scan."""
    from amdrec import _lib
    from amdrec.index import flat_search_mixed
    if _PREFILTER != "bf16":
        pytest.skip("mixed engine only")
    rng = np.random.default_rng(77)
    n, d, k = 20_000, 256, 100
    base = np.float32(2.0 ** -4)
    a = np.float32(2.0 ** -4 * (1 + 0.99 * 2.0 ** -8))                    # rounds DOWN to 2^-4 in bf16
    bump = np.float32(2.0 ** -4 * (1 + 2.0 ** -7))                         # exactly representable in bf16
    xb = np.full((n, d), base, dtype=np.float32)
    kinds = np.zeros(n, dtype=np.int64)                                     # 0 weak, 1 medium, 2 strong, 3 victim
    perm = rng.permutation(n)
    kinds[perm[:50]] = 3
    kinds[perm[50:200]] = 2
    kinds[perm[200:1700]] = 1
    for r in range(n):
        if kinds[r] == 3:
            xb[r, :] = a
            continue
        cnt = {0: rng.integers(0, 26), 1: rng.integers(77, 91), 2: 115}[int(kinds[r])]
        xb[r, rng.choice(d, cnt, replace=False)] = bump
    xq = np.stack([np.full(d, a, dtype=np.float32), _mk(1, d, 5)[0]])
    X = torch.from_numpy(xb).cuda()
    X16 = torch.empty((n, d), dtype=torch.bfloat16, device="cuda")
    mx = torch.zeros(2, dtype=torch.float32, device="cuda")
    lib = _lib.load()
    _lib.check(lib.amdrec_bf16_rows(_lib.ptr(X), n, d, d, _lib.ptr(X16), d, _lib.ptr(mx), _lib.stream_ptr(X.device)))
    Q = torch.from_numpy(xq).cuda()
    D = torch.empty((2, k), dtype=torch.float32, device="cuda")
    I = torch.empty((2, k), dtype=torch.int64, device="cuda")
    nfix = torch.zeros(1, dtype=torch.int32, device="cuda")
    flat_search_mixed(X, X16, mx, n, Q, k, D, I, n_fixup=nfix)
    got_ids = I.cpu().numpy()
    exact = xb.astype(np.float64) @ xq[0].astype(np.float64)
    approx = X16.float().cpu().numpy().astype(np.float64) @ Q[0].to(torch.bfloat16).float().cpu().numpy().astype(np.float64)
    victims = np.nonzero(kinds == 3)[0]
    # the construction really is adversarial: the bf16 error of the victims exceeds the round-1 bound, not the sound one
    rel = np.abs(approx[victims] - exact[victims]).max() / (np.linalg.norm(xq[0].astype(np.float64)) * np.linalg.norm(xb[victims[0]].astype(np.float64)))
    assert 0.00392 + 256 * 1.2e-7 < rel < 0.00783
    assert set(victims.tolist()) <= set(got_ids[0].tolist()), "victim rows (true rank 1..50) missing: unsound certificate"
    rD, rI = oracle.search.flat_ip_search(xb, xq, k, dtype=np.float64)
    oracle.search.check_topk(rD, rI, D.cpu().numpy(), got_ids, tau=cases.TOPK_TAU, score_tol=2 * cases.SCORE_ATOL)
    assert int(nfix.item()) >= 1                                            # query 0 went through the exact fix-up


def test_more_than_65535_queries_in_one_call():
    """the fix-up scan used gridDim.y = nq (limit 65535); the entry points accept nq < 2^24."""
    from amdrec.index import FAISSIndex
    if _PREFILTER != "fp32":
        pytest.skip("one engine is enough: both share the fix-up launch")
    xb, xq = _mk(3_000, 32, 41), _mk(66_000, 32, 42)
    idx = FAISSIndex(32, index_type="Flat", prefilter=_PREFILTER)
    idx.add(xb)
    ids, D = idx.search(xq, 5)
    sub = np.r_[0:40, 65_500:65_560, 65_990:66_000]
    rD, rI = oracle.search.flat_ip_search(xb, xq[sub], 5)
    oracle.search.check_topk(rD, rI, D[sub], ids[sub], tau=cases.TOPK_TAU, score_tol=cases.SCORE_ATOL)


@pytest.mark.parametrize("hot_tiles,expect_fixup", [(3, False), (24, True)])
def test_hits_concentrated_in_one_workgroups_segment(hot_tiles, expect_fixup):
    """The streaming pass files a hit in the segment of the workgroup that found it (CAND_CAP / 256 = 32 slots per
    query at this size); workgroup w owns the 128-row tiles w, w + 256, ...  Rows of `hot_tiles` tiles of ONE workgroup
    are made near-copies of query 0, so that its segment overflows: a few hundred extra hits go through the overflow
    block (exact result, no fix-up), more than 2048 make the query take the exact fix-up scan.  Query 1 is ordinary."""
    from amdrec import _lib
    from amdrec.index import flat_search_mixed
    if _PREFILTER != "bf16":
        pytest.skip("mixed engine only")
    n, d, k = 256 * 128 * 26, 256, 500
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    X = torch.randn((n, d), generator=g, device="cuda")
    X /= X.norm(dim=1, keepdim=True)
    Q = X[:2].clone() + 0.05 * torch.randn((2, d), generator=g, device="cuda")
    Q /= Q.norm(dim=1, keepdim=True)
    for i in range(hot_tiles):
        t = 5 + 256 * i
        rows = Q[0][None, :] + 0.02 * torch.randn((128, d), generator=g, device="cuda")
        X[t * 128:(t + 1) * 128] = rows / rows.norm(dim=1, keepdim=True)
    X16 = torch.empty((n, d), dtype=torch.bfloat16, device="cuda")
    mx = torch.zeros(2, dtype=torch.float32, device="cuda")
    lib = _lib.load()
    _lib.check(lib.amdrec_bf16_rows(_lib.ptr(X), n, d, d, _lib.ptr(X16), d, _lib.ptr(mx), _lib.stream_ptr(X.device)))
    D = torch.empty((2, k), dtype=torch.float32, device="cuda")
    I = torch.empty((2, k), dtype=torch.int64, device="cuda")
    nfix = torch.zeros(1, dtype=torch.int32, device="cuda")
    flat_search_mixed(X, X16, mx, n, Q, k, D, I, n_fixup=nfix)
    xb, xq = X.cpu().numpy(), Q.cpu().numpy()
    rD, rI = oracle.search.flat_ip_search(xb, xq, k, dtype=np.float64)
    oracle.search.check_topk(rD, rI, D.cpu().numpy(), I.cpu().numpy(), tau=cases.TOPK_TAU, score_tol=2 * cases.SCORE_ATOL)
    hot = set()
    for i in range(hot_tiles):
        hot |= set(range((5 + 256 * i) * 128, (5 + 256 * i + 1) * 128))
    assert len(set(I[0].cpu().tolist()) & hot) >= min(k, len(hot)) - 1      # query 0's top-k IS the hot rows
    assert (int(nfix.item()) >= 1) == expect_fixup


def test_profile_hook_times_tagged_launches_and_honours_the_prefix_filter():
    """amdrec_profile_enable / _only / _report (include/amdrec.h): every tagged launch is timed, or only the tags with
    the given prefix (bench.py times every kernel in a pre-pass and only the dominant one in its timed region)."""
    from amdrec import _lib
    from amdrec.index import FAISSIndex
    if _PREFILTER != "bf16":
        pytest.skip("one engine is enough")
    idx = FAISSIndex(256, index_type="Flat")
    idx.add(_mk(20_000, 256, 1))
    q = torch.from_numpy(_mk(4, 256, 2)).cuda()
    idx.search_device(q, 10)
    try:
        _lib.profile_enable(True)
        idx.search_device(q, 10)
        idx.search_device(q, 10)
        allp = _lib.profile_report()
        # 4 queries: the threshold is computed inside the corpus pass (no "search_threshold" launch)
        assert {"search_sample_max128x512_bf16", "search_filter_stream128x512_bf16", "search_finalize_mixed",
                "search_fixup"} <= set(allp) and "search_threshold" not in allp
        assert all(v["launches"] == 2 and v["total_ms"] > 0 for v in allp.values())
        _lib.profile_enable(True, only="search_filter")
        idx.search_device(q, 10)
        only = _lib.profile_report()
        assert list(only) == [t for t in allp if t.startswith("search_filter")] and len(only) == 1
        assert next(iter(only.values()))["launches"] == 1
    finally:
        _lib.profile_enable(False)
    assert _lib.profile_report() == {}
