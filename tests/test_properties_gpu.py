"""Size-independent properties of the HIP search (hypothesis-driven shapes on the GPU) and the
corpus build at BASELINE size."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings, strategies as st

import oracle
from amdrec import synth
from tests import cases

pytestmark = pytest.mark.gpu


def _search(xb, xq, k):
    from amdrec.index import FAISSIndex
    idx = FAISSIndex(xb.shape[1], index_type="Flat")
    idx.add(xb)
    return idx.search(xq, k)


@settings(max_examples=25, deadline=None, suppress_health_check=list(HealthCheck))
@given(n=st.integers(1, 30_000), nq=st.integers(1, 140), k=st.integers(1, 600),
       d=st.sampled_from([32, 64, 128, 256]), seed=st.integers(0, 10_000))
def test_topk_matches_oracle_on_random_shapes(n, nq, k, d, seed):
    xb, xq = synth.unit_corpus(n, d, seed=seed), synth.unit_corpus(nq, d, seed=seed + 1)
    ids, D = _search(xb, xq, k)
    rD, rI = oracle.search.flat_ip_search(xb, xq, k)
    oracle.search.check_topk(rD, rI, D, ids if n >= k else np.where(np.isfinite(D), ids, -1), tau=cases.TOPK_TAU,
                             score_tol=cases.SCORE_ATOL)


@settings(max_examples=8, deadline=None, suppress_health_check=list(HealthCheck))
@given(seed=st.integers(0, 1000), n=st.integers(9_000, 40_000))
def test_permutation_invariance_of_scores_and_row_identity(seed, n):
    """Shuffling the corpus permutes the returned positions and leaves the sorted scores unchanged
    (up to fp32 rounding of the same dot products: bit-identical here, the k-order of the sum is fixed)."""
    xb, xq = synth.unit_corpus(n, 128, seed=seed), synth.unit_corpus(17, 128, seed=seed + 7)
    perm = np.random.default_rng(seed).permutation(n)
    ids1, D1 = _search(xb, xq, 200)
    ids2, D2 = _search(xb[perm], xq, 200)
    assert np.array_equal(D1, D2)
    strict = np.r_[True, np.diff(D1[0]) < 0] & np.r_[np.diff(D1[0]) < 0, True]      # rows without a tied neighbour
    assert np.array_equal(perm[ids2[0]][strict], ids1[0][strict])


def test_search_is_deterministic_run_to_run():
    xb, xq = synth.unit_corpus(300_000, 256, seed=5), synth.unit_corpus(512, 256, seed=6)
    a = _search(xb, xq, 500)
    b = _search(xb, xq, 500)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])


def test_corpus_build_1m_rows_through_ad_tower_and_index():
    """§8f #2 / a18: build_faiss_index over 1M ad rows (AdTower streaming in 262144-row passes,
    embeddings never leave the device); spot-check rows against the oracle and self-retrieve."""
    import time
    from amdrec.pipeline import build_faiss_index
    from amdrec.towers import TwoTowerModel
    user, ad, nnum = synth.demo_dims()
    sd = synth.two_tower_state(user, ad, nnum, seed=9)
    m = TwoTowerModel(dict(user), dict(ad), nnum)
    m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    table = synth.ad_features(ad, 1_000_000, seed=10)
    torch.cuda.synchronize()
    t0 = time.time()
    idx = build_faiss_index(m, torch.from_numpy(table).cuda(), index_type="Flat")
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f"corpus build 1M rows: {dt * 1e3:.1f} ms ({0.7209 / dt:.1f} TFLOP/s algorithmic)")
    assert idx.index.ntotal == 1_000_000
    rows = np.r_[0:50, 262_140:262_150, 999_990:1_000_000]
    ref = oracle.search.normalize_l2(oracle.towers.ad_tower(sd, table[rows]))
    got = idx._xb[torch.from_numpy(rows).cuda()].cpu().numpy()
    assert np.abs(got - ref).max() <= cases.EMB_ATOL
    ids, D = idx.search(ref[:8], 3)                        # each row retrieves itself (duplicates may tie)
    assert np.all(np.abs(D[:, 0] - 1.0) <= 1e-5)
    assert all(np.array_equal(table[ids[i, 0]], table[rows[i]]) for i in range(8))


@pytest.mark.parametrize("k_c,top_k", [(500, 10), (100, 10), (500, 32), (7, 10), (500, 64), (600, 10)])
def test_select_topk_both_kernels_ties_nan_and_short_lists(k_c, top_k):
    """amdrec_select_topk (np.argsort(ctr)[::-1][:top_k] of inference.py:263, on logits): order (logit desc, candidate slot
    asc) on ties, NaN logits last, sigmoid of every task at the winners, -1 / 0-probability padding when a user has fewer
    than top_k candidates.  top_k <= 32 with <= 512 candidates takes the one-wave-per-user kernel (no sort), everything
    else the LDS bitonic sort: both must agree with numpy's lexsort."""
    import ctypes as C
    from amdrec import _lib
    lib = _lib.load()
    U, T = 9, 3
    rng = np.random.default_rng(k_c * 131 + top_k)
    logits = rng.standard_normal((T, U * k_c)).astype(np.float32)
    lg = logits[0].reshape(U, k_c)
    lg[1, :] = 0.25                                            # all tied: the lowest slots win, in order
    if k_c > 5:
        lg[2, ::3] = lg[2, 1]                                  # many ties scattered
        lg[3, [0, 4]] = np.nan                                 # NaN ranks last
        lg[4, :] = -np.inf
    logits[0] = lg.reshape(-1)
    cand = rng.permutation(10_000_000)[:U * k_c].reshape(U, k_c).astype(np.int64)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()          # noqa: E731
    L, Cd = d(logits), d(cand)
    ids = torch.full((U, top_k), -7, dtype=torch.int64, device="cuda")
    sc = torch.full((T, U, top_k), -7.0, dtype=torch.float32, device="cuda")
    slots = torch.full((U, top_k), -7, dtype=torch.int32, device="cuda")
    _lib.check(lib.amdrec_select_topk(_lib.ptr(L), L.stride(0), T, 0, _lib.ptr(Cd), U, k_c, top_k, _lib.ptr(ids), _lib.ptr(sc),
                                      _lib.ptr(slots), _lib.stream_ptr(L.device)))
    torch.cuda.synchronize()
    ids, sc, slots = ids.cpu().numpy(), sc.cpu().numpy(), slots.cpu().numpy()
    n = min(k_c, top_k)
    for u in range(U):
        v = lg[u].astype(np.float64)
        key = np.where(np.isnan(v), -np.inf, v)                # NaN last; among NaNs (and -inf) lower slot first
        nan_last = np.isnan(v).astype(np.int64)
        order = np.lexsort((np.arange(k_c), -key, nan_last))[:n]
        assert slots[u, :n].tolist() == order.tolist(), (u, slots[u], order)
        assert ids[u, :n].tolist() == cand[u][order].tolist()
        for t in range(T):
            x = logits[t].reshape(U, k_c)[u][order].astype(np.float64)
            want = 1.0 / (1.0 + np.exp(-x))
            got = sc[t, u, :n]
            assert np.allclose(got[~np.isnan(x)], want[~np.isnan(x)], atol=1e-6)
        assert (ids[u, n:] == -1).all() and (slots[u, n:] == -1).all() and (sc[:, u, n:] == 0).all()
