"""IVF-Flat on the GPU: (i) the scan is exact over the probed lists given this build's centroids
and assignments (vs oracle.search.ivf_search), (ii) recall@k vs the Flat result, (iii) the
FAISSIndex drop-in defaults (index_type='IVF', nlist=100, nprobe=10: faiss_retrieval.py:20-25)."""
import numpy as np
import pytest
import torch

import oracle
from amdrec import synth
from tests import cases

pytestmark = pytest.mark.gpu


def _clustered(n, d, n_clusters, seed, spread=0.35):
    rng = np.random.default_rng(seed)
    c = rng.standard_normal((n_clusters, d)).astype(np.float32)
    x = c[rng.integers(0, n_clusters, n)] + spread * rng.standard_normal((n, d)).astype(np.float32)
    return x / np.linalg.norm(x, axis=1, keepdims=True)


@pytest.mark.parametrize("n,nlist,nprobe,k,nq", [(20_000, 100, 10, 100, 33), (50_000, 256, 16, 500, 64),
                                                 (5_000, 64, 64, 50, 5), (30_000, 37, 5, 200, 300),
                                                 # round 4: lists of ~1250 rows and of ~5000 rows - a workgroup of the grouped
                                                 # scan walks several row tiles of its list, 128-row tiles (longest list
                                                 # <= 1536 rows) and 256-row tiles
                                                 (40_000, 32, 8, 300, 64), (80_000, 16, 4, 500, 128)])
def test_ivf_scan_is_exact_given_centroids(n, nlist, nprobe, k, nq):
    from amdrec.index import FAISSIndex
    xb = _clustered(n, 256, 40, 1)
    xq = _clustered(nq, 256, 40, 2)
    idx = FAISSIndex(256, index_type="IVF", nlist=nlist, nprobe=nprobe)
    idx.add(xb)
    assert idx.index.is_trained and idx.index.ntotal == n and idx.index.nprobe == nprobe
    ids, D = idx.search(xq, k)
    cent = idx._ivf.centroids.cpu().numpy()
    assign = idx._ivf.assign.cpu().numpy()
    xbn, xqn = oracle.search.normalize_l2(xb), oracle.search.normalize_l2(xq)
    # the oracle with ITS OWN coarse selection (probes may differ only in coarse near-ties)
    rD, rI = oracle.search.ivf_search(xbn, assign, cent, xqn, k, nprobe)
    overlap = np.mean([len(set(a) & set(b)) / k for a, b in zip(ids, rI)])
    assert overlap >= 0.98                         # differences: coarse near-ties (a different list probed)
    assert np.abs(D[:, 0] - rD[:, 0]).max() <= cases.SCORE_ATOL
    # the scan in isolation: feed the GPU's probes to the oracle -> must match for every query
    cs = torch.empty((nq, nprobe), dtype=torch.float32, device="cuda")
    pr = torch.empty((nq, nprobe), dtype=torch.int64, device="cuda")
    from amdrec.index import flat_search
    flat_search(idx._ivf.centroids, nlist, torch.from_numpy(xqn).cuda(), nprobe, cs, pr)
    rD2, rI2 = oracle.search.ivf_search(xbn, assign, cent, xqn, k, nprobe, probes=pr.cpu().numpy())
    oracle.search.check_topk(rD2, rI2, D, ids, tau=cases.TOPK_TAU, score_tol=cases.SCORE_ATOL)
    assert (ids == rI2).mean() > 0.98              # same sets; order may differ inside fp32 near-ties


def test_ivf_full_probe_equals_flat_and_recall_grows_with_nprobe():
    from amdrec.index import FAISSIndex
    n, k = 60_000, 500
    xb, xq = _clustered(n, 256, 64, 3), _clustered(48, 256, 64, 4)
    flat = FAISSIndex(256, index_type="Flat")
    flat.add(xb)
    fids, fD = flat.search(xq, k)
    ivf = FAISSIndex(256, index_type="IVF", nlist=128, nprobe=128)
    ivf.add(xb)
    ids, D = ivf.search(xq, k)                       # probing every list == exhaustive search
    oracle.search.check_topk(fD, fids, D, ids, tau=cases.TOPK_TAU, score_tol=cases.SCORE_ATOL)
    recalls = []
    for nprobe in (1, 4, 16, 64):
        ivf.index.nprobe = nprobe                    # the reference sets index.nprobe before searching (:150-151)
        ids, _ = ivf.search(xq, k)
        recalls.append(np.mean([len(set(a) & set(b)) / k for a, b in zip(ids, fids)]))
    assert recalls == sorted(recalls) and recalls[-1] > 0.9
    print("recall@500 vs Flat by nprobe (1,4,16,64):", [round(r, 3) for r in recalls])


def test_ivf_defaults_underfill_and_save_load(tmp_path):
    from amdrec.index import FAISSIndex
    xb, xq = _clustered(3_000, 256, 20, 5), _clustered(4, 256, 20, 6)
    idx = FAISSIndex(256)                             # reference defaults: IVF, nlist=100, nprobe=10
    assert idx.index_type == "IVF" and not idx.index.is_trained
    idx.add(xb[:2000])
    idx.add(xb[2000:], ad_ids=list(range(10_000, 11_000)))    # second add: no re-training, custom ids
    ids, D = idx.search(xq, 500)                      # ~300 rows in 10 lists < k=500 -> under-filled like faiss
    filled = np.isfinite(D)
    assert filled.sum(axis=1).min() < 500 and np.all(ids[~filled] == idx.id_map[-1])   # id_map[-1] (:159)
    assert set(ids[filled].tolist()) <= set(range(2000)) | set(range(10_000, 11_000))
    p = str(tmp_path / "ivf.bin")
    idx.save(p)
    idx2 = FAISSIndex(256)
    idx2.load(p)
    ids2, D2 = idx2.search(xq, 500)
    assert np.array_equal(ids, ids2) and np.array_equal(D, D2)
    assert idx2.get_stats() == idx.get_stats()


def test_assign_is_argmax_and_training_is_bit_reproducible():
    """amdrec_ivf_assign == arg max_c <x, c> (ties -> lower centroid), one amdrec_ivf_kmeans_step == assign + mean +
    renormalise computed in numpy float64, and two trainings of the same data give IDENTICAL centroids (the centroid
    sums are order-independent fixed-point integer atomics)."""
    from amdrec import _lib, ivf
    rng = np.random.default_rng(7)
    x = _clustered(40_000, 256, 30, 7)
    cent = x[rng.choice(len(x), 200, replace=False)].copy()
    cent[13] = cent[5]                                               # two identical centroids: exact ties
    xd, cd = torch.from_numpy(x).cuda(), torch.from_numpy(cent).cuda()
    got = ivf._assign(xd, cd).cpu().numpy()
    sc = x.astype(np.float64) @ cent.astype(np.float64).T
    ref = np.argmax(sc, axis=1)                                      # first maximum = lower index
    bad = got != ref
    # fp32 vs float64 may disagree only where the two best scores are within fp32 rounding of each other
    assert bad.mean() < 1e-3 and (np.abs(sc[bad, got[bad]] - sc[bad, ref[bad]]) <= 2e-6).all()
    assert not (got == 13).any()                                     # the duplicate never wins a tie against its lower twin
    # one Lloyd step
    lib = _lib.load()
    c1 = cd.clone()
    nbytes = _lib.C.c_size_t(0)
    _lib.check(lib.amdrec_ivf_kmeans_workspace(len(x), 256, 200, _lib.C.byref(nbytes)))
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device="cuda")
    _lib.check(lib.amdrec_ivf_kmeans_step(_lib.ptr(xd), len(x), 256, 256, _lib.ptr(c1), 200, 256, _lib.ptr(ws), ws.numel(),
                                          _lib.stream_ptr(xd.device)))
    exp = cent.astype(np.float64).copy()
    for c in range(200):
        m = x[got == c]
        if len(m):
            s = m.astype(np.float64).sum(axis=0)
            exp[c] = s / np.linalg.norm(s)
    assert np.abs(c1.cpu().numpy() - exp).max() <= 2e-7
    a = ivf.IVFState.train(xd, 128).centroids
    b = ivf.IVFState.train(xd.clone(), 128).centroids
    assert torch.equal(a, b)


def test_two_shards_sharing_centroids_equal_the_unsharded_ivf_bit_for_bit():
    """SURVEY.md section 8e for IVF: ranks share the trained centroids and hold their slice of every list; per-shard
    searches merged by amdrec_topk_merge == the unsharded IVF search, exactly (positions and score bits)."""
    from amdrec.index import FAISSIndex
    from amdrec.sharded import HipEngine, packed_layout
    n, nq, k, nlist, nprobe = 50_000, 40, 200, 128, 9
    xb, xq = _clustered(n, 256, 50, 11), _clustered(nq, 256, 50, 12)
    full = FAISSIndex(256, index_type="IVF", nlist=nlist, nprobe=nprobe)
    full.add(xb)
    q = torch.from_numpy(oracle.search.normalize_l2(xq)).cuda()
    ref_pos, ref_sc = full.search_device(q, k, normalize=False, return_positions=True)
    s_bytes, chunk = packed_layout(nq, k)
    G = 2
    gathered = torch.empty(chunk * G, dtype=torch.uint8, device="cuda")
    cuts = [0, 21_337, n]                                              # uneven shards
    for g in range(G):
        sh = FAISSIndex(256, index_type="IVF", nlist=nlist, nprobe=nprobe)
        sh.set_trained_centroids(full.centroids)
        sh.add(xb[cuts[g]:cuts[g + 1]])
        pos, sc = sh.search_device(q, k, normalize=False, return_positions=True, pos_offset=cuts[g])
        c = gathered[g * chunk:(g + 1) * chunk]
        c[:nq * k * 4].view(torch.float32).copy_(sc.reshape(-1))
        c[s_bytes:].view(torch.int32).copy_(pos.reshape(-1))
    sc, pos = HipEngine(None, 0).merge(gathered, G, nq, k, 0, nq)
    assert torch.equal(pos, ref_pos) and torch.equal(sc, ref_sc)
    # and the unsharded result is the oracle's given the same centroids / probes
    cent = full.centroids.cpu().numpy()
    assign = full._ivf.assign.cpu().numpy()
    rD, rI = oracle.search.ivf_search(oracle.search.normalize_l2(xb), assign, cent, q.cpu().numpy(), k, nprobe)
    agree = np.mean([len(set(a) & set(b)) / k for a, b in zip(ref_pos.cpu().numpy(), rI)])
    assert agree >= 0.98                                               # differences: coarse near-ties only


def test_ivf_pipeline_captures_in_a_hip_graph():
    """an IVF search used to call torch.bincount (a host sync) per query batch, which is illegal during
    stream capture; the grouping is now kernels (amdrec_ivf_group).  Capture + replay == eager, for a batch on the
    per-pair scan (B = 4) and one on the grouped MFMA scan (B = 32)."""
    from tests.test_pipeline_gpu import _setup
    rec, _, (user, ad, nnum) = _setup(20_000, 1.0 / 16, index_type="IVF")
    assert rec.faiss_index.index_type == "IVF"
    for B in (4, 32):
        uc, un = synth.user_batch(user, nnum, B, seed=50 + B)
        uc, un = torch.from_numpy(uc).cuda(), torch.from_numpy(un).cuda()
        eager = rec.recommend_device(uc, un, 10, 200)
        ids, sc = eager["ad_ids"].clone(), eager["scores"].clone()
        g = rec.capture(B, 10, 200)
        out = g(uc, un)
        torch.cuda.synchronize()
        assert torch.equal(out["ad_ids"], ids) and torch.equal(out["scores"], sc)


@pytest.mark.parametrize("copies", [1500, 6000])
def test_ivf_select_boundary_ties(copies):
    """`copies` identical rows tie at the k-th score (k = 500): 1500 ties still fit the select's gather (k-th SCORE by a
    3-pass radix select, ties settled by the 64-bit sort), 6000 take its fallback (6-pass select of the exact k-th key).
    Either way the result is the lowest positions among the tied rows, as the oracle's order rule says."""
    from amdrec.index import FAISSIndex, flat_search
    n, nlist, nprobe, k, nq = 20_000, 32, 32, 500, 3
    xb = _clustered(n, 256, 10, 5)
    rng = np.random.default_rng(6)
    where = np.sort(rng.choice(n, copies, replace=False))
    xb[where] = xb[where[0]]                                   # identical rows, scattered over the corpus
    xq = np.stack([xb[where[0]] + 0.5 * _clustered(1, 256, 10, 7 + i)[0] for i in range(nq)])
    # a few hundred rows score above the copies for every query, so the copies straddle rank 500
    idx = FAISSIndex(256, index_type="IVF", nlist=nlist, nprobe=nprobe)       # all lists probed: == exact search
    idx.add(xb)
    ids, D = idx.search(xq, k)
    xbn, xqn = oracle.search.normalize_l2(xb), oracle.search.normalize_l2(xq)
    cs = torch.empty((nq, nprobe), dtype=torch.float32, device="cuda")
    pr = torch.empty((nq, nprobe), dtype=torch.int64, device="cuda")
    flat_search(idx._ivf.centroids, nlist, torch.from_numpy(xqn).cuda(), nprobe, cs, pr)
    rD, rI = oracle.search.ivf_search(xbn, idx._ivf.assign.cpu().numpy(), idx._ivf.centroids.cpu().numpy(), xqn, k, nprobe,
                                      probes=pr.cpu().numpy())
    tied = set(where.tolist())
    for q in range(nq):
        got_t = [i for i in ids[q] if int(i) in tied]
        ref_t = [i for i in rI[q] if int(i) in tied]
        assert 0 < len(ref_t) < copies                          # the copies do straddle the boundary
        assert got_t == ref_t == sorted(ref_t)                  # the tied rows that made it: the lowest positions, in order
    oracle.search.check_topk(rD, rI, D, ids, tau=cases.TOPK_TAU, score_tol=cases.SCORE_ATOL)


@pytest.mark.parametrize("n,nlist,k,copies", [(6_000, 64, 900, 3),       # phase-1 pool (2 lists) smaller than k for some queries
                                              (40_000, 64, 300, 4)])     # phase-1 pool larger than k for most queries: tau is a real score
def test_two_phase_grouped_scan_with_ties_at_tau(n, nlist, k, copies):
    """the exact two-phase scan (>= 16 queries take the grouped scan, >= 16 probes split it: the nearest eighth
    of the probes unfiltered -> tau = that subset's k-th score -> the other probes keep rows with score >= tau) on a corpus
    of duplicated rows: every row `copies` times, so rows tie AT tau and across the two phases; with the small corpus the
    first phase's pool is smaller than k (tau = -inf: the filter must keep everything).  Result == the oracle's given the
    same probes: tolerance-aware on scores, and the tied rows that made the cut are the lowest positions."""
    from amdrec.index import FAISSIndex, flat_search
    from amdrec import ivf
    nq, nprobe = 24, 16
    assert ivf.use_grouped_scan(nq, nprobe, nlist) and nprobe >= ivf.TWO_PHASE_MIN_PROBES
    base = _clustered(n // copies, 256, 12, 21)
    xb = np.concatenate([base] * copies)                        # copy j of row i at position j * (n // copies) + i
    xq = _clustered(nq, 256, 12, 22)
    idx = FAISSIndex(256, index_type="IVF", nlist=nlist, nprobe=nprobe)
    idx.add(xb)
    ids, D = idx.search(xq, k)
    xbn, xqn = oracle.search.normalize_l2(xb), oracle.search.normalize_l2(xq)
    cs = torch.empty((nq, nprobe), dtype=torch.float32, device="cuda")
    pr = torch.empty((nq, nprobe), dtype=torch.int64, device="cuda")
    flat_search(idx._ivf.centroids, nlist, torch.from_numpy(xqn).cuda(), nprobe, cs, pr)
    rD, rI = oracle.search.ivf_search(xbn, idx._ivf.assign.cpu().numpy(), idx._ivf.centroids.cpu().numpy(), xqn, k, nprobe,
                                      probes=pr.cpu().numpy())
    oracle.search.check_topk(rD, rI, D, ids, tau=cases.TOPK_TAU, score_tol=cases.SCORE_ATOL)
    m = n // copies
    for q in range(nq):
        valid = ids[q][ids[q] >= 0]
        assert len(set(valid.tolist())) == len(valid)           # a row filed in one phase is not appended again by the other
        # identical copies carry bit-identical scores: among the copies of one base row the lower positions come first
        for i in set((valid % m).tolist()):
            got = [int(p) for p in valid if p % m == i]
            assert got == sorted(got)
    lens = torch.bincount(idx._ivf.assign, minlength=nlist)
    pool1 = lens[pr[:, :max(2, nprobe // 8)]].sum(1)           # rows of every query's first-phase probes
    if n == 6_000:
        assert bool((pool1 < k).any())                          # tau = -inf for those queries: the filter keeps everything
    else:
        assert bool((pool1 >= k).any())                         # tau is a real score there, with copies tying at it


@pytest.mark.parametrize("nlist,nprobe,nq", [(37, 5, 300), (4096, 64, 70), (1000, 1000, 3), (257, 16, 33), (100, 10, 1)])
def test_coarse_key_table_gives_the_exact_search_probes(nlist, nprobe, nq):
    """amdrec_ivf_coarse_keys + amdrec_ivf_select == amdrec_flat_search over the centroid table, bit for bit (scores and
    centroid ids, ties towards the lower id): the IndexFlatIP quantizer of IndexIVFFlat (faiss_retrieval.py:118)."""
    from amdrec import _lib
    from amdrec.index import flat_search
    lib = _lib.load()
    g = torch.Generator().manual_seed(nlist + nq)
    cent = torch.nn.functional.normalize(torch.randn(nlist, 256, generator=g), dim=1)   # unit rows: SCORE_ATOL applies
    cent[nlist // 2] = cent[0]                       # a duplicated centroid: equal scores, the lower id first
    cent = cent.cuda()
    q = torch.nn.functional.normalize(torch.randn(nq, 256, generator=g), dim=1).cuda()
    cs = torch.empty((nq, nprobe), dtype=torch.float32, device="cuda")
    pr = torch.empty((nq, nprobe), dtype=torch.int64, device="cuda")
    flat_search(cent, nlist, q, nprobe, cs, pr)
    ld = (nlist + 1) // 2 * 2
    keys = torch.empty((nq, ld), dtype=torch.int64, device="cuda")
    cnt = torch.full((nq,), nlist, dtype=torch.int64, device="cuda")
    cs2, pr2 = torch.empty_like(cs), torch.empty_like(pr)
    _lib.check(lib.amdrec_ivf_coarse_keys(_lib.ptr(cent), nlist, cent.stride(0), 256, _lib.ptr(q), nq, q.stride(0),
                                          _lib.ptr(keys), ld, _lib.stream_ptr(cent.device)))
    _lib.check(lib.amdrec_ivf_select(_lib.ptr(keys), ld, _lib.ptr(cnt), nq, nprobe, _lib.ptr(cs2), _lib.ptr(pr2),
                                     _lib.stream_ptr(cent.device)))
    torch.cuda.synchronize()
    assert torch.equal(cs, cs2)
    assert torch.equal(pr, pr2)
    ref = (q.double() @ cent.double().T).cpu().numpy()
    assert np.abs(np.sort(ref, axis=1)[:, ::-1][:, :nprobe] - cs2.cpu().numpy()).max() <= cases.SCORE_ATOL


@pytest.mark.parametrize("nq,n,k,slices", [(1, 100_000, 500, 24), (3, 37_001, 500, 9), (2, 5_000, 500, 16), (5, 300, 500, 4),
                                           (2, 70_000, 10, 64),
                                           # fewer than k rows in a pool whose zero-padded partial lists exceed the
                                           # 2048-key sort buffer (slices * k > 2048) - the padding must not count as keys
                                           (2, 300, 500, 16), (1, 1500, 2048, 8)])
def test_split_select_equals_the_single_workgroup_select(nq, n, k, slices):
    """amdrec_ivf_select_split (several workgroups per query, partial lists + a ticket) == amdrec_ivf_select, bit for bit:
    pools with duplicated scores (ties broken by position), ragged pool sizes, fewer keys than k, repeated calls on the same
    tickets (the kernel leaves them zero)."""
    from amdrec import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(n + k)
    sc = torch.randn(nq, n, generator=g)
    sc[:, ::7] = sc[:, 1::7][:, :sc[:, ::7].shape[1]]          # exact score ties at different positions
    pos = torch.stack([torch.randperm(n, generator=g) for _ in range(nq)])
    u = sc.view(torch.int32).to(torch.int64) & 0xFFFFFFFF
    u = torch.where(u >= 0x80000000, (~u) & 0xFFFFFFFF, u | 0x80000000)              # order-preserving image of the score
    keys = ((u << 32) | ((~pos) & 0xFFFFFFFF)).cuda()                                # make_key(score, position)
    cnt = torch.tensor([n - 13 * i for i in range(nq)], dtype=torch.int64).clamp(min=1).cuda()
    ref_s, ref_p = torch.empty((nq, k), device="cuda"), torch.empty((nq, k), dtype=torch.int64, device="cuda")
    _lib.check(lib.amdrec_ivf_select(_lib.ptr(keys), n, _lib.ptr(cnt), nq, k, _lib.ptr(ref_s), _lib.ptr(ref_p),
                                     _lib.stream_ptr(keys.device)))
    ws = torch.empty(nq * slices * k * 8, dtype=torch.uint8, device="cuda")
    tickets = torch.zeros(nq, dtype=torch.int32, device="cuda")
    for _ in range(2):
        out_s, out_p = torch.full_like(ref_s, float("nan")), torch.full_like(ref_p, -7)
        _lib.check(lib.amdrec_ivf_select_split(_lib.ptr(keys), n, _lib.ptr(cnt), nq, k, slices, _lib.ptr(out_s), _lib.ptr(out_p),
                                               _lib.ptr(ws), ws.numel(), _lib.ptr(tickets), _lib.stream_ptr(keys.device)))
        torch.cuda.synchronize()
        assert torch.equal(out_p, ref_p) and torch.equal(out_s, ref_s)
        assert int(tickets.abs().sum()) == 0
    # against a plain sort of the first query's pool
    c0 = int(cnt[0])
    flipped = keys[0, :c0].cpu() ^ torch.iinfo(torch.int64).min      # unsigned 64-bit order as signed order
    order = torch.argsort(flipped, descending=True, stable=True)[:k]
    exp = pos[0][order]
    have = min(k, c0)
    assert torch.equal(ref_p[0, :have].cpu(), exp[:have])


@pytest.mark.parametrize("dim", [64, 96, 136])
def test_bf16_prefiltered_second_phase_at_other_dimensions_and_with_nan_rows(dim, monkeypatch):
    """The prefiltered scan at dimensions that are not a whole number of 64-element bf16 K-steps (96, 136: the last step is
    masked) and with NaN / inf rows in the lists: a NaN bf16 score is nominated and the fp32 re-score decides (a NaN row ranks
    last, as in the fp32 scan).  Forced on vs forced off: same result."""
    from amdrec.index import FAISSIndex
    n, nlist, nprobe, k, nq = 30_000, 64, 32, 200, 80
    xb, xq = _clustered(n, dim, 40, 21), _clustered(nq, dim, 40, 22)
    xb[5] = np.nan
    xb[77, 3] = np.inf
    idx = FAISSIndex(dim, index_type="IVF", nlist=nlist, nprobe=nprobe)
    idx.train(xb[100:])                                      # (k-means on the clean rows)
    idx.add(xb)
    monkeypatch.setenv("AMDREC_IVF_MIXED", "1")
    ids_m, D_m = idx.search(xq, k)
    monkeypatch.setenv("AMDREC_IVF_MIXED", "0")
    ids_f, D_f = idx.search(xq, k)
    fin = np.isfinite(D_f)
    assert np.array_equal(np.isfinite(D_m), fin)
    oracle.search.check_topk(np.where(fin, D_f, -np.inf), ids_f, np.where(fin, D_m, -np.inf), ids_m, tau=cases.TOPK_TAU,
                             score_tol=cases.SCORE_ATOL)
    assert not np.isin(ids_m[fin], [5, 77]).any()


@pytest.mark.parametrize("n,nlist,nprobe,k,nq", [(60_000, 128, 32, 100, 96), (40_000, 32, 16, 300, 64), (200_000, 64, 16, 500, 40),
                                                 # first phase shorter than k (8 lists of ~39 rows): tau = -inf, EVERY row of the second
                                                 # phase is nominated - the nomination list of a tile overflows into the per-lane path
                                                 (20_000, 512, 64, 500, 128)])
def test_bf16_prefiltered_second_phase_equals_the_fp32_scan(n, nlist, nprobe, k, nq, monkeypatch):
    """Round 4: the second phase of the two-phase grouped scan nominates rows on a bf16 shadow of the lists (bf16 MFMA) and
    re-scores the nominated rows in fp32 (csrc/ivf.hip EpiIvfPrefilter).  Every row the fp32 filter keeps must be kept:
    the result equals the fp32-only scan's - same ids wherever the scores are not fp32 near-ties at the k-th place - and the
    oracle's given the probes; the profile shows which scan ran."""
    from amdrec import _lib, ivf
    from amdrec.index import FAISSIndex
    xb, xq = _clustered(n, 256, 50, 11), _clustered(nq, 256, 50, 12)
    idx = FAISSIndex(256, index_type="IVF", nlist=nlist, nprobe=nprobe)
    idx.add(xb)
    assert ivf.use_grouped_scan(nq, nprobe, nlist) and nprobe >= ivf.TWO_PHASE_MIN_PROBES
    monkeypatch.setenv("AMDREC_IVF_MIXED", "1")            # (the default picks it by the first phase's rows per wanted result)
    _lib.profile_enable(True)
    ids_m, D_m = idx.search(xq, k)
    tags_m = set(_lib.profile_report())
    monkeypatch.setenv("AMDREC_IVF_MIXED", "0")
    _lib.profile_enable(True)
    ids_f, D_f = idx.search(xq, k)
    tags_f = set(_lib.profile_report())
    _lib.profile_enable(False)
    assert any(t.startswith("ivf_scan_grouped_bf16") for t in tags_m) and "ivf_filter_bounds" in tags_m
    assert not any(t.startswith("ivf_scan_grouped_bf16") for t in tags_f)
    # same result up to fp32 near-ties at the boundary (a re-scored row's sum has another order than the MFMA's)
    oracle.search.check_topk(D_f, ids_f, D_m, ids_m, tau=cases.TOPK_TAU, score_tol=cases.SCORE_ATOL)
    assert (ids_m == ids_f).mean() > 0.995
    if n == 20_000:                                         # the case is what its comment says
        assert (n // nlist) * max(2, nprobe // 8) < k
    # and exact given the probes
    cent, assign = idx._ivf.centroids.cpu().numpy(), idx._ivf.assign.cpu().numpy()
    xbn, xqn = oracle.search.normalize_l2(xb), oracle.search.normalize_l2(xq)
    cs = torch.empty((nq, nprobe), dtype=torch.float32, device="cuda")
    pr = torch.empty((nq, nprobe), dtype=torch.int64, device="cuda")
    from amdrec.index import flat_search
    flat_search(idx._ivf.centroids, nlist, torch.from_numpy(xqn).cuda(), nprobe, cs, pr)
    rD, rI = oracle.search.ivf_search(xbn, assign, cent, xqn, k, nprobe, probes=pr.cpu().numpy())
    oracle.search.check_topk(rD, rI, D_m, ids_m, tau=cases.TOPK_TAU, score_tol=cases.SCORE_ATOL)
