"""End-to-end parity: AdRecommenderInference on the GPU vs oracle.pipeline.recommend (the
restated recommend_ads, inference.py:199-288) on the same seeded models, corpus and users."""
import numpy as np
import pytest
import torch

import oracle
from amdrec import synth
from tests import cases

pytestmark = pytest.mark.gpu


def _t(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def _setup(n_ads, cross_scale, seed=21, index_type="Flat"):
    from amdrec.pipeline import AdRecommenderInference, build_faiss_index
    from amdrec.ranker import TransformerRanker
    from amdrec.towers import TwoTowerModel
    user, ad, nnum = cases.small_dims()
    tt_sd = synth.two_tower_state(user, ad, nnum, seed=seed)
    rk_sd = synth.ranker_state(user, ad, nnum, seed=seed + 1, cross_scale=cross_scale)
    tt = TwoTowerModel(dict(user), dict(ad), nnum)
    tt.load_state_dict(_t(tt_sd))
    rk = TransformerRanker(dict(user), dict(ad), nnum)
    rk.load_state_dict(_t(rk_sd))
    ad_table = synth.ad_features(ad, n_ads, seed=seed + 2)
    index = build_faiss_index(tt, ad_table, index_type=index_type)
    rec = AdRecommenderInference(two_tower_model=tt, transformer_ranker=rk, faiss_index=index,
                                 ad_features=ad_table)
    # oracle side: AdTower corpus -> FlatIndex
    oidx = oracle.search.FlatIndex(256)
    oidx.add(oracle.towers.ad_tower(tt_sd, ad_table))
    return rec, (tt_sd, rk_sd, oidx, ad_table), (user, ad, nnum)


@pytest.mark.parametrize("cross_scale", [1.0 / 16, 1.0])
def test_recommend_matches_oracle_pipeline(cross_scale):
    rec, (tt_sd, rk_sd, oidx, ad_table), (user, ad, nnum) = _setup(4096 * 3, cross_scale)
    B, top_k, k1 = 5, 10, 500
    uc, un = synth.user_batch(user, nnum, B, seed=31)
    out = rec.recommend_device(torch.from_numpy(uc).cuda(), torch.from_numpy(un).cuda(), top_k, k1,
                               check_indices=True)
    ref = oracle.pipeline.recommend(tt_sd, rk_sd, oidx, ad_table, uc, un, top_k, k1)
    cand = out["candidate_ids"].cpu().numpy()
    cs = out["candidate_scores"].cpu().numpy()
    logits = out["logits"].cpu().numpy().reshape(3, B, k1)
    ids = out["ad_ids"].cpu().numpy()
    sc = out["scores"].cpu().numpy()
    for b in range(B):
        r = ref[b]
        # stage 1: tolerance-aware top-500 set + scores
        oracle.search.check_topk(r["candidate_scores"][None], r["candidate_ids"][None], cs[b][None], cand[b][None],
                                 tau=cases.TOPK_TAU, score_tol=2 * cases.SCORE_ATOL)
        # stage 2 logits, compared per candidate id (orders may differ inside near-ties)
        pos_ref = {int(i): j for j, i in enumerate(r["candidate_ids"])}
        common = [j for j, i in enumerate(cand[b]) if int(i) in pos_ref]
        assert len(common) >= k1 - 5
        sel = np.array([pos_ref[int(cand[b][j])] for j in common])
        scale = cases.logit_scale(r["logits"])
        for ti, t in enumerate(oracle.ranker.TASKS):
            ok, err = cases.logit_close(logits[ti, b][common], r["logits"][t][sel],
                                         "randn" if cross_scale == 1.0 else "scaled", scale=scale)
            assert ok, (b, t, err)
        # final top-10 on the GPU's own logits must be the exact (logit desc, slot asc) selection
        top = oracle.pipeline.select_top(logits[0, b], top_k)
        assert np.array_equal(ids[b], cand[b][top])
        for ti in range(3):
            assert np.abs(sc[ti, b] - oracle.pipeline.sigmoid(logits[ti, b][top])).max() <= 1e-6
        # and agree with the oracle's top-10 up to logit near-ties
        miss = set(r["ad_ids"]) - set(ids[b].tolist())
        if miss:
            kth = np.sort(r["logits"]["ctr"])[::-1][top_k - 1]
            for i in miss:
                li = r["logits"]["ctr"][pos_ref[i]]
                assert abs(li - kth) <= max(cases.LOGIT_RTOL * max(1, abs(kth)), cases.LOGIT_SCALE_RTOL * scale) * 2


def test_reference_api_schema_and_preprocessing():
    from amdrec.pipeline import Preprocessor
    rec, _, (user, ad, nnum) = _setup(3000, 1.0 / 16)
    classes = {c: [f"cat_{j}" for j in range(card - 1)] + ["rare"] for c, card in user.items()}
    rec.preprocessor = Preprocessor(classes, [f"I{i}" for i in range(1, 14)], np.full(13, 1.5), np.full(13, 0.7))
    rng = np.random.default_rng(5)
    users = [{"categorical": {f"C{i}": f"cat_{rng.integers(0, 50)}" for i in range(1, 7)},
              "numerical": {f"I{i}": float(rng.random() * 100) for i in range(1, 14)}} for _ in range(4)]
    users[1]["categorical"]["C3"] = "never-seen"           # -> 'rare'
    del users[2]["numerical"]["I7"]                         # -> 0 (inference.py:188)
    r = rec.recommend_ads(users[0])                         # defaults top_k=10, stage1_k=500
    assert set(r) == {"ad_ids", "timing", "scores"}
    assert set(r["timing"]) == {"stage1_ms", "stage2_ms", "total_ms"}
    assert list(r["scores"]) == ["ctr", "engagement", "revenue"]
    assert len(r["ad_ids"]) == 10 and all(isinstance(i, int) for i in r["ad_ids"])
    assert all(len(v) == 10 and all(0.0 <= x <= 1.0 for x in v) for v in r["scores"].values())
    assert r["scores"]["ctr"] == sorted(r["scores"]["ctr"], reverse=True)
    rs = rec.batch_recommend(users, top_k=7, stage1_k=100)
    assert len(rs) == 4 and all(len(x["ad_ids"]) == 7 for x in rs)
    assert rs[0]["ad_ids"][:3] != [] and "scores" not in rec.recommend_ads(users[0], return_scores=False)
    # batch == single (same kernels, deterministic)
    single = rec.batch_recommend([users[3]], top_k=7, stage1_k=100)[0]
    assert single["ad_ids"] == rs[3]["ad_ids"]
    # device feature prep (amdrec_prep_numerical) == the host float64 transform of inference.py:186-195
    cat_d, num_d = rec.preprocess_batch(users)
    host = [rec.preprocess_user_features(u) for u in users]
    assert torch.equal(cat_d.cpu(), torch.cat([h[0] for h in host]))
    assert num_d.dtype == torch.float32 and num_d.is_cuda
    assert (num_d.cpu() - torch.cat([h[1] for h in host])).abs().max().item() <= 5e-6


def test_reference_api_call_is_one_sync_and_still_raises_on_bad_indices():
    """Round 4: recommend_tensors (what recommend_ads / batch_recommend end in) validates indices without a mid-call
    read-back, times its stages with events and brings ids + scores + verdict back in ONE copy.  Same results as the
    device-resident call; an out-of-range user index still raises IndexError (torch.nn.Embedding's behaviour in the
    reference, two_tower_model.py:44) before anything is returned, so does an out-of-range ad-feature table, and the dict
    API skips the per-request check only while the label encoder cannot produce an index outside the models' tables."""
    from amdrec.pipeline import Preprocessor
    rec, _, (user, ad, nnum) = _setup(3000, 1.0 / 16)
    uc_h, un_h = synth.user_batch(user, nnum, 5, seed=77)
    uc, un = torch.from_numpy(uc_h).cuda(), torch.from_numpy(un_h).cuda()
    dev = rec.recommend_device(uc, un, 10, 200)
    res = rec.recommend_tensors(torch.from_numpy(uc_h), torch.from_numpy(un_h), top_k=10, stage1_k=200)   # host tensors in
    assert [r["ad_ids"] for r in res] == dev["ad_ids"].cpu().tolist()
    sc = dev["scores"].cpu().numpy()
    for b, r in enumerate(res):
        assert r["scores"]["ctr"] == sc[0, b].tolist() and r["scores"]["revenue"] == sc[2, b].tolist()
        t = r["timing"]
        assert 0 < t["stage1_ms"] <= t["total_ms"] and abs(t["stage1_ms"] + t["stage2_ms"] - t["total_ms"]) < 1e-9
    bad = uc_h.copy()
    bad[3, 2] = list(user.values())[2]                      # == cardinality: one past the last row
    with pytest.raises(IndexError):
        rec.recommend_tensors(torch.from_numpy(bad), torch.from_numpy(un_h))
    bad[3, 2] = -1
    with pytest.raises(IndexError):
        rec.recommend_tensors(torch.from_numpy(bad), torch.from_numpy(un_h))
    assert rec.recommend_tensors(torch.from_numpy(uc_h), torch.from_numpy(un_h), stage1_k=200)[0]["ad_ids"] == res[0]["ad_ids"]
    # dict API: an encoder with MORE classes than the tables have rows is not trusted - its out-of-table index raises
    classes = {c: [f"cat_{j}" for j in range(card + 3)] for c, card in user.items()}
    rec.preprocessor = Preprocessor(classes, [f"I{i}" for i in range(1, 14)], np.zeros(13), np.ones(13))
    assert not rec._encoder_fits()
    u = {"categorical": {f"C{i}": "cat_1" for i in range(1, 7)}, "numerical": {}}
    assert len(rec.recommend_ads(u)["ad_ids"]) == 10
    u["categorical"]["C2"] = f"cat_{list(user.values())[1] + 2}"
    with pytest.raises(IndexError):
        rec.recommend_ads(u)
    classes = {c: [f"cat_{j}" for j in range(card)] for c, card in user.items()}
    rec.preprocessor = Preprocessor(classes, [f"I{i}" for i in range(1, 14)], np.zeros(13), np.ones(13))
    assert rec._encoder_fits()
    # the ad-feature table is validated once per table, not per request
    rec.ad_features[7, 0] = list(ad.values())[0]
    with pytest.raises(IndexError):
        rec.recommend_tensors(uc, un)


def test_microbatcher_over_the_device_pipeline():
    """Concurrent recommend_ads callers coalesced into batch_recommend passes return what a direct call returns."""
    import threading
    from amdrec.pipeline import Preprocessor
    from amdrec.serving import MicroBatcher
    rec, _, (user, ad, nnum) = _setup(3000, 1.0 / 16)
    classes = {c: [f"cat_{j}" for j in range(card - 1)] + ["rare"] for c, card in user.items()}
    rec.preprocessor = Preprocessor(classes, [f"I{i}" for i in range(1, 14)], np.full(13, 1.5), np.full(13, 0.7))
    rng = np.random.default_rng(9)
    users = [{"categorical": {f"C{i}": f"cat_{rng.integers(0, 50)}" for i in range(1, 7)},
              "numerical": {f"I{i}": float(rng.random() * 100) for i in range(1, 14)}} for _ in range(24)]
    direct = rec.batch_recommend(users, top_k=5, stage1_k=100)
    mb = MicroBatcher(lambda us: rec.batch_recommend(us, top_k=5, stage1_k=100), max_batch=16, max_wait_ms=20)
    out = {}
    ts = [threading.Thread(target=lambda i=i: out.__setitem__(i, mb.recommend_ads(users[i]))) for i in range(24)]
    [t.start() for t in ts]
    [t.join(30) for t in ts]
    mb.close()
    assert all(out[i]["ad_ids"] == direct[i]["ad_ids"] for i in range(24))
    assert sum(mb.batches) == 24 and len(mb.batches) < 24


def test_stage1_is_bitwise_deterministic_run_to_run():
    """The candidate lists are filled by atomics (nondeterministic order): every value that comes out must still be
    independent of that order, bit for bit (the fp32 re-score uses one explicit fma chain in every slot)."""
    rec, _, (user, ad, nnum) = _setup(9000, 1.0 / 16)
    uc, un = synth.user_batch(user, nnum, 5, seed=3)
    uc, un = torch.from_numpy(uc).cuda(), torch.from_numpy(un).cuda()
    ref = None
    for _ in range(12):
        out = rec.recommend_device(uc, un, 10, 500)
        cur = {k: out[k].clone() for k in ("ad_ids", "scores", "candidate_ids", "candidate_scores")}
        if ref is None:
            ref = cur
            continue
        for k in ref:
            assert torch.equal(ref[k].view(torch.int32) if ref[k].dtype == torch.float32 else ref[k],
                               cur[k].view(torch.int32) if cur[k].dtype == torch.float32 else cur[k]), k


def test_hipgraph_replay_equals_eager():
    rec, _, (user, ad, nnum) = _setup(9000, 1.0 / 16)
    for B in (1, 5):
        g = rec.capture(B, top_k=10, stage1_k=500)
        for seed in range(1, 9):
            uc, un = synth.user_batch(user, nnum, B, seed=seed)
            uc, un = torch.from_numpy(uc).cuda(), torch.from_numpy(un).cuda()
            a = rec.recommend_device(uc, un, 10, 500)
            a = {k: v.clone() for k, v in a.items() if isinstance(v, torch.Tensor)}
            b = g(uc, un)
            torch.cuda.synchronize()
            for k in ("ad_ids", "scores", "candidate_ids", "candidate_scores"):
                assert torch.equal(a[k], b[k]), (B, seed, k)
    with pytest.raises(ValueError):
        g(uc[:1].repeat(7, 1), un[:1].repeat(7, 1))
    # a later, larger eager call re-allocates the shared workspace; the graph owns a private one
    big_uc, big_un = synth.user_batch(user, nnum, 64, seed=99)
    rec.recommend_device(torch.from_numpy(big_uc).cuda(), torch.from_numpy(big_un).cuda(), 10, 500)
    a = rec.recommend_device(uc, un, 10, 500)
    a = {k: v.clone() for k, v in a.items() if isinstance(v, torch.Tensor)}
    b = g(uc, un)
    torch.cuda.synchronize()
    assert torch.equal(a["ad_ids"], b["ad_ids"]) and torch.equal(a["candidate_scores"], b["candidate_scores"])


def test_two_stage_retriever_and_benchmark_helper():
    from amdrec.index import benchmark_faiss_index
    from amdrec.pipeline import TwoStageRetriever
    rec, (tt_sd, rk_sd, oidx, ad_table), (user, ad, nnum) = _setup(5000, 1.0 / 16)
    r = TwoStageRetriever(rec.two_tower_model, rec.transformer_ranker, rec.faiss_index)
    uc, un = synth.user_batch(user, nnum, 1, seed=8)
    ids1, d1 = r.retrieve_and_rank(torch.from_numpy(uc), torch.from_numpy(un), stage1_k=100)       # stage 1 only
    assert len(ids1) == 100 == len(d1) and d1 == sorted(d1, reverse=True)
    ids, scores = r.retrieve_and_rank(torch.from_numpy(uc), torch.from_numpy(un), stage1_k=500, stage2_k=10,
                                      ad_features_lookup=ad_table)
    ref = oracle.pipeline.recommend(tt_sd, rk_sd, oidx, ad_table, uc, un, 10, 500)[0]
    assert len(ids) == 10 and len(set(ids) & set(ref["ad_ids"])) >= 9
    assert np.allclose(sorted(scores, reverse=True), scores) and all(0 <= s <= 1 for s in scores)
    res = benchmark_faiss_index(dimension=64, num_vectors=20000, num_queries=10, k=10)
    assert set(res) == {"Flat", "IVF"} and all(set(v) == {"add_time", "search_time_ms", "per_query_ms"} for v in res.values())


def test_index_save_load_roundtrip(tmp_path):
    rec, (tt_sd, rk_sd, oidx, ad_table), (user, ad, nnum) = _setup(2000, 1.0 / 16)
    from amdrec.index import FAISSIndex
    p = str(tmp_path / "m" / "faiss_index.bin")
    rec.faiss_index.save(p)
    idx2 = FAISSIndex(256, index_type="Flat")
    idx2.load(p)
    q = synth.unit_corpus(3, 256, seed=3)
    a, b = rec.faiss_index.search(q, 50), idx2.search(q, 50)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert idx2.get_stats() == rec.faiss_index.get_stats()


def test_tutorial_architecture_end_to_end_matches_oracle_pipeline():
    """the whole path at the tutorial's architecture (tutorial.ipynb cells 10, 19, 26: Flat index over
    128-dimensional embeddings, d_model-128 ranker) against oracle.pipeline - search kernels at d = 128, ad projection
    cache at d_model 128, the generic ranker path, top-10."""
    from amdrec.pipeline import AdRecommenderInference, build_faiss_index
    from amdrec.ranker import TransformerRanker
    from amdrec.towers import TwoTowerModel
    user, ad, nnum = cases.small_dims()
    a = cases.arch("tutorial")
    tt_sd = synth.two_tower_state(user, ad, nnum, seed=61, **a["tt"])
    rk_sd = synth.ranker_state(user, ad, nnum, seed=62, cross_scale=1.0 / 16, **a["rk"])
    tt = TwoTowerModel(dict(user), dict(ad), nnum, **a["tt"])
    tt.load_state_dict(_t(tt_sd))
    rk = TransformerRanker(dict(user), dict(ad), nnum, **a["rk"])
    rk.load_state_dict(_t(rk_sd))
    n_ads, B, top_k, k1 = 20_000, 6, 10, 500
    ad_table = synth.ad_features(ad, n_ads, seed=63)
    index = build_faiss_index(tt, ad_table, index_type="Flat")
    assert index.dimension == 128
    rec = AdRecommenderInference(two_tower_model=tt, transformer_ranker=rk, faiss_index=index, ad_features=ad_table)
    assert rk.x3_fallback_reason() == "d_model 128 != 256" and rk.gemm_engine_for(B * k1) == "fp32"
    oidx = oracle.search.FlatIndex(128)
    oidx.add(oracle.towers.ad_tower(tt_sd, ad_table))
    uc, un = synth.user_batch(user, nnum, B, seed=64)
    out = rec.recommend_device(torch.from_numpy(uc).cuda(), torch.from_numpy(un).cuda(), top_k, k1, check_indices=True)
    ref = oracle.pipeline.recommend(tt_sd, rk_sd, oidx, ad_table, uc, un, top_k, k1)
    cand, cs = out["candidate_ids"].cpu().numpy(), out["candidate_scores"].cpu().numpy()
    logits = out["logits"].cpu().numpy().reshape(3, B, k1)
    ids, sc = out["ad_ids"].cpu().numpy(), out["scores"].cpu().numpy()
    for b in range(B):
        r = ref[b]
        oracle.search.check_topk(r["candidate_scores"][None], r["candidate_ids"][None], cs[b][None], cand[b][None],
                                 tau=cases.TOPK_TAU, score_tol=2 * cases.SCORE_ATOL)
        pos_ref = {int(i): j for j, i in enumerate(r["candidate_ids"])}
        common = [j for j, i in enumerate(cand[b]) if int(i) in pos_ref]
        assert len(common) >= k1 - 5
        sel = np.array([pos_ref[int(cand[b][j])] for j in common])
        for ti, t in enumerate(oracle.ranker.TASKS):
            ok, err = cases.logit_close(logits[ti, b][common], r["logits"][t][sel], "scaled")
            assert ok, (b, t, err)
        # the GPU's winners are the winners of its own logits, and the oracle's unless the 10th place is a near-tie
        own = cand[b][oracle.pipeline.select_top(logits[0, b], top_k)]
        assert np.array_equal(own, ids[b])
        miss = set(r["ad_ids"]) - set(ids[b].tolist())
        if miss:
            kth = np.sort(r["logits"]["ctr"])[::-1][top_k - 1]
            for i in miss:
                assert abs(r["logits"]["ctr"][pos_ref[i]] - kth) <= 2 * cases.LOGIT_STRICT_RTOL * max(1, abs(kth))
        assert np.all((sc[:, b] > 0) & (sc[:, b] < 1))
    # the reference API on top of it (dict of strings in, lists out) needs a preprocessor: covered by test_config0_gpu
