"""BASELINE configs[0] plumbing, end to end on the GPU: synthetic Criteo-like data (n=10000) ->
fitted preprocessor -> a model_dir on disk (checkpoints in both of the reference's formats, the
default IVF(100,10) index over the 7000 training rows as build_faiss_index does, the ad-feature
table, preprocessor.json) -> AdRecommenderInference(model_dir) -> demo users (inference.py:334-396).
Checked against the CPU oracle given the same centroids."""
import numpy as np
import pytest
import torch

import oracle
from amdrec import synth
from tests import cases

pytestmark = pytest.mark.gpu


def _t(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def test_config0_model_dir_roundtrip_and_demo(tmp_path):
    from amdrec import prep
    from amdrec.pipeline import AdRecommenderInference, USER_COLS, AD_COLS, build_faiss_index
    from amdrec.towers import TwoTowerModel
    numerical, categorical, labels = prep.synthetic_criteo(10000)
    assert 0.15 < labels.mean() < 0.5
    pp, num_scaled, cat_enc = prep.fit_preprocessor(numerical, categorical)
    user_dims = {c: pp.feature_dims[c] for c in USER_COLS}
    ad_dims = {c: pp.feature_dims[c] for c in AD_COLS}
    assert user_dims["C3"] == 100 and ad_dims["C26"] == 10          # cardinalities survive (no class < 10 there)
    n_train = 7000                                                   # 70 % split (train.py prepare_data)
    ad_table = cat_enc[:n_train, 6:]
    tt_sd = synth.two_tower_state(user_dims, ad_dims, 13, seed=51)
    rk_sd = synth.ranker_state(user_dims, ad_dims, 13, seed=52, cross_scale=1.0 / 16)
    d = tmp_path / "models"
    d.mkdir()
    pp.save(d / "preprocessor.json")
    torch.save({"epoch": 3, "model_state_dict": _t(tt_sd), "val_auc": 0.5}, d / "two_tower_best.pt")   # dict form
    torch.save(_t(rk_sd), d / "transformer_ranker_final.pt")                                           # bare form
    np.save(d / "ad_features.npy", ad_table)
    tt = TwoTowerModel(user_dims, ad_dims, 13)
    tt.load_state_dict(_t(tt_sd))
    build_faiss_index(tt, ad_table, save_path=str(d / "faiss_index.bin"))     # IVF nlist=100 nprobe=10 defaults
    rec = AdRecommenderInference(str(d))
    assert rec.faiss_index.index_type == "IVF" and rec.faiss_index.index.ntotal == n_train

    rng = np.random.default_rng(7)
    users = [{"categorical": {f"C{i}": f"cat_{rng.integers(0, 50)}" for i in range(1, 7)},       # :342-351
              "numerical": {f"I{i}": float(rng.random() * 100) for i in range(1, 14)}} for _ in range(21)]
    single = rec.recommend_ads(users[0], top_k=10, stage1_k=500)
    batch = rec.batch_recommend(users[1:], top_k=10, stage1_k=500)
    assert len(batch) == 20 and all(len(r["ad_ids"]) == 10 for r in batch)
    assert all(0 <= i < n_train for r in [single] + batch for i in r["ad_ids"])

    # oracle with the same centroids / assignments
    feats = [rec.preprocess_user_features(u) for u in users]
    uc = torch.cat([f[0] for f in feats]).numpy()
    un = torch.cat([f[1] for f in feats]).numpy()
    corpus = oracle.towers.ad_tower(tt_sd, ad_table)
    cent = rec.faiss_index._ivf.centroids.cpu().numpy()
    assign = rec.faiss_index._ivf.assign.cpu().numpy()
    emb = oracle.search.normalize_l2(oracle.towers.user_tower(tt_sd, uc, un))
    D, I = oracle.search.ivf_search(oracle.search.normalize_l2(corpus), assign, cent, emb, 500, 10)
    agree = 0
    for b, r in enumerate([single] + batch):
        ids = I[b][I[b] >= 0]
        lg = oracle.ranker.forward(rk_sd, np.repeat(uc[b:b + 1], len(ids), 0), ad_table[ids],
                                   np.repeat(un[b:b + 1], len(ids), 0))
        top = oracle.pipeline.select_top(lg["ctr"], 10)
        ref_ids = ids[top].tolist()
        agree += len(set(ref_ids) & set(r["ad_ids"]))
        if ref_ids == r["ad_ids"]:
            assert np.allclose(r["scores"]["ctr"], oracle.pipeline.sigmoid(lg["ctr"][top]), atol=1e-5)
    assert agree >= 0.97 * 10 * len(users)          # only coarse / logit near-ties may differ
