"""Oracle for the serving pipeline (test infrastructure).

Restates AdRecommenderInference.recommend_ads (inference.py:199-288) from the tensor
level down (feature preprocessing, :160-197, is CPU string work outside the hot path):

    user_emb = UserTower(user_cat, user_num)                 :223-227
    ids, _   = index.search(user_emb, k=stage1_k)            :229-232
    preds    = TransformerRanker(user_cat x500, ad_cat[ids], user_num x500)   :241-255
    top      = argsort(sigmoid(ctr))[::-1][:top_k]           :258-263
    -> ad_ids[top], sigmoid(ctr|engagement|revenue)[top]     :272-288

Documented deviations forced by reference defects:
* candidate ad features are looked up in a real ``ad_cat[N,20]`` table by candidate id
  (the reference draws torch.randint placeholders, inference.py:246-248);
* top-k is selected on the ctr *logit* (monotone under sigmoid) with ties -> lower
  candidate slot first; np.argsort tie order on saturated sigmoids is not reproducible.
"""
from __future__ import annotations

import numpy as np

from . import ranker, search, towers


def sigmoid(x):
    x = np.asarray(x, dtype=np.float32)
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(np.float32)


def select_top(logits, top_k):
    """Indices of the top_k largest logits, (logit desc, slot asc)."""
    logits = np.asarray(logits, dtype=np.float32)
    return np.lexsort((np.arange(len(logits)), -logits.astype(np.float64)))[:top_k]


def recommend(tt_sd, rk_sd, index: "search.FlatIndex", ad_cat_table, user_cat, user_num,
              top_k=10, stage1_k=500, user_chunk=1):
    """Batch version: user_cat [B,6], user_num [B,13] -> list of result dicts.
    ``user_chunk``: users whose candidate rows go through the ranker in ONE forward (the reference's
    batch_recommend is a serial loop of 500-row forwards, inference.py:310-317; rows are independent, so chunking
    only changes how well the CPU's BLAS is fed - used by bench.py's cpu_baseline leg)."""
    user_cat = np.asarray(user_cat)
    user_num = np.asarray(user_num, dtype=np.float32)
    emb = towers.user_tower(tt_sd, user_cat, user_num)
    cand_ids, cand_scores = index.search(emb, k=stage1_k)
    out = []
    B = user_cat.shape[0]
    for b0 in range(0, B, max(1, user_chunk)):
        b1 = min(B, b0 + max(1, user_chunk))
        kk = cand_ids.shape[1]
        logits = ranker.forward(rk_sd,
                                np.repeat(user_cat[b0:b1], kk, axis=0),
                                ad_cat_table[cand_ids[b0:b1].reshape(-1)],
                                np.repeat(user_num[b0:b1], kk, axis=0))
        for b in range(b0, b1):
            ids = cand_ids[b]
            lg = {t: logits[t][(b - b0) * kk:(b - b0 + 1) * kk] for t in ranker.TASKS}
            top = select_top(lg["ctr"], top_k)
            out.append({
                "ad_ids": ids[top].tolist(),
                "candidate_ids": ids, "candidate_scores": cand_scores[b],
                "logits": lg,
                "scores": {t: sigmoid(lg[t][top]).tolist() for t in ranker.TASKS},
            })
    return out
