"""Oracle for the two-tower encoders (test infrastructure, see oracle/__init__.py).

Restates, in numpy float32, eval-mode forward of
* EmbeddingLayer.forward      two_tower_model.py:33-49
* UserTower.forward           two_tower_model.py:98-121 (mlp built :83-95)
* AdTower.forward             two_tower_model.py:167-184
* TwoTowerModel.predict_scores two_tower_model.py:287-304
"""
from __future__ import annotations

import numpy as np

BN_EPS = 1e-5       # nn.BatchNorm1d default (two_tower_model.py:86)
NORM_EPS = 1e-12    # F.normalize default (two_tower_model.py:119)


def embed(sd, prefix, cat):
    """EmbeddingLayer.forward: column i <-> i-th key in insertion order, concat."""
    names = [k for k in sd if k.startswith(prefix) and k.endswith(".weight")]
    cat = np.asarray(cat).astype(np.int64)
    assert cat.shape[1] == len(names), (cat.shape, len(names))
    cols = []
    for i, k in enumerate(names):
        table = sd[k]
        idx = cat[:, i]
        if idx.size and (idx.min() < 0 or idx.max() >= table.shape[0]):
            raise IndexError("index out of range in self")   # torch's message
        cols.append(table[idx])
    return np.concatenate(cols, axis=1).astype(np.float32)


def _mlp(sd, prefix, x):
    """Sequential(Linear, BN, ReLU, Dropout)*n + Linear in eval mode."""
    idx = 0
    while f"{prefix}.{idx + 1}.running_mean" in sd:
        w, b = sd[f"{prefix}.{idx}.weight"], sd[f"{prefix}.{idx}.bias"]
        x = x @ w.T + b
        g, be = sd[f"{prefix}.{idx + 1}.weight"], sd[f"{prefix}.{idx + 1}.bias"]
        mu, var = sd[f"{prefix}.{idx + 1}.running_mean"], sd[f"{prefix}.{idx + 1}.running_var"]
        x = (x - mu) / np.sqrt(var + np.float32(BN_EPS)) * g + be
        x = np.maximum(x, np.float32(0))
        idx += 4
    w, b = sd[f"{prefix}.{idx}.weight"], sd[f"{prefix}.{idx}.bias"]
    return (x @ w.T + b).astype(np.float32)


def l2_normalize(x, eps=NORM_EPS):
    n = np.sqrt((x.astype(np.float32) ** 2).sum(axis=1, keepdims=True, dtype=np.float32))
    return (x / np.maximum(n, np.float32(eps))).astype(np.float32)


def user_tower(sd, user_cat, user_num):
    e = embed(sd, "user_tower.embedding_layer.embeddings.", user_cat)
    x = np.concatenate([e, np.asarray(user_num, dtype=np.float32)], axis=1)  # :113
    return l2_normalize(_mlp(sd, "user_tower.mlp", x))


def ad_tower(sd, ad_cat):
    e = embed(sd, "ad_tower.embedding_layer.embeddings.", ad_cat)
    return l2_normalize(_mlp(sd, "ad_tower.mlp", e))


def predict_scores(sd, user_cat, user_num, ad_cat):
    return (user_tower(sd, user_cat, user_num) * ad_tower(sd, ad_cat)).sum(
        axis=1, dtype=np.float32)
