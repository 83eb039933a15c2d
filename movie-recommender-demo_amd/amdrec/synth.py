"""Seeded synthetic weights and inputs (numpy only).

The reference ships no checkpoints and no golden vectors, so every
fixture, parity test and benchmark uses weights drawn here from
``numpy.random.default_rng(seed)``.  Key names and shapes are exactly those of the
reference modules' ``state_dict()``:

* two-tower: two_tower_model.py:25-28 (embeddings), :83-95 (mlp Sequential indices
  0/4/8 Linear, 1/5 BatchNorm1d)
* ranker: transformer_ranker.py:242-305 (embeddings, feature_projection,
  positional_encoding, transformer_layers, feature_interaction, prediction_heads)

Distributions are chosen so that eval-mode BatchNorm / LayerNorm are non-trivial
(random running stats, gamma, beta) - a restatement that forgot one of them cannot
pass parity.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict
from typing import Dict, List, Sequence

import numpy as np

USER_COLS = [f"C{i}" for i in range(1, 7)]     # inference.py:46
AD_COLS = [f"C{i}" for i in range(7, 27)]      # inference.py:47
NUM_COLS = [f"I{i}" for i in range(1, 14)]     # data_preprocessing.py:251

# cardinalities of create_synthetic_criteo_data (data_preprocessing.py:261)
CRITEO_SYNTH_CARDS = [1000, 500, 100, 50] * 6 + [20, 10]


def demo_dims():
    """Feature dims of the reference's __main__ smoke blocks
    (two_tower_model.py:373-375, transformer_ranker.py:480-482), with C* names."""
    user = OrderedDict((c, 100) for c in USER_COLS)
    ad = OrderedDict((c, 200) for c in AD_COLS)
    return user, ad, 13


def criteo_dims():
    """Upper-bound feature dims of the synthetic Criteo generator."""
    user = OrderedDict((c, CRITEO_SYNTH_CARDS[i]) for i, c in enumerate(USER_COLS))
    ad = OrderedDict((c, CRITEO_SYNTH_CARDS[6 + i]) for i, c in enumerate(AD_COLS))
    return user, ad, 13


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _linear(rng, out_f, in_f, prefix, sd):
    bound = 1.0 / np.sqrt(in_f)
    sd[prefix + ".weight"] = _f32(rng.uniform(-bound, bound, (out_f, in_f)))
    sd[prefix + ".bias"] = _f32(rng.uniform(-bound, bound, (out_f,)))


def _batchnorm(rng, n, prefix, sd):
    sd[prefix + ".weight"] = _f32(rng.uniform(0.5, 1.5, (n,)))
    sd[prefix + ".bias"] = _f32(rng.normal(0, 0.1, (n,)))
    sd[prefix + ".running_mean"] = _f32(rng.normal(0, 0.2, (n,)))
    sd[prefix + ".running_var"] = _f32(rng.uniform(0.5, 1.5, (n,)))
    sd[prefix + ".num_batches_tracked"] = np.array(17, dtype=np.int64)


def _layernorm(rng, n, prefix, sd):
    sd[prefix + ".weight"] = _f32(rng.uniform(0.5, 1.5, (n,)))
    sd[prefix + ".bias"] = _f32(rng.normal(0, 0.1, (n,)))


def _tower(rng, prefix, dims: Dict[str, int], n_num, emb_dim, hidden, out_dim, sd):
    for name, card in dims.items():
        sd[f"{prefix}.embedding_layer.embeddings.{name}.weight"] = _f32(
            rng.normal(0, 1, (card, emb_dim)))
    prev = len(dims) * emb_dim + n_num
    idx = 0
    for h in hidden:
        _linear(rng, h, prev, f"{prefix}.mlp.{idx}", sd)
        _batchnorm(rng, h, f"{prefix}.mlp.{idx + 1}", sd)
        idx += 4            # Linear, BatchNorm1d, ReLU, Dropout
        prev = h
    _linear(rng, out_dim, prev, f"{prefix}.mlp.{idx}", sd)


def two_tower_state(user_dims, ad_dims, numerical_dim, seed=0, embedding_dim=16,
                    hidden_dims: Sequence[int] = (512, 256), output_dim=256):
    """state_dict (numpy) for reference TwoTowerModel(two_tower_model.py:193-233)."""
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    _tower(rng, "user_tower", user_dims, numerical_dim, embedding_dim, hidden_dims,
           output_dim, sd)
    _tower(rng, "ad_tower", ad_dims, 0, embedding_dim, hidden_dims, output_dim, sd)
    return sd


def ranker_state(user_dims, ad_dims, numerical_dim, seed=0, embedding_dim=32,
                 d_model=256, num_heads=8, num_layers=3, d_ff=1024, max_seq_len=50,
                 cross_scale=1.0, head_hidden=(256, 64)):
    """state_dict (numpy) for reference TransformerRanker (transformer_ranker.py:213-308).

    cross_scale=1.0 reproduces the reference's unscaled ``randn`` cross weights
    (transformer_ranker.py:177-180: logits reach 1e3); 1/16 gives a trained-looking net.
    """
    rng = np.random.default_rng(seed)
    sd: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, card in user_dims.items():
        sd[f"user_embeddings.{name}.weight"] = _f32(rng.normal(0, 1, (card, embedding_dim)))
    for name, card in ad_dims.items():
        sd[f"ad_embeddings.{name}.weight"] = _f32(rng.normal(0, 1, (card, embedding_dim)))
    total = (len(user_dims) + len(ad_dims)) * embedding_dim + numerical_dim
    _linear(rng, d_model, total, "feature_projection", sd)
    sd["positional_encoding"] = _f32(rng.normal(0, 1, (1, max_seq_len, d_model)))
    for l in range(num_layers):
        p = f"transformer_layers.{l}"
        for w in ("W_q", "W_k", "W_v", "W_o"):
            _linear(rng, d_model, d_model, f"{p}.self_attention.{w}", sd)
        _linear(rng, d_ff, d_model, f"{p}.feed_forward.fc1", sd)
        _linear(rng, d_model, d_ff, f"{p}.feed_forward.fc2", sd)
        _layernorm(rng, d_model, f"{p}.norm1", sd)
        _layernorm(rng, d_model, f"{p}.norm2", sd)
    for i in range(3):
        sd[f"feature_interaction.cross_weights.{i}"] = _f32(
            rng.normal(0, 1, (d_model, d_model)) * cross_scale)
    for i in range(3):
        sd[f"feature_interaction.cross_biases.{i}"] = _f32(
            rng.normal(0, 1, (d_model,)) * cross_scale)
    for task in ("ctr", "engagement", "revenue"):
        p = f"prediction_heads.{task}"
        _linear(rng, head_hidden[0], d_model, f"{p}.0", sd)
        _linear(rng, head_hidden[1], head_hidden[0], f"{p}.3", sd)
        _linear(rng, 1, head_hidden[1], f"{p}.6", sd)
    return sd


def state_sha256(sd) -> str:
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(v).tobytes())
    return h.hexdigest()


def user_batch(user_dims, numerical_dim, batch, seed=0):
    """Synthetic users (cf. two_tower_model.py:392-393): cat ~ U{0..card-1}, num ~ N(0,1)."""
    rng = np.random.default_rng(seed)
    cards = np.array(list(user_dims.values()), dtype=np.int64)
    cat = (rng.random((batch, len(cards))) * cards).astype(np.int64)
    num = _f32(rng.standard_normal((batch, numerical_dim)))
    return cat, num


def ad_features(ad_dims, n_ads, seed=0, dtype=np.int64):
    """Synthetic per-ad categorical table ad_cat[N,20] (the lookup the reference stubs
    out with torch.randint at inference.py:246-248)."""
    rng = np.random.default_rng(seed)
    cards = np.array(list(ad_dims.values()), dtype=np.int64)
    return (rng.random((n_ads, len(cards))) * cards).astype(dtype)


def unit_corpus(n, dim=256, seed=1234, chunk=1 << 18):
    """The reference benchmark's corpus distribution (faiss_retrieval.py:390):
    randn(N,dim) float32, rows L2-normalised as FAISSIndex.add does (:114-115)."""
    rng = np.random.default_rng(seed)
    out = np.empty((n, dim), dtype=np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        x = rng.standard_normal((e - s, dim), dtype=np.float32)
        x /= np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-30)
        out[s:e] = x
    return out
