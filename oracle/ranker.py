"""Oracle for TransformerRanker.forward in eval mode (test infrastructure).

Restates, in numpy float32:
* embed_features                 transformer_ranker.py:310-330  concat [user(6x32), ad(20x32), num(13)]
* feature_projection + pos[0]    transformer_ranker.py:355-361
* TransformerEncoderLayer        transformer_ranker.py:136-155  post-LN residual blocks
* MultiHeadAttention             transformer_ranker.py:40-90    with seq_len == 1 (:358):
      softmax over a single key is exactly 1.0, so MHA(x) == W_o(W_v x + b_v) + b_o;
      W_q / W_k / mask / 1/sqrt(d_k) are dead.  ``mha_full`` below keeps the literal
      8-head computation so tests can show the two agree bit-for-bit in float32.
* PositionwiseFeedForward        transformer_ranker.py:106-114
* FeatureInteractionLayer        transformer_ranker.py:188-204  xl = x0*(xl@W_i + b_i) + xl
* prediction_heads               transformer_ranker.py:277-305, :375-378  -> logits
"""
from __future__ import annotations

import numpy as np

LN_EPS = 1e-5   # nn.LayerNorm default (transformer_ranker.py:130-131)
TASKS = ("ctr", "engagement", "revenue")


_DT = [np.float32]     # working dtype of the restatement (float32 = the reference's; see forward(dtype=...))


def _dt():
    return _DT[0]


def _embed(sd, prefix, cat):
    names = [k for k in sd if k.startswith(prefix) and k.endswith(".weight")]
    cat = np.asarray(cat).astype(np.int64)
    assert cat.shape[1] == len(names)
    cols = []
    for i, k in enumerate(names):
        t = sd[k]
        idx = cat[:, i]
        if idx.size and (idx.min() < 0 or idx.max() >= t.shape[0]):
            raise IndexError("index out of range in self")
        cols.append(t[idx])
    return np.concatenate(cols, axis=1).astype(_dt())


def embed_features(sd, user_cat, ad_cat, numerical):
    return np.concatenate([_embed(sd, "user_embeddings.", user_cat),
                           _embed(sd, "ad_embeddings.", ad_cat),
                           np.asarray(numerical, dtype=np.float32).astype(_dt())], axis=1)


def _lin(sd, p, x):
    return (x @ sd[p + ".weight"].astype(_dt()).T + sd[p + ".bias"].astype(_dt())).astype(_dt())


def layer_norm(x, g, b, eps=LN_EPS):
    dt = x.dtype.type if x.dtype == np.float64 else np.float32
    mu = x.mean(axis=1, keepdims=True, dtype=dt)
    xc = x - mu
    var = (xc * xc).mean(axis=1, keepdims=True, dtype=dt)
    return (xc / np.sqrt(var + dt(eps)) * g.astype(dt) + b.astype(dt)).astype(dt)


def mha_seq1(sd, p, x):
    """Degenerate attention: W_o(W_v x + b_v) + b_o."""
    return _lin(sd, p + ".W_o", _lin(sd, p + ".W_v", x))


def mha_full(sd, p, x, num_heads=8):
    """Literal transformer_ranker.py:59-88 on [B,1,d] (for the degeneracy test)."""
    B, d = x.shape
    dk = d // num_heads
    q = _lin(sd, p + ".W_q", x).reshape(B, 1, num_heads, dk).transpose(0, 2, 1, 3)
    k = _lin(sd, p + ".W_k", x).reshape(B, 1, num_heads, dk).transpose(0, 2, 1, 3)
    v = _lin(sd, p + ".W_v", x).reshape(B, 1, num_heads, dk).transpose(0, 2, 1, 3)
    s = (q @ k.transpose(0, 1, 3, 2)) / _dt()(np.sqrt(dk))
    s = s - s.max(axis=-1, keepdims=True)
    w = np.exp(s)
    w = w / w.sum(axis=-1, keepdims=True)
    ctx = (w @ v).transpose(0, 2, 1, 3).reshape(B, d).astype(_dt())
    return _lin(sd, p + ".W_o", ctx)


def encoder_layer(sd, p, x, full_attention=False):
    a = (mha_full if full_attention else mha_seq1)(sd, p + ".self_attention", x)
    x = layer_norm(x + a, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
    h = np.maximum(_lin(sd, p + ".feed_forward.fc1", x), _dt()(0))
    f = _lin(sd, p + ".feed_forward.fc2", h)
    return layer_norm(x + f, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])


def cross(sd, x):
    x0, xl = x, x
    i = 0
    while f"feature_interaction.cross_weights.{i}" in sd:
        w = sd[f"feature_interaction.cross_weights.{i}"]
        b = sd[f"feature_interaction.cross_biases.{i}"]
        xl = (x0 * (xl @ w.astype(_dt()) + b.astype(_dt())) + xl).astype(_dt())     # note: xl @ W, no transpose
        i += 1
    return xl


def head(sd, task, x):
    p = f"prediction_heads.{task}"
    h = np.maximum(_lin(sd, p + ".0", x), _dt()(0))
    h = np.maximum(_lin(sd, p + ".3", h), _dt()(0))
    return _lin(sd, p + ".6", h)[:, 0]


def trunk(sd, user_cat, ad_cat, numerical, full_attention=False):
    """Everything up to (and including) the cross layers: [B, d_model]."""
    feats = embed_features(sd, user_cat, ad_cat, numerical)
    x = _lin(sd, "feature_projection", feats) + sd["positional_encoding"][0, 0].astype(_dt())
    l = 0
    while f"transformer_layers.{l}.norm1.weight" in sd:
        x = encoder_layer(sd, f"transformer_layers.{l}", x, full_attention)
        l += 1
    return cross(sd, x)


def chain_states(sd, x, dtype=np.float64):
    """Rows ``x`` [B, d_model] (the feature projection's output incl. pos[0]) -> the list of row states after each phase
    of the chain in the order the row-owner kernel runs them: per encoder layer [LN1(x + attn), LN2(x + ffn)], then
    each cross layer, and finally the logits dict.  Same arithmetic as ``forward`` (tests localise an error to a phase)."""
    _DT[0] = dtype
    try:
        x = np.asarray(x).astype(dtype)
        states = []
        l = 0
        while f"transformer_layers.{l}.norm1.weight" in sd:
            p = f"transformer_layers.{l}"
            a = mha_seq1(sd, p + ".self_attention", x)
            x = layer_norm(x + a, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"])
            states.append(x)
            h = np.maximum(_lin(sd, p + ".feed_forward.fc1", x), dtype(0))
            x = layer_norm(x + _lin(sd, p + ".feed_forward.fc2", h), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"])
            states.append(x)
            l += 1
        x0, xl = x, x
        i = 0
        while f"feature_interaction.cross_weights.{i}" in sd:
            w = sd[f"feature_interaction.cross_weights.{i}"].astype(dtype)
            xl = (x0 * (xl @ w + sd[f"feature_interaction.cross_biases.{i}"].astype(dtype)) + xl).astype(dtype)
            states.append(xl)
            i += 1
        states.append({t: head(sd, t, xl) for t in TASKS})
        return states
    finally:
        _DT[0] = np.float32


def forward(sd, user_cat, ad_cat, numerical, full_attention=False, dtype=np.float32):
    """-> dict of logits, keys in the reference's order (transformer_ranker.py:375-378).
    ``dtype=np.float64`` evaluates the same network in double precision: the "truth" against which the fp32-level
    error of an engine (and of the reference's own fp32 output) is sized in the accuracy tests."""
    _DT[0] = dtype
    try:
        x = trunk(sd, user_cat, ad_cat, numerical, full_attention)
        return {t: head(sd, t, x) for t in TASKS}
    finally:
        _DT[0] = np.float32
