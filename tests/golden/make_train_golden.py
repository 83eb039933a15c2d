#!/usr/bin/env python3
"""Loss / gradient / one-optimizer-step fixtures from the REFERENCE modules, run in this container
(/root/reference importable: two_tower_model.py, transformer_ranker.py; training_pipeline.py is not - it imports faiss
transitively - so the trainers' inner loop is restated here line by line from training_pipeline.py:118-146, :322-358).

    python tests/golden/make_train_golden.py

Models are built with dropout = 0 (the dropout masks of two implementations cannot be matched; BatchNorm batch
statistics, LayerNorm, the literal 8-head attention and every loss term are all still exercised in train mode),
weights from amdrec.synth, seeded batches.  Written: inputs, labels, the loss terms, the pre-clip gradient norm, a few
full gradients and the same tensors after ONE optimizer step.  Only these arrays travel."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "movie-recommender-demo_amd"))
sys.path.insert(0, "/root/reference")

from amdrec import synth  # noqa: E402
import two_tower_model as ref_tt  # noqa: E402
import transformer_ranker as ref_rk  # noqa: E402

TT_WATCH = ["user_tower.mlp.0.weight", "user_tower.mlp.1.weight", "ad_tower.mlp.8.bias",
            "user_tower.embedding_layer.embeddings.C3.weight", "ad_tower.mlp.5.running_var"]
RK_WATCH = ["feature_projection.bias", "transformer_layers.0.self_attention.W_q.weight",
            "transformer_layers.1.feed_forward.fc2.bias", "transformer_layers.2.norm2.weight",
            "feature_interaction.cross_weights.1", "prediction_heads.revenue.6.weight", "positional_encoding"]


def cut(a):
    """Fixtures stay small: the leading [32, 48] block of a big tensor (plus its full norm, stored beside it)."""
    a = np.asarray(a)
    return a[:32, :48].copy() if a.ndim == 2 and a.size > 4096 else (a[0, :4].copy() if a.ndim == 3 else a.copy())


def to_torch(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def main():
    torch.set_num_threads(1)
    user, ad, nnum = synth.demo_dims()
    B = 48
    rng = np.random.default_rng(5)
    ucat, unum = synth.user_batch(user, nnum, B, seed=51)
    acat = synth.ad_features(ad, B, seed=52)
    labels = rng.integers(0, 2, B).astype(np.float32)
    eng = rng.integers(0, 2, B).astype(np.float32)
    rev = rng.integers(0, 2, B).astype(np.float32)
    common = {"user_cat": ucat.astype(np.int16), "user_num": unum, "ad_cat": acat.astype(np.int16), "labels": labels,
              "engagement_labels": eng, "revenue_labels": rev}

    # ---- stage 1 (training_pipeline.py:90-146) ----
    sd = synth.two_tower_state(user, ad, nnum, seed=61)
    m = ref_tt.TwoTowerModel(dict(user), dict(ad), nnum, dropout=0.0)
    m.load_state_dict(to_torch(sd))
    m.train()
    opt = torch.optim.Adam(m.parameters(), lr=0.001, weight_decay=1e-5)
    loss_fn = ref_tt.TwoTowerLoss(alpha=0.5)
    ue, ae = m(torch.from_numpy(ucat), torch.from_numpy(unum), torch.from_numpy(acat))
    loss, ld = loss_fn(ue, ae, torch.from_numpy(labels))
    cl = m.compute_loss(ue, ae, torch.from_numpy(labels))
    opt.zero_grad()
    loss.backward()
    named = dict(m.named_parameters())
    out = dict(common, weights_sha256=synth.state_sha256(sd), seed=61, **{k: np.float64(v) for k, v in ld.items()},
               model_compute_loss=np.float64(cl.item()), user_emb=ue.detach().numpy(), ad_emb=ae.detach().numpy())
    for k in TT_WATCH:
        if k in named:
            out["grad/" + k] = cut(named[k].grad.numpy())
            out["gradnorm/" + k] = np.float64(named[k].grad.norm().item())
    out["grad_norm"] = np.float64(torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0))
    opt.step()
    after = m.state_dict()
    for k in TT_WATCH:
        out["after/" + k] = cut(after[k].numpy())
        out["afternorm/" + k] = np.float64(after[k].double().norm().item())
    np.savez_compressed(os.path.join(HERE, "train_two_tower.npz"), **out)

    # ---- stage 2 (training_pipeline.py:296-358) ----
    sd = synth.ranker_state(user, ad, nnum, seed=62, cross_scale=1.0 / 16)
    m = ref_rk.TransformerRanker(dict(user), dict(ad), nnum, dropout=0.0)
    m.load_state_dict(to_torch(sd))
    m.train()
    opt = torch.optim.AdamW(m.parameters(), lr=0.0001, weight_decay=1e-5)
    pred = m(torch.from_numpy(ucat), torch.from_numpy(acat), torch.from_numpy(unum))
    lab = {"ctr": torch.from_numpy(labels), "engagement": torch.from_numpy(eng), "revenue": torch.from_numpy(rev)}
    loss, ld = m.compute_loss(pred, lab, {"ctr": 1.0, "engagement": 0.5, "revenue": 0.3})
    opt.zero_grad()
    loss.backward()
    named = dict(m.named_parameters())
    out = dict(common, weights_sha256=synth.state_sha256(sd), seed=62, **{k: np.float64(v) for k, v in ld.items()},
               **{"pred/" + t: v.detach().numpy() for t, v in pred.items()})
    for k in RK_WATCH:
        g = named[k].grad
        out["grad/" + k] = cut(g.numpy())
        out["gradnorm/" + k] = np.float64(g.norm().item())
    out["grad_norm"] = np.float64(torch.nn.utils.clip_grad_norm_(m.parameters(), 1.0))
    opt.step()
    after = m.state_dict()
    for k in RK_WATCH:
        out["after/" + k] = cut(after[k].numpy())
        out["afternorm/" + k] = np.float64(after[k].double().norm().item())
    np.savez_compressed(os.path.join(HERE, "train_ranker.npz"), **out)
    print("training fixtures written to", HERE)


if __name__ == "__main__":
    main()
