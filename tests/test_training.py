"""Training step pinned to the reference: losses, gradients and one optimizer step of the
drop-in modules' train-mode (stock PyTorch autograd) path against fixtures captured from the reference modules
themselves (tests/golden/make_train_golden.py: two_tower_model.py / transformer_ranker.py imported in the build
container, dropout 0).  CPU tests: the autograd path is plain torch; the GPU variant below runs the same step on the
device and then serves the UPDATED weights through the HIP eval path."""
import numpy as np
import pytest
import torch

from amdrec import synth, training
from amdrec.ranker import TransformerRanker
from amdrec.towers import TwoTowerModel
from tests.conftest import load_golden


def _t(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def _cut(a):
    a = np.asarray(a)
    return a[:32, :48] if a.ndim == 2 and a.size > 4096 else (a[0, :4] if a.ndim == 3 else a)


def _batch(g, device="cpu"):
    return {"user_categorical": torch.from_numpy(g["user_cat"].astype(np.int64)).to(device),
            "ad_categorical": torch.from_numpy(g["ad_cat"].astype(np.int64)).to(device),
            "numerical": torch.from_numpy(g["user_num"]).to(device), "labels": torch.from_numpy(g["labels"]).to(device),
            "engagement_labels": torch.from_numpy(g["engagement_labels"]).to(device),
            "revenue_labels": torch.from_numpy(g["revenue_labels"]).to(device)}


def _two_tower(g, device="cpu"):
    user, ad, nnum = synth.demo_dims()
    sd = synth.two_tower_state(user, ad, nnum, seed=int(g["seed"]))
    assert synth.state_sha256(sd) == str(g["weights_sha256"])
    m = TwoTowerModel(dict(user), dict(ad), nnum, dropout=0.0)
    m.load_state_dict(_t(sd))
    return m.to(device)


def _ranker(g, device="cpu"):
    user, ad, nnum = synth.demo_dims()
    sd = synth.ranker_state(user, ad, nnum, seed=int(g["seed"]), cross_scale=1.0 / 16)
    assert synth.state_sha256(sd) == str(g["weights_sha256"])
    m = TransformerRanker(dict(user), dict(ad), nnum, dropout=0.0)
    m.load_state_dict(_t(sd))
    return m.to(device)


def _check_step(g, model, loss_dict, rtol):
    named = dict(model.named_parameters())
    after = model.state_dict()
    for key in g.files:
        if key.startswith("after/"):
            k = key[len("after/"):]
            ref = g[key]
            got = _cut(after[k].detach().cpu().numpy())
            assert np.abs(got - ref).max() <= rtol * max(1.0, np.abs(ref).max()), key
            n = float(after[k].double().norm().item())
            assert abs(n - float(g["afternorm/" + k])) <= rtol * max(1.0, n), key
    assert abs(loss_dict["grad_norm"] - float(g["grad_norm"])) <= 10 * rtol * max(1.0, float(g["grad_norm"]))
    del named


def test_two_tower_train_step_matches_reference_fixture():
    torch.set_num_threads(1)
    g = load_golden("train_two_tower.npz")
    m = _two_tower(g)
    b = _batch(g)
    m.train()
    ue, ae = m(b["user_categorical"], b["numerical"], b["ad_categorical"])          # BatchNorm batch statistics
    assert np.abs(ue.detach().numpy() - g["user_emb"]).max() <= 1e-6
    assert np.abs(ae.detach().numpy() - g["ad_emb"]).max() <= 1e-6
    loss, ld = training.TwoTowerLoss(alpha=0.5)(ue, ae, b["labels"])
    for k in ("total_loss", "pointwise_loss", "contrastive_loss"):
        assert abs(ld[k] - float(g[k])) <= 1e-6 * max(1.0, abs(float(g[k]))), k
    assert abs(m.compute_loss(ue, ae, b["labels"]).item() - float(g["model_compute_loss"])) <= 1e-6 * 10     # :256-285
    loss.backward()
    named = dict(m.named_parameters())
    for key in g.files:
        if key.startswith("grad/"):
            k = key[len("grad/"):]
            got = _cut(named[k].grad.numpy())
            assert np.abs(got - g[key]).max() <= 1e-6 * max(1.0, np.abs(g[key]).max()), key
            assert abs(named[k].grad.norm().item() - float(g["gradnorm/" + k])) <= 1e-5 * max(1.0, float(g["gradnorm/" + k]))
    # the trainer's whole inner loop on a fresh copy: zero_grad -> backward -> clip(1.0) -> Adam step
    m2 = _two_tower(g)
    nbt0 = int(m2.user_tower.mlp[1].num_batches_tracked)
    tr = training.TwoTowerTrainer(m2, device="cpu")
    ld2 = tr.train_step(b)
    assert abs(ld2["total_loss"] - float(g["total_loss"])) <= 1e-6 * 10
    _check_step(g, m2, ld2, 2e-6)
    assert int(m2.user_tower.mlp[1].num_batches_tracked) == nbt0 + 1                   # BatchNorm ran in train mode


def test_ranker_train_step_matches_reference_fixture():
    torch.set_num_threads(1)
    g = load_golden("train_ranker.npz")
    m = _ranker(g)
    b = _batch(g)
    m.train()
    pred = m(b["user_categorical"], b["ad_categorical"], b["numerical"])
    for t in ("ctr", "engagement", "revenue"):
        assert np.abs(pred[t].detach().numpy() - g["pred/" + t]).max() <= 2e-6
    labels = {"ctr": b["labels"], "engagement": b["engagement_labels"], "revenue": b["revenue_labels"]}
    loss, ld = m.compute_loss(pred, labels)                                            # default task weights :398-399
    for k in ("total_loss", "ctr_loss", "engagement_loss", "revenue_loss"):
        assert abs(ld[k] - float(g[k])) <= 2e-6 * max(1.0, abs(float(g[k]))), k
    loss.backward()
    named = dict(m.named_parameters())
    for key in g.files:
        if key.startswith("grad/"):
            k = key[len("grad/"):]
            ref = g[key]
            assert np.abs(_cut(named[k].grad.numpy()) - ref).max() <= 2e-6 * max(1.0, np.abs(ref).max()) + 1e-9, key
    # W_q receives a gradient of exactly zero: softmax over a single key (the dead parameters of SURVEY fact 1)
    assert float(named["transformer_layers.0.self_attention.W_q.weight"].grad.abs().max()) == 0.0
    m2 = _ranker(g)
    tr = training.TransformerTrainer(m2, device="cpu")
    ld2 = tr.train_step(b)
    assert abs(ld2["total_loss"] - float(g["total_loss"])) <= 2e-6 * 10
    _check_step(g, m2, ld2, 5e-6)


def test_train_mode_dropout_is_active_and_eval_mode_is_the_hip_path():
    user, ad, nnum = synth.demo_dims()
    m = TransformerRanker(dict(user), dict(ad), nnum)                                   # dropout 0.1 as the reference
    uc, un = synth.user_batch(user, nnum, 16, seed=1)
    ac = synth.ad_features(ad, 16, seed=2)
    m.train()
    torch.manual_seed(0)
    a = m(torch.from_numpy(uc), torch.from_numpy(ac), torch.from_numpy(un))["ctr"]
    b = m(torch.from_numpy(uc), torch.from_numpy(ac), torch.from_numpy(un))["ctr"]
    assert not torch.equal(a, b)                                                        # two dropout draws
    from amdrec import _lib
    m.eval()
    with pytest.raises(_lib.AmdrecError):                                               # eval on CPU tensors: no fallback
        m(torch.from_numpy(uc), torch.from_numpy(ac), torch.from_numpy(un))


@pytest.mark.gpu
def test_train_step_on_device_then_serve_updated_weights_through_hip():
    import oracle
    g = load_golden("train_ranker.npz")
    m = _ranker(g, "cuda")
    b = _batch(g, "cuda")
    m.eval()
    before = m(b["user_categorical"], b["ad_categorical"], b["numerical"])["ctr"].clone()
    tr = training.TransformerTrainer(m, device="cuda")
    ld = tr.train_step(b)
    assert abs(ld["total_loss"] - float(g["total_loss"])) <= 1e-4 * max(1.0, float(g["total_loss"]))     # device fp32 vs CPU fp32
    m.eval()
    after = m(b["user_categorical"], b["ad_categorical"], b["numerical"])
    assert not torch.equal(after["ctr"], before)                                       # the HIP path re-packed the new weights
    sd = {k: v.detach().cpu().numpy() for k, v in m.state_dict().items()}
    ref = oracle.ranker.forward(sd, g["user_cat"].astype(np.int64), g["ad_cat"].astype(np.int64), g["user_num"])
    for t in ref:
        assert np.abs(after[t].cpu().numpy() - ref[t]).max() <= 1e-5 * max(1.0, np.abs(ref[t]).max())
    gt = load_golden("train_two_tower.npz")
    mt = _two_tower(gt, "cuda")
    ld = training.TwoTowerTrainer(mt, device="cuda").train_step(_batch(gt, "cuda"))
    assert abs(ld["total_loss"] - float(gt["total_loss"])) <= 1e-4 * max(1.0, float(gt["total_loss"]))
    mt.eval()
    sd = {k: v.detach().cpu().numpy() for k, v in mt.state_dict().items()}
    ue = mt.get_user_embeddings(_batch(gt, "cuda")["user_categorical"], _batch(gt, "cuda")["numerical"])
    assert np.abs(ue.cpu().numpy() - oracle.towers.user_tower(sd, gt["user_cat"].astype(np.int64), gt["user_num"])).max() <= 1e-5
