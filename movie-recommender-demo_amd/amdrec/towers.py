"""Drop-in for the reference's ``TwoTowerModel`` (two_tower_model.py:187-314) whose eval-mode
forward runs on libamdrec's fused HIP kernels instead of ATen.

Constructor signature, sub-module names and parameter shapes are those of the reference, so
its checkpoints (``model_state_dict`` or bare ``state_dict``, inference.py:99-106) load with
``load_state_dict`` unchanged:
    user_tower.embedding_layer.embeddings.<col>.weight [card,16]
    user_tower.mlp.{0,4,8}.{weight,bias}, user_tower.mlp.{1,5}.{weight,bias,running_mean,
    running_var,num_batches_tracked}; ad_tower.* likewise.
Training: in ``train()`` mode the forward is the reference's own op sequence in stock
PyTorch autograd on the module's device (BatchNorm batch statistics, Dropout active) - the hand-written HIP
kernels are the eval-mode hot path only; ``compute_loss`` is two_tower_model.py:256-285 and the trainers live in
``amdrec.training``.  A weight update re-packs the HIP weights on the next eval-mode call (parameter versions).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import torch
import torch.nn as nn

from . import _lib, weights


class _EmbeddingLayer(nn.Module):
    def __init__(self, feature_dims: Dict[str, int], embedding_dim: int):
        super().__init__()
        self.embeddings = nn.ModuleDict({n: nn.Embedding(c, embedding_dim) for n, c in feature_dims.items()})
        self.embedding_dim = embedding_dim
        self.num_features = len(feature_dims)


def _mlp(in_dim: int, hidden_dims: Sequence[int], out_dim: int, dropout: float) -> nn.Sequential:
    mods: List[nn.Module] = []
    for h in hidden_dims:
        mods += [nn.Linear(in_dim, h), nn.BatchNorm1d(h), nn.ReLU(), nn.Dropout(dropout)]
        in_dim = h
    mods.append(nn.Linear(in_dim, out_dim))
    return nn.Sequential(*mods)


class _Tower(nn.Module):
    """Holds the parameters under the reference's names; forward = amdrec_tower_forward."""

    def __init__(self, feature_dims, numerical_dim, embedding_dim, hidden_dims, output_dim, dropout, prefix):
        super().__init__()
        self.embedding_layer = _EmbeddingLayer(feature_dims, embedding_dim)
        self.mlp = _mlp(len(feature_dims) * embedding_dim + numerical_dim, hidden_dims, output_dim, dropout)
        self.output_dim = output_dim
        self._names = list(feature_dims)
        self._n_num = numerical_dim
        self._prefix = prefix
        self._packed = None

    def invalidate(self):
        self._packed = None
        _lib.drop_tensor_list(self)

    def _apply(self, fn, *a, **k):                 # .to() / .cuda() / .float(): tensors may be replaced
        r = super()._apply(fn, *a, **k)
        self.invalidate()
        return r

    def _pack(self, device):
        key = (str(device), _lib.tensor_versions(self))
        if self._packed is None or self._packed[0] != key:
            sd = {f"{self._prefix}.{k}": v for k, v in self.state_dict().items()}
            params, keep = weights.pack_tower(sd, self._prefix, self._names, self._n_num, device)
            self._packed = (key, params, keep)
        return self._packed[1]

    def autograd_forward(self, cat: torch.Tensor, num: torch.Tensor = None) -> torch.Tensor:
        """The reference's op sequence on ATen with autograd (two_tower_model.py:33-49, :110-119 / :176-182): used in
        train mode, where BatchNorm takes batch statistics and Dropout is active."""
        embs = [emb(cat[:, i].long()) for i, emb in enumerate(self.embedding_layer.embeddings.values())]   # :43-44
        x = torch.cat(embs, dim=1)                                                                         # :47
        if self._n_num:
            x = torch.cat([x, num], dim=1)                                                                 # :113
        return torch.nn.functional.normalize(self.mlp(x), p=2, dim=1)                                      # :116-119

    def encode(self, cat: torch.Tensor, num: torch.Tensor = None, check_indices: bool = True,
               renormalize: bool = False) -> torch.Tensor:
        """``renormalize``: also apply ``faiss.normalize_L2`` to the (already normalised) rows in the same launch, for a
        caller that feeds them to ``FAISSIndex.search_device(..., normalize=False)`` - the serving path's second
        normalisation (faiss_retrieval.py:147) without its own kernel launch; bit-identical to the two calls."""
        if self.training:
            if renormalize:
                raise ValueError("renormalize is an inference-path option")
            return self.autograd_forward(cat, num)
        cat = _lib.require_gpu(cat, "categorical_features")
        dev = cat.device
        if cat.dim() != 2 or cat.shape[1] != len(self._names):
            raise ValueError(f"categorical_features must be [B, {len(self._names)}]")
        cat = cat.long().contiguous()                                   # two_tower_model.py:44 (.long())
        rows = cat.shape[0]
        if self._n_num:
            num = _lib.require_gpu(num, "numerical_features").to(torch.float32).contiguous()
            if num.shape != (rows, self._n_num):
                raise ValueError(f"numerical_features must be [{rows}, {self._n_num}]")
        else:
            num = None
        params = self._pack(dev)
        params.renormalize = 1 if renormalize else 0
        lib = _lib.load()
        out = torch.empty((rows, self.output_dim), dtype=torch.float32, device=dev)
        if rows == 0:
            return out
        flag = torch.zeros(1, dtype=torch.int32, device=dev) if check_indices else None
        nbytes = _lib.C.c_size_t(0)
        _lib.check(lib.amdrec_tower_workspace(_lib.C.byref(params), rows, _lib.C.byref(nbytes)))
        ws = _lib.WORKSPACE.get(nbytes.value, dev)
        _lib.check(lib.amdrec_tower_forward(_lib.C.byref(params), _lib.ptr(cat), _lib.ptr(num), rows,
                                            _lib.ptr(out), out.stride(0), _lib.ptr(flag), _lib.ptr(ws),
                                            ws.numel(), _lib.stream_ptr(dev)))
        if check_indices and int(flag.item()):
            raise IndexError("index out of range in self")             # torch.nn.Embedding's message
        return out


class UserTower(_Tower):
    def __init__(self, user_feature_dims, numerical_dim, embedding_dim=16, hidden_dims=(512, 256),
                 output_dim=256, dropout=0.3):
        super().__init__(user_feature_dims, numerical_dim, embedding_dim, hidden_dims, output_dim, dropout,
                         "user_tower")

    def forward(self, categorical_features, numerical_features):
        return self.encode(categorical_features, numerical_features)


class AdTower(_Tower):
    def __init__(self, ad_feature_dims, embedding_dim=16, hidden_dims=(512, 256), output_dim=256, dropout=0.3):
        super().__init__(ad_feature_dims, 0, embedding_dim, hidden_dims, output_dim, dropout, "ad_tower")

    def forward(self, categorical_features):
        return self.encode(categorical_features)


class TwoTowerModel(nn.Module):
    """two_tower_model.py:193-201 signature; forward surfaces :235-254, :287-314."""

    def __init__(self, user_feature_dims: Dict[str, int], ad_feature_dims: Dict[str, int], numerical_dim: int,
                 embedding_dim: int = 16, hidden_dims: Sequence[int] = (512, 256), output_dim: int = 256,
                 dropout: float = 0.3, temperature: float = 0.07):
        super().__init__()
        self.user_tower = UserTower(user_feature_dims, numerical_dim, embedding_dim, hidden_dims, output_dim,
                                    dropout)
        self.ad_tower = AdTower(ad_feature_dims, embedding_dim, hidden_dims, output_dim, dropout)
        self.temperature = temperature
        self.output_dim = output_dim

    def forward(self, user_categorical, user_numerical, ad_categorical) -> Tuple[torch.Tensor, torch.Tensor]:
        return self.user_tower(user_categorical, user_numerical), self.ad_tower(ad_categorical)

    def get_user_embeddings(self, user_categorical, user_numerical):
        return self.user_tower(user_categorical, user_numerical)

    def get_ad_embeddings(self, ad_categorical):
        return self.ad_tower(ad_categorical)

    def predict_scores(self, user_categorical, user_numerical, ad_categorical):
        u, a = self.forward(user_categorical, user_numerical, ad_categorical)
        return (u * a).sum(dim=1)

    def compute_loss(self, user_embeddings: torch.Tensor, ad_embeddings: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        """two_tower_model.py:256-285: in-batch softmax cross-entropy at ``self.temperature`` (diagonal = positives; the
        reference builds a positive mask from ``labels`` and never uses it)."""
        sim = torch.matmul(user_embeddings, ad_embeddings.T) / self.temperature
        return torch.nn.functional.cross_entropy(sim.view(-1, sim.size(1)), torch.arange(sim.size(0), device=sim.device))

    def load_state_dict(self, state_dict, *a, **k):
        r = super().load_state_dict(state_dict, *a, **k)
        self.user_tower.invalidate()
        self.ad_tower.invalidate()
        return r
