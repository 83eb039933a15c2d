"""Drop-in for the reference's ``TransformerRanker`` (transformer_ranker.py:207-415): same
constructor, same parameter names/shapes (reference checkpoints load unchanged, including the
dead W_q / W_k and the unused positional rows 1..49), eval-mode ``forward`` on libamdrec.

``forward(user_categorical, ad_categorical, numerical, mask=None)`` keeps the reference's
argument order (ad_categorical is SECOND, unlike TwoTowerModel) and returns the logits dict
``{'ctr','engagement','revenue'}``.  ``mask`` is accepted and ignored: with the sequence
length fixed at 1 (:358) softmax over a single key is 1.0 whatever the mask says.
``score_candidates`` is the pipeline entry: one user row broadcast over its stage-1
candidates, ad features gathered from a resident table by candidate id.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib, weights


class _Attention(nn.Module):
    def __init__(self, d_model, num_heads, dropout):
        super().__init__()
        assert d_model % num_heads == 0
        self.d_model, self.num_heads, self.d_k = d_model, num_heads, d_model // num_heads
        for n in ("W_q", "W_k", "W_v", "W_o"):
            setattr(self, n, nn.Linear(d_model, d_model))
        self.dropout = nn.Dropout(dropout)


class _FFN(nn.Module):
    def __init__(self, d_model, d_ff, dropout):
        super().__init__()
        self.fc1, self.fc2 = nn.Linear(d_model, d_ff), nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)


class _EncoderLayer(nn.Module):
    def __init__(self, d_model, num_heads, d_ff, dropout):
        super().__init__()
        self.self_attention = _Attention(d_model, num_heads, dropout)
        self.feed_forward = _FFN(d_model, d_ff, dropout)
        self.norm1, self.norm2 = nn.LayerNorm(d_model), nn.LayerNorm(d_model)
        self.dropout1, self.dropout2 = nn.Dropout(dropout), nn.Dropout(dropout)


class _Cross(nn.Module):
    def __init__(self, dim, num_crosses, dropout):
        super().__init__()
        self.num_crosses = num_crosses
        self.cross_weights = nn.ParameterList([nn.Parameter(torch.randn(dim, dim)) for _ in range(num_crosses)])
        self.cross_biases = nn.ParameterList([nn.Parameter(torch.randn(dim)) for _ in range(num_crosses)])
        self.dropout = nn.Dropout(dropout)


def _head(d_model, dropout):
    return nn.Sequential(nn.Linear(d_model, 256), nn.ReLU(), nn.Dropout(dropout), nn.Linear(256, 64), nn.ReLU(),
                         nn.Dropout(dropout), nn.Linear(64, 1))


class TransformerRanker(nn.Module):
    def __init__(self, user_feature_dims: Dict[str, int], ad_feature_dims: Dict[str, int], numerical_dim: int,
                 embedding_dim: int = 32, d_model: int = 256, num_heads: int = 8, num_layers: int = 3,
                 d_ff: int = 1024, max_seq_len: int = 50, dropout: float = 0.1, num_objectives: int = 3):
        super().__init__()
        self.user_embeddings = nn.ModuleDict({n: nn.Embedding(c, embedding_dim) for n, c in user_feature_dims.items()})
        self.ad_embeddings = nn.ModuleDict({n: nn.Embedding(c, embedding_dim) for n, c in ad_feature_dims.items()})
        total = (len(user_feature_dims) + len(ad_feature_dims)) * embedding_dim + numerical_dim
        self.feature_projection = nn.Linear(total, d_model)
        self.positional_encoding = nn.Parameter(torch.randn(1, max_seq_len, d_model))
        self.transformer_layers = nn.ModuleList([_EncoderLayer(d_model, num_heads, d_ff, dropout)
                                                 for _ in range(num_layers)])
        self.feature_interaction = _Cross(d_model, 3, dropout)
        self.prediction_heads = nn.ModuleDict({t: _head(d_model, dropout) for t in ("ctr", "engagement", "revenue")})
        self.d_model = d_model
        self.dropout = nn.Dropout(dropout)
        self._user_names, self._ad_names, self._n_num = list(user_feature_dims), list(ad_feature_dims), numerical_dim
        self._packed = None
        self._ad_cache = None
        # W_ov = W_o W_v pre-multiplied on the host (exact algebra at seq_len 1; set False to run the two
        # GEMMs in the reference's order)
        self.fuse_attention = True
        # engine of the big passes (> 8192 rows), all fp32 in / fp32 out with fp32-level error:
        #  "f16x3"  (default) the row-owner kernel (csrc/rowowner.hpp): operands split into two fp16 planes, three
        #           fp16-MFMA products per MAC, everything after the projection in ONE kernel, activations in registers
        #  "bf16x6" the round-1 tile GEMMs on three bf16 planes, six products per MAC (also the fallback of "f16x3" for
        #           architectures the row-owner kernel is not written for)
        #  "fp32"   fp32 MFMA everywhere
        self.gemm_engine = "f16x3"
        # smallest pass that takes the row-owner kernel.  1 = every batch: one launch (~0.35 ms per 128-row round, bound by
        # one CU's weight-stream rate) matches the ~25 small launches of the fp32 path at a single request (0.60 ms end to
        # end either way) and beats it from 2 users up (B = 16: 0.65 vs 0.79 ms; tools/latency_by_engine.py)
        self.x3_min_rows = 1
        self.x3_variant = 16                          # 16: rowowner16.hpp (two waves per SIMD, default: 15 % faster); 32: rowowner.hpp
        # passes of at most this many rows take the column-split kernel (csrc/rowowner16c.hpp: 16 rows per workgroup, the
        # waves split the output features; bit-identical results).  0 = the library's default (4096 rows = 8 requests of
        # 500 candidates), -1 = never
        self.x3_cs_max_rows = 0

    ENGINES = ("f16x3", "bf16x6", "fp32")
    SMALL_ROWS = 8192       # passes of at most this many rows run the fp32-MFMA small shapes (csrc/layers.hip)

    def x3_fallback_reason(self):
        """Why ``gemm_engine = "f16x3"`` would NOT run the row-owner kernel for these weights (None: it runs): the
        engine is written for the reference's default architecture; anything else takes the generic tile GEMMs
        (bf16x6 above SMALL_ROWS rows, fp32 MFMA below).  The first such pack also logs a warning - the fallback is
        correct (golden-tested on the tutorial's architecture) but several times slower, and must not be silent."""
        return weights.x3_ineligible_reason(self.state_dict(), self.fuse_attention)

    def gemm_engine_for(self, rows: int) -> str:
        """The engine a pass of ``rows`` rows actually runs on."""
        eng = self.gemm_engine
        if eng == "f16x3":
            if weights.x3_eligible(self.state_dict(), self.fuse_attention):
                return "f16x3" if rows >= self.x3_min_rows else "fp32"
            eng = "bf16x6"
        return eng if rows > self.SMALL_ROWS else "fp32"

    # -- packing ----------------------------------------------------------------------
    def invalidate(self):
        self._packed = None
        self._ad_cache = None
        _lib.drop_tensor_list(self)

    def _apply(self, fn, *a, **k):                 # .to() / .cuda() / .float(): tensors may be replaced
        r = super()._apply(fn, *a, **k)
        self.invalidate()
        return r

    def cache_ad_projection(self, ad_table: Optional[torch.Tensor]):
        """Candidate-side cache for ``score_candidates``: the ad half of the feature projection,
        W_proj[:, ad columns] . emb(ad_table[a]), for every row of the resident ad-feature table
        ([N, d_model] fp32, N * 1 KB of HBM at d_model 256).  Like the ad-tower embeddings in the index it depends
        on the ad and the weights only; with it stage 2 replaces the widest-K GEMM of the ranker by a row gather and
        returns bit-identical logits.  ``None`` drops the cache.  Re-packing the weights (load_state_dict, a
        parameter update) drops it too; ``score_candidates`` uses it only for the very table it was built from."""
        self._ad_cache = None
        if ad_table is None:
            return None
        table = _lib.require_gpu(ad_table, "ad_table", torch.int64)
        if not table.is_contiguous() or table.dim() != 2 or table.shape[1] != len(self._ad_names):
            raise ValueError("ad_table must be a contiguous [N, n_ad_feat] int64 tensor")
        dev = table.device
        params, _ = self._pack(dev)
        if not params.w_proj_ad:
            return None                                     # no ad features / no split projection
        lib = _lib.load()
        out = torch.empty((table.shape[0], self.d_model), dtype=torch.float32, device=dev)
        ws = _lib.WORKSPACE.get(4 * self.d_model + 256, dev)
        _lib.check(lib.amdrec_ranker_project_ads(_lib.C.byref(params), _lib.ptr(table), table.shape[0], _lib.ptr(out),
                                                 out.stride(0), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
        self._ad_cache = (self._packed[0], table.data_ptr(), tuple(table.shape), table._version, out)
        return out

    def ensure_ad_cache(self, ad_table):
        """Build the cache for ``ad_table`` unless a valid one exists (weights or table changed -> rebuilt)."""
        self._pack(ad_table.device)
        if self._cache_for(ad_table) is None:
            self.cache_ad_projection(ad_table)

    def _cache_for(self, table):
        c = getattr(self, "_ad_cache", None)
        if c is None or self._packed is None:
            return None
        if c[0] != self._packed[0] or c[1] != table.data_ptr() or c[2] != tuple(table.shape) or c[3] != table._version:
            return None
        return c[4]

    def _pack(self, device):
        if self.gemm_engine not in self.ENGINES:
            raise ValueError(f"gemm_engine must be one of {self.ENGINES}")
        key = (str(device), self.fuse_attention, self.gemm_engine, int(self.x3_min_rows), int(self.x3_variant),
               int(self.x3_cs_max_rows),
               _lib.tensor_versions(self))
        if self._packed is None or self._packed[0] != key:
            sd = self.state_dict()
            why = weights.x3_ineligible_reason(sd, self.fuse_attention) if self.gemm_engine == "f16x3" else None
            x3 = self.gemm_engine == "f16x3" and why is None
            if why is not None:
                import warnings
                warnings.warn(f"amdrec TransformerRanker: gemm_engine 'f16x3' is not available for these weights ({why}); "
                              f"running the generic tile GEMMs (bf16x6 above {self.SMALL_ROWS} rows, fp32 MFMA below)",
                              RuntimeWarning, stacklevel=3)
            params, keep, tasks = weights.pack_ranker(sd, self._user_names, self._ad_names,
                                                      self._n_num, device, fuse_attention=self.fuse_attention,
                                                      x6=self.gemm_engine == "bf16x6" or
                                                      (self.gemm_engine == "f16x3" and not x3),
                                                      x3=x3, x3_min_rows=self.x3_min_rows, x3_variant=self.x3_variant,
                                                      x3_cs_max_rows=self.x3_cs_max_rows)
            self._packed = (key, params, keep, tasks)
        return self._packed[1], self._packed[3]

    def load_state_dict(self, state_dict, *a, **k):
        r = super().load_state_dict(state_dict, *a, **k)
        self.invalidate()
        return r

    # -- forward ----------------------------------------------------------------------
    def autograd_forward(self, user_categorical, ad_categorical, numerical, mask=None):
        """The reference's op sequence on ATen with autograd (transformer_ranker.py:310-380 with :59-88, :114, :148-153,
        :199-203): used in train mode.  The LITERAL 8-head attention is kept here - with the attention-weight dropout
        (:73) active the seq-len-1 softmax weight is 0 or 1/(1-p) per (row, head), not the constant 1 of eval mode."""
        import math
        F = torch.nn.functional
        ue = torch.cat([e(user_categorical[:, i].long()) for i, e in enumerate(self.user_embeddings.values())], dim=1)
        ae = torch.cat([e(ad_categorical[:, i].long()) for i, e in enumerate(self.ad_embeddings.values())], dim=1)
        x = self.feature_projection(torch.cat([ue, ae, numerical], dim=1)).unsqueeze(1)        # :328, :355-358
        x = self.dropout(x + self.positional_encoding[:, :1, :])                                # :361-362
        for layer in self.transformer_layers:
            at = layer.self_attention
            B = x.size(0)
            q = at.W_q(x).view(B, -1, at.num_heads, at.d_k).transpose(1, 2)
            k = at.W_k(x).view(B, -1, at.num_heads, at.d_k).transpose(1, 2)
            v = at.W_v(x).view(B, -1, at.num_heads, at.d_k).transpose(1, 2)
            scores = torch.matmul(q, k.transpose(-2, -1)) / math.sqrt(at.d_k)
            if mask is not None:
                scores = scores.masked_fill(mask == 0, -1e9)
            w = at.dropout(F.softmax(scores, dim=-1))
            ctx = torch.matmul(w, v).transpose(1, 2).contiguous().view(B, -1, at.d_model)
            x = layer.norm1(x + layer.dropout1(at.W_o(ctx)))                                    # :148-149
            ff = layer.feed_forward
            x = layer.norm2(x + layer.dropout2(ff.fc2(ff.dropout(F.relu(ff.fc1(x))))))           # :114, :152-153
        x = x.squeeze(1)
        fi = self.feature_interaction
        x0, xl = x, x
        for i in range(fi.num_crosses):
            xl = fi.dropout(x0 * (torch.matmul(xl, fi.cross_weights[i]) + fi.cross_biases[i]) + xl)   # :201-202
        return {t: head(xl).squeeze(1) for t, head in self.prediction_heads.items()}            # :375-378

    def _run(self, user_cat, numerical, user_rowdiv, ad_cat, ad_rowmap, rows, check_indices=True, raw=False,
             use_cache=False):
        if self.training:
            raise NotImplementedError("score_candidates / the HIP forward implement eval() semantics only; call .eval()")
        dev = ad_cat.device
        params, tasks = self._pack(dev)
        cache = self._cache_for(ad_cat) if use_cache else None
        params.ad_proj_cache = cache.data_ptr() if cache is not None else None
        params.ld_ad_proj_cache = cache.stride(0) if cache is not None else 0
        lib = _lib.load()
        logits = torch.empty((len(tasks), rows), dtype=torch.float32, device=dev)
        if rows == 0:
            return (tasks, logits) if raw else {t: logits[i] for i, t in enumerate(tasks)}
        flag = torch.zeros(1, dtype=torch.int32, device=dev) if check_indices else None
        nbytes = _lib.C.c_size_t(0)
        _lib.check(lib.amdrec_ranker_workspace(_lib.C.byref(params), rows, _lib.C.byref(nbytes)))
        ws = _lib.WORKSPACE.get(nbytes.value, dev)
        _lib.check(lib.amdrec_ranker_forward(
            _lib.C.byref(params), _lib.ptr(user_cat), _lib.ptr(numerical), user_rowdiv, _lib.ptr(ad_cat),
            _lib.ptr(ad_rowmap), rows, _lib.ptr(logits), logits.stride(0), _lib.ptr(flag),
            user_cat.shape[0], ad_cat.shape[0], _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
        if check_indices and int(flag.item()):
            raise IndexError("index out of range in self")
        return (tasks, logits) if raw else {t: logits[i] for i, t in enumerate(tasks)}

    def forward(self, user_categorical, ad_categorical, numerical, mask: Optional[torch.Tensor] = None):
        """transformer_ranker.py:332-380 -> {'ctr','engagement','revenue'}: logits [B].  train() mode: the same network
        in stock PyTorch autograd (``autograd_forward``), for amdrec.training."""
        if self.training:
            return self.autograd_forward(user_categorical, ad_categorical, numerical, mask)
        uc = _lib.require_gpu(user_categorical, "user_categorical").long().contiguous()
        ac = _lib.require_gpu(ad_categorical, "ad_categorical").long().contiguous()
        nm = _lib.require_gpu(numerical, "numerical").to(torch.float32).contiguous()
        B = uc.shape[0]
        if uc.shape != (B, len(self._user_names)) or ac.shape != (B, len(self._ad_names)) or \
                nm.shape != (B, self._n_num):
            raise ValueError("bad feature shapes")
        return self._run(uc, nm, 1, ac, None, B)

    def score_candidates(self, user_categorical, numerical, candidate_rows, ad_table, check_indices=False,
                         raw=False):
        """Pipeline form of inference.py:241-255: user u's features are broadcast over its
        ``k = candidate_rows.shape[1]`` candidates (no .repeat), ad features are gathered from the
        resident ``ad_table [N, n_ad_feat]`` by candidate row.  -> logits dict, each [U*k]
        (``raw=True``: (task names, one [n_tasks, U*k] tensor))."""
        uc = _lib.require_gpu(user_categorical, "user_categorical").long().contiguous()
        nm = _lib.require_gpu(numerical, "numerical").to(torch.float32).contiguous()
        cand = _lib.require_gpu(candidate_rows, "candidate_rows", torch.int64).contiguous()
        table = _lib.require_gpu(ad_table, "ad_table", torch.int64)
        if not table.is_contiguous():
            raise ValueError("ad_table must be contiguous")
        U, k = cand.shape
        if uc.shape[0] != U or nm.shape[0] != U:
            raise ValueError("one user row per candidate list expected")
        return self._run(uc, nm, k, table, cand.view(-1), U * k, check_indices, raw, use_cache=True)

    def compute_loss(self, predictions, labels, task_weights=None):
        """transformer_ranker.py:382-415: weighted sum of per-task BCE-with-logits -> (total, dict of floats)."""
        if task_weights is None:
            task_weights = {"ctr": 1.0, "engagement": 0.5, "revenue": 0.3}
        losses, total = {}, 0
        for task in predictions.keys():
            if task in labels:
                tl = torch.nn.functional.binary_cross_entropy_with_logits(predictions[task], labels[task].float())
                losses[f"{task}_loss"] = tl.item()
                total = total + task_weights.get(task, 1.0) * tl
        losses["total_loss"] = total.item()
        return total, losses
