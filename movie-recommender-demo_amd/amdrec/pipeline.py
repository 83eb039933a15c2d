"""Drop-in for the reference's serving pipeline ``AdRecommenderInference`` (inference.py:21-331).

``recommend_ads(user_data, top_k=10, stage1_k=500, return_scores=True)`` keeps the reference's
argument names / defaults, stage order, ranking key (CTR only, inference.py:263) and result
schema (:272-288).  The whole hot path stays on the device between the feature tensors and
the final ``[B, top_k]`` result: UserTower -> L2 renorm -> exact IP top-``stage1_k`` ->
ranker over the candidates (user row broadcast, ad features gathered from a resident table)
-> top-``top_k``.  The reference crosses the host/device boundary four times per request
(inference.py:225-229, :241-248, :258-260); this path crosses it once each way.

Deviations, each forced by a reference defect:
* candidate ad features come from a real ``ad_features[N, 20]`` table indexed by corpus position
  (the reference draws ``torch.randint`` placeholders, inference.py:246-248);
* unknown categories map to ``'rare'`` if the encoder has it, else class 0 (the reference asks
  the encoder for a ``'missing'`` class that does not exist, inference.py:180);
* numerical features are cast to float32 after scaling (the reference feeds float64 into a
  float32 model, inference.py:193-195);
* the preprocessor sidecar is JSON (``preprocessor.json``), never a pickle.
"""
from __future__ import annotations

import json
import time
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import torch

from . import _lib
from .index import FAISSIndex
from .ranker import TransformerRanker
from .towers import TwoTowerModel

USER_COLS = [f"C{i}" for i in range(1, 7)]      # inference.py:46
AD_COLS = [f"C{i}" for i in range(7, 27)]       # inference.py:47
TASKS = ("ctr", "engagement", "revenue")


class Preprocessor:
    """The fitted state of CriteoDataPreprocessor that inference needs (data_preprocessing.py:
    label encoders' classes, numerical column list, StandardScaler mean/scale), as plain data."""

    def __init__(self, classes: Dict[str, List[str]], numerical_cols: List[str], mean, scale):
        self.classes = {c: list(v) for c, v in classes.items()}
        self._lookup = {c: {s: i for i, s in enumerate(v)} for c, v in self.classes.items()}
        self.numerical_cols = list(numerical_cols)
        self.mean = np.asarray(mean, dtype=np.float64)
        self.scale = np.asarray(scale, dtype=np.float64)
        self.feature_dims = {c: len(v) for c, v in self.classes.items()}

    def encode(self, col, value) -> int:
        lut = self._lookup[col]
        if value in lut:
            return lut[value]
        return lut.get("rare", 0)

    def save(self, path):
        with open(path, "w") as f:
            json.dump({"classes": self.classes, "numerical_cols": self.numerical_cols,
                       "mean": self.mean.tolist(), "scale": self.scale.tolist()}, f)

    @classmethod
    def load(cls, path):
        with open(path) as f:
            d = json.load(f)
        return cls(d["classes"], d["numerical_cols"], d["mean"], d["scale"])


def _load_checkpoint(model, path, device):
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    sd = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt  # :101-106
    model.load_state_dict(sd)
    return model.to(device).eval()


class AdRecommenderInference:
    def __init__(self, model_dir: Optional[str] = None, device: str = "cuda", *,
                 two_tower_model: Optional[TwoTowerModel] = None,
                 transformer_ranker: Optional[TransformerRanker] = None,
                 faiss_index: Optional[FAISSIndex] = None, ad_features=None,
                 preprocessor: Optional[Preprocessor] = None, verbose: bool = False,
                 cache_ad_projection: bool = True):
        """Either ``model_dir`` (files below) or the components directly.
        model_dir: preprocessor.json, two_tower_{best,final}.pt, transformer_ranker_{best,final}.pt,
        faiss_index.bin (+ .metadata) in this build's format, ad_features.npy [N, 20]."""
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.AmdrecError("AdRecommenderInference needs a HIP device (no CPU fallback)")
        self.verbose = verbose
        # candidate-side cache of the ranker's ad-half projection (TransformerRanker.cache_ad_projection):
        # N x d_model fp32 next to the ad-feature table, rebuilt when the weights change
        self.cache_ad_projection = cache_ad_projection
        _lib.load()
        if model_dir is not None:
            d = Path(model_dir)
            self.preprocessor = Preprocessor.load(d / "preprocessor.json")
            self.user_feature_dims = {c: self.preprocessor.feature_dims[c] for c in USER_COLS
                                      if c in self.preprocessor.feature_dims}
            self.ad_feature_dims = {c: self.preprocessor.feature_dims[c] for c in AD_COLS
                                    if c in self.preprocessor.feature_dims}
            self.numerical_dim = len(self.preprocessor.numerical_cols)
            tt = TwoTowerModel(self.user_feature_dims, self.ad_feature_dims, self.numerical_dim)   # :84-92
            p = d / "two_tower_best.pt"
            self.two_tower_model = _load_checkpoint(tt, p if p.exists() else d / "two_tower_final.pt", self.device)
            rk = TransformerRanker(self.user_feature_dims, self.ad_feature_dims, self.numerical_dim)  # :114-124
            p = d / "transformer_ranker_best.pt"
            self.transformer_ranker = _load_checkpoint(
                rk, p if p.exists() else d / "transformer_ranker_final.pt", self.device)
            self.faiss_index = FAISSIndex(256, index_type="IVF", nlist=100, nprobe=10, device=self.device)  # :145-151
            self.faiss_index.load(str(d / "faiss_index.bin"))
            ad_features = np.load(d / "ad_features.npy", allow_pickle=False)
        else:
            if two_tower_model is None or transformer_ranker is None or faiss_index is None or ad_features is None:
                raise ValueError("pass model_dir or all of two_tower_model, transformer_ranker, faiss_index, "
                                 "ad_features")
            self.preprocessor = preprocessor
            self.two_tower_model = two_tower_model.to(self.device).eval()
            self.transformer_ranker = transformer_ranker.to(self.device).eval()
            self.faiss_index = faiss_index
        if isinstance(ad_features, torch.Tensor):
            self.ad_features = ad_features.to(device=self.device, dtype=torch.int64).contiguous()
        else:
            self.ad_features = torch.from_numpy(np.ascontiguousarray(ad_features, dtype=np.int64)).to(self.device)
        if self.ad_features.shape[0] < self.faiss_index.index.ntotal:
            raise ValueError("ad_features has fewer rows than the index")

    # -- host-side feature prep (inference.py:160-197; CPU string work, not the hot path) -----
    def preprocess_user_features(self, user_data: dict):
        if self.preprocessor is None:
            raise ValueError("no preprocessor loaded")
        pp = self.preprocessor
        cat = [pp.encode(c, user_data["categorical"].get(c, "missing")) for c in USER_COLS if c in pp.classes]
        num = np.array([np.log1p(np.abs(user_data["numerical"].get(c, 0))) for c in pp.numerical_cols],
                       dtype=np.float64)
        num = ((num - pp.mean) / pp.scale).astype(np.float32)
        return torch.tensor([cat], dtype=torch.long), torch.from_numpy(num[None, :])

    def preprocess_batch(self, user_data_list: list):
        """Batch form of preprocess_user_features with the numerical transform on the device: categorical strings
        are label-encoded on the host (dictionary lookups), raw numericals are shipped as one float32 block and
        log1p / standardised by amdrec_prep_numerical.  -> (user_categorical [B,6] int64, user_numerical [B,13]
        float32), both on the device."""
        pp = self.preprocessor
        if pp is None:
            raise ValueError("no preprocessor loaded")
        cols = [c for c in USER_COLS if c in pp.classes]
        B, nc, nn_ = len(user_data_list), len(cols), len(pp.numerical_cols)
        dev = self.ad_features.device
        if getattr(self, "_pp_dev", (None,))[0] is not pp:
            self._pp_dev = (pp, torch.from_numpy(pp.mean.astype(np.float32)).to(dev),
                            torch.from_numpy(pp.scale.astype(np.float32)).to(dev))
        # ONE pinned staging block [categorical int64 | raw numerical float32] and ONE H2D copy per call (two pageable
        # copies cost ~25 us of a 0.4 ms request); the block is reused once its previous copy has completed
        cat_bytes = B * nc * 8
        host, done = self._staging(cat_bytes + B * nn_ * 4)
        hv = host.numpy()
        if nc:
            hv[:cat_bytes].view(np.int64).reshape(B, nc)[:] = [[pp.encode(c, u["categorical"].get(c, "missing")) for c in cols]
                                                               for u in user_data_list]
        if nn_:
            hv[cat_bytes:].view(np.float32).reshape(B, nn_)[:] = [[float(u["numerical"].get(c, 0)) for c in pp.numerical_cols]
                                                                  for u in user_data_list]
        blk = host.to(dev, non_blocking=True)
        done.record(torch.cuda.current_stream(dev))
        cat = blk[:cat_bytes].view(torch.int64).view(B, nc)
        x = blk[cat_bytes:].view(torch.float32).view(B, nn_)
        out = torch.empty_like(x)
        lib = _lib.load()
        _lib.check(lib.amdrec_prep_numerical(_lib.ptr(x), _lib.ptr(self._pp_dev[1]), _lib.ptr(self._pp_dev[2]),
                                             _lib.ptr(out), x.shape[0], x.shape[1], _lib.stream_ptr(dev)))
        return cat, out

    def _staging(self, nbytes: int):
        """Pinned host block of at least ``nbytes`` + the event of its last use (waited for before it is handed out again)."""
        st = self.__dict__.setdefault("_stage_bufs", {})
        size = 256
        while size < nbytes:
            size *= 2
        ent = st.get(size)
        if ent is None:
            ent = st[size] = (torch.empty(size, dtype=torch.uint8, pin_memory=True), torch.cuda.Event())
        else:
            ent[1].synchronize()
        return ent[0][:nbytes], ent[1]

    # -- the device hot path ------------------------------------------------------------------
    def _stage1(self, uc, un, stage1_k, check_indices):
        # the tower's launch also applies the search's query normalisation (faiss_retrieval.py:147): one launch fewer
        emb = self.two_tower_model.user_tower.encode(uc, un, check_indices=check_indices, renormalize=True)   # :223-227
        return self.faiss_index.search_device(emb, stage1_k, normalize=False,                  # :230-232
                                              return_positions=True)

    def _stage2(self, uc, un, cand_pos, top_k, check_indices, ids_are_positions=False, mark=None, out=None):
        lib = _lib.load()
        B, stage1_k = cand_pos.shape
        if self.cache_ad_projection:
            self.transformer_ranker.ensure_ad_cache(self.ad_features)
        tasks, logits = self.transformer_ranker.score_candidates(uc, un, cand_pos, self.ad_features,  # :241-255
                                                                 check_indices=check_indices, raw=True)
        if mark is not None:                     # amdrec.sharded.StageTimer: the ranker ends here, the selection follows
            mark("ranker")
        if out is not None:                      # (ad_ids, scores) views of one block: the reference API's single D2H copy
            ad_ids, scores = out
        else:
            ad_ids = torch.empty((B, top_k), dtype=torch.int64, device=uc.device)
            scores = torch.empty((len(tasks), B, top_k), dtype=torch.float32, device=uc.device)
        idx = self.faiss_index
        if ids_are_positions:
            cand_ids = cand_pos
        elif idx._identity:
            unfilled = idx._n and (stage1_k > idx._n or idx.index_type == "IVF")     # else every slot is filled
            # an unfilled slot (-1) reads id_map[-1] like the reference's list indexing (faiss_retrieval.py:159-160): one
            # launch (Python-style remainder: -1 -> n - 1, valid positions unchanged) instead of compare + add + where
            cand_ids = torch.remainder(cand_pos, idx._n) if unfilled else cand_pos
        else:
            cand_ids = torch.empty_like(cand_pos)
            _lib.check(lib.amdrec_remap_ids(_lib.ptr(cand_pos), _lib.ptr(idx._ids), idx._n, _lib.ptr(cand_ids),
                                            cand_pos.numel(), _lib.stream_ptr(uc.device)))
        if B:
            _lib.check(lib.amdrec_select_topk(_lib.ptr(logits), logits.stride(0), len(tasks), tasks.index("ctr"),
                                              _lib.ptr(cand_ids), B, stage1_k, top_k, _lib.ptr(ad_ids),
                                              _lib.ptr(scores), None, _lib.stream_ptr(uc.device)))
        return {"ad_ids": ad_ids, "scores": scores, "tasks": tasks, "candidate_ids": cand_ids, "logits": logits}

    @torch.no_grad()
    def recommend_device(self, user_categorical: torch.Tensor, user_numerical: torch.Tensor, top_k: int = 10,
                         stage1_k: int = 500, check_indices: bool = False):
        """[B,6] / [B,13] device tensors -> dict of device tensors, no host synchronisation:
        ad_ids [B,top_k] int64, scores [3,B,top_k] float32 (sigmoid of the logits), candidate_ids
        [B,stage1_k], candidate_scores [B,stage1_k], logits [3, B*stage1_k]."""
        uc = _lib.require_gpu(user_categorical, "user_categorical")
        un = _lib.require_gpu(user_numerical, "user_numerical")
        cand_pos, cand_scores = self._stage1(uc, un, stage1_k, check_indices)
        out = self._stage2(uc, un, cand_pos, top_k, check_indices)
        out["candidate_scores"] = cand_scores
        return out

    def capture(self, batch_size: int, top_k: int = 10, stage1_k: int = 500, warmup: int = 2) -> "GraphedRecommender":
        """Capture one recommend_device call for a fixed batch shape into a HIP graph (the ~40 kernel
        launches of a request are host-launch-bound at small batch: 1.76 ms eager at B = 1) and return a
        replayer.  Weights, index and ad table must not change afterwards."""
        return GraphedRecommender(self, batch_size, top_k, stage1_k, warmup)

    # -- reference API ------------------------------------------------------------------------
    def recommend_ads(self, user_data: dict, top_k: int = 10, stage1_k: int = 500,
                      return_scores: bool = True) -> dict:
        """inference.py:199-288."""
        return self.batch_recommend([user_data], top_k=top_k, stage1_k=stage1_k,
                                    return_scores=return_scores)[0]

    def batch_recommend(self, user_data_list: list, top_k: int = 10, stage1_k: int = 500,
                        return_scores: bool = True) -> list:
        """inference.py:290-331 - but one device pass for the whole list instead of a serial loop.
        ``timing`` reports the batch's stage times divided by the number of users."""
        if not user_data_list:
            return []
        uc, un = self.preprocess_batch(user_data_list)
        return self.recommend_tensors(uc, un, top_k, stage1_k, return_scores, _encoded=True)

    def _user_limits(self):
        """Per user column, the number of rows of the SMALLER of the tower's and the ranker's embedding tables: an index is
        valid iff it is below it (torch raises IndexError otherwise, two_tower_model.py:44 / transformer_ranker.py:318)."""
        key = (_lib._REG_EPOCH[0], self.two_tower_model, self.transformer_ranker)
        c = self.__dict__.get("_limits")
        if c is None or c[0][0] != key[0] or c[0][1] is not key[1] or c[0][2] is not key[2]:
            tw = self.two_tower_model.user_tower.embedding_layer.embeddings
            rk = self.transformer_ranker.user_embeddings
            lim = [min(tw[n].weight.shape[0], rk[n].weight.shape[0]) for n in tw.keys()]
            c = self.__dict__["_limits"] = (key, lim, torch.tensor(lim, dtype=torch.int64, device=self.device))
        return c[1], c[2]

    def _encoder_fits(self) -> bool:
        """True when every index the preprocessor can produce is inside the models' tables: the dict API then needs no
        per-request index check (the label encoder's output is its own class count at most)."""
        pp = self.preprocessor
        lim, _ = self._user_limits()
        c = self.__dict__.get("_enc_fit")
        if c is None or c[0] is not pp or c[1] != lim:                    # (the object itself is kept: an id can be reused)
            cols = [c_ for c_ in USER_COLS if c_ in pp.classes]
            ok = len(cols) == len(lim) and all(len(pp.classes[c_]) <= m for c_, m in zip(cols, lim))
            c = self.__dict__["_enc_fit"] = (pp, list(lim), ok)
        return c[2]

    def _ad_table_valid(self) -> bool:
        """The resident ad-feature table against the ranker's ad embedding tables, checked ONCE per table / model (one
        reduction + host read), not per request: round 3 re-validated all N x 20 indices inside every checked request."""
        t = self.ad_features
        key = (_lib._REG_EPOCH[0], self.transformer_ranker, t, t._version)
        c = self.__dict__.get("_ad_ok")
        if c is None or c[0][0] != key[0] or c[0][1] is not key[1] or c[0][2] is not key[2] or c[0][3] != key[3]:
            cards = torch.tensor([e.weight.shape[0] for e in self.transformer_ranker.ad_embeddings.values()],
                                 dtype=torch.int64, device=t.device)
            ok = t.numel() == 0 or not bool(((t < 0) | (t >= cards)).any().item())
            c = self.__dict__["_ad_ok"] = (key, ok)
        return c[1]

    @torch.no_grad()
    def recommend_tensors(self, user_categorical, user_numerical, top_k=10, stage1_k=500, return_scores=True,
                          _encoded=False):
        """Tensor-level entry (cf. TwoStageRetriever.retrieve_and_rank, faiss_retrieval.py:283-369); result dicts follow
        inference.py:272-288.  ONE host synchronisation per call (round 3 had four: the tower's index flag, the stage-1
        timing sync, the ranker's index flag, the result copy - each exposing the launch work queued behind it): indices
        are validated without a read-back in the middle (the verdict travels with the results), the stage times come from
        events, ids + scores + verdict come back in one copy.  An out-of-range index raises IndexError like the reference's
        embedding lookup, before any result is returned."""
        t0 = time.time()
        dev = self.device
        uc = user_categorical.to(dev)
        un = user_numerical.to(dev)
        n = uc.shape[0]
        if not n:
            return []
        tasks = list(self.transformer_ranker.prediction_heads.keys())        # the ranker's task order (= out["tasks"])
        ev = self.__dict__.get("_ev")
        if ev is None:
            ev = self.__dict__["_ev"] = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        ids_b, sc_b = n * top_k * 8, len(tasks) * n * top_k * 4
        blk = torch.empty(ids_b + sc_b + 8, dtype=torch.uint8, device=dev)
        ad_ids = blk[:ids_b].view(torch.int64).view(n, top_k)
        scores = blk[ids_b:ids_b + sc_b].view(torch.float32).view(len(tasks), n, top_k)
        verdict = blk[ids_b + sc_b:].view(torch.int32)                       # [2]: nonzero = an index out of range
        trusted = _encoded and self.preprocessor is not None and self._encoder_fits()
        if trusted:
            verdict.zero_()
        else:
            if uc.dim() != 2 or uc.shape[1] != len(self._user_limits()[0]):
                raise ValueError(f"user_categorical must be [B, {len(self._user_limits()[0])}]")
            ucl = uc.long()
            verdict.copy_(((ucl < 0) | (ucl >= self._user_limits()[1])).any().to(torch.int32).expand(2))
        if not self._ad_table_valid():
            raise IndexError("index out of range in self")                    # (the ad-feature table, transformer_ranker.py:322)
        st = torch.cuda.current_stream(dev)
        ev[0].record(st)
        cand_pos, _ = self._stage1(uc, un, stage1_k, False)
        ev[1].record(st)
        out = self._stage2(uc, un, cand_pos, top_k, False, out=(ad_ids, scores))
        assert list(out["tasks"]) == tasks
        host, done = self._staging(blk.numel())
        host.copy_(blk, non_blocking=True)
        ev[2].record(st)
        done.record(st)
        ev[2].synchronize()
        hv = host.numpy()
        if int(hv[ids_b + sc_b:].view(np.int32)[0]):
            raise IndexError("index out of range in self")                    # torch.nn.Embedding's message
        ids = hv[:ids_b].view(np.int64).reshape(n, top_k).tolist()
        sc = hv[ids_b:ids_b + sc_b].view(np.float32).reshape(len(tasks), n, top_k).tolist() if return_scores else None
        t2 = time.time()
        # stage 1 = its GPU time (tower + search, as the reference brackets them); stage 2 = the rest of the call's wall time
        # (ranker, selection, the copy back and the list conversion: what the reference's second bracket holds)
        total_ms = (t2 - t0) * 1000 / n
        stage1_ms = min(ev[0].elapsed_time(ev[1]) / n, total_ms)
        res = []
        for b in range(n):
            r = {"ad_ids": ids[b],
                 "timing": {"stage1_ms": stage1_ms, "stage2_ms": total_ms - stage1_ms, "total_ms": total_ms}}
            if return_scores:
                r["scores"] = {t: sc[i][b] for i, t in enumerate(tasks)}
            res.append(r)
        return res


class TwoStageRetriever:
    """Drop-in for faiss_retrieval.py:259-369: the reference's second caller of the same path, taking tensors
    instead of dicts.  ``retrieve_and_rank`` keeps the reference's signature and tuple-of-lists return.
    ``ad_features_lookup``: None -> stage 1 only, returns (candidate ids, distances) exactly like the reference
    (:329-331); otherwise the ``[N, n_ad_feat]`` integer table of ad features indexed by corpus position (the
    reference collects per-id dicts and then scores all-zero placeholders, :338-345 - a documented stub)."""

    def __init__(self, two_tower_model: TwoTowerModel, transformer_ranker: TransformerRanker, faiss_index: FAISSIndex,
                 device: str = "cuda"):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.AmdrecError("TwoStageRetriever needs a HIP device (no CPU fallback)")
        self.two_tower_model = two_tower_model.to(self.device).eval()
        self.transformer_ranker = transformer_ranker.to(self.device).eval()
        self.faiss_index = faiss_index
        self._rec = None

    @torch.no_grad()
    def retrieve_and_rank(self, user_categorical: torch.Tensor, user_numerical: torch.Tensor, stage1_k: int = 500,
                          stage2_k: int = 10, ad_features_lookup=None):
        uc = user_categorical.to(self.device)
        un = user_numerical.to(self.device, dtype=torch.float32)
        if ad_features_lookup is None:                                            # :329-331
            emb = self.two_tower_model.get_user_embeddings(uc, un)
            ids, dist = self.faiss_index.search_device(emb, stage1_k)
            return ids[0].tolist(), dist[0].tolist()
        if self._rec is None or self._rec_table is not ad_features_lookup:
            self._rec = AdRecommenderInference(device=str(self.device), two_tower_model=self.two_tower_model,
                                               transformer_ranker=self.transformer_ranker,
                                               faiss_index=self.faiss_index, ad_features=ad_features_lookup)
            self._rec_table = ad_features_lookup
        r = self._rec.recommend_tensors(uc[:1], un[:1], stage2_k, stage1_k)[0]    # one synchronisation, indices validated
        return r["ad_ids"], r["scores"]["ctr"]                                    # :359-369 (ids, ctr probabilities)


class GraphedRecommender:
    """hipGraph replay of AdRecommenderInference.recommend_device for one (batch, top_k, stage1_k) shape.
    The graph owns everything its kernel nodes point at: static input/output tensors (torch's graph
    memory pool) and a private, fixed-size workspace (never the shared grow-only one)."""

    def __init__(self, rec: AdRecommenderInference, batch_size: int, top_k: int, stage1_k: int, warmup: int = 2):
        self.rec, self.batch_size, self.top_k, self.stage1_k = rec, batch_size, top_k, stage1_k
        dev = rec.ad_features.device
        n_cat = len(rec.two_tower_model.user_tower._names)
        n_num = rec.two_tower_model.user_tower._n_num
        self._uc = torch.zeros((batch_size, n_cat), dtype=torch.int64, device=dev)
        self._un = torch.zeros((batch_size, n_num), dtype=torch.float32, device=dev)
        # sizing + warm-up pass on the shared workspace (packs weights, sets kernel attributes)
        probe = _lib.MeasuringArena(_lib.Workspace())
        with _lib.WORKSPACE.private(probe):
            for _ in range(max(1, warmup)):
                rec.recommend_device(self._uc, self._un, top_k, stage1_k)
        torch.cuda.synchronize(dev)
        self._arena = _lib.FixedArena(probe.high_water, dev)
        with _lib.WORKSPACE.private(self._arena):
            rec.recommend_device(self._uc, self._un, top_k, stage1_k)          # one eager pass on the private arena
            torch.cuda.synchronize(dev)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._out = rec.recommend_device(self._uc, self._un, top_k, stage1_k)
        torch.cuda.synchronize(dev)
        # pin every device buffer the kernel nodes point at: a later load_state_dict / index.add() then
        # makes the graph stale (documented) but can never leave it with dangling pointers
        idx = rec.faiss_index
        self._pinned = (rec.two_tower_model.user_tower._packed, rec.transformer_ranker._packed, idx._xb, idx._ids,
                        idx._xb16, idx._maxnorm, rec.ad_features, rec.transformer_ranker._ad_cache, getattr(idx, "_ivf", None) and idx._ivf._lists,
                        getattr(idx, "_ivf", None) and getattr(idx._ivf, "_shadow", None))

    @torch.no_grad()
    def __call__(self, user_categorical: torch.Tensor, user_numerical: torch.Tensor):
        """Device tensors [batch_size, ...] -> the same dict as recommend_device (static output buffers,
        overwritten by the next call)."""
        if user_categorical.shape[0] != self.batch_size:
            raise ValueError(f"captured for batch {self.batch_size}, got {user_categorical.shape[0]}")
        self._uc.copy_(user_categorical)
        self._un.copy_(user_numerical)
        self._graph.replay()
        return self._out


def build_faiss_index(model: TwoTowerModel, ad_categorical, device="cuda", save_path: Optional[str] = None,
                      batch_size: int = 1 << 18, index_type: str = "IVF", nlist: int = 100,
                      nprobe: int = 10) -> FAISSIndex:
    """Corpus build (training_pipeline.py:488-546): AdTower over every row in dataset order
    (shuffle=False, :513), ad id = row index (:523), add to the index, optionally save.
    ``ad_categorical`` is the [N, 20] integer table; embeddings never leave the device."""
    model = model.to(device).eval()
    table = ad_categorical if isinstance(ad_categorical, torch.Tensor) else torch.from_numpy(
        np.ascontiguousarray(ad_categorical))
    idx = FAISSIndex(model.output_dim, index_type=index_type, nlist=nlist, nprobe=nprobe, device=device)
    embs = []
    with torch.no_grad():
        for s in range(0, table.shape[0], batch_size):
            embs.append(model.get_ad_embeddings(table[s:s + batch_size].to(device)))
    emb = torch.cat(embs) if embs else torch.empty((0, model.output_dim), device=device)
    idx.add(emb)                                         # default ids = arange (:523 / faiss_retrieval.py:121)
    if save_path:
        idx.save(save_path)
    return idx
