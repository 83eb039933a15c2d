"""Device-resident retrieval index: drop-in for the reference's ``FAISSIndex`` wrapper
(faiss_retrieval.py:14-256) with the faiss calls replaced by libamdrec HIP kernels.

Same constructor arguments, method names, argument meaning and return conventions:
``search`` returns ``(ad_ids, distances)`` in that order (faiss_retrieval.py:164-166),
numpy in / numpy out, inputs are never mutated (the wrapper copies via ``astype``,
:114, :146), default ids are a running ``arange`` (:121-123), unknown ``index_type`` raises
``ValueError`` (:73).  Extra, for the on-device pipeline: ``search_device`` takes and returns
device tensors and never synchronises with the host.

Layout in HBM: corpus ``[capacity, dimension]`` float32 row-major, rows L2-normalised at
``add`` time, 1 KiB per row at d=256 (1.024 GB per 1M ads: the whole 10M corpus of
BASELINE config 4 is 10.24 GB, 3.6 % of one MI355X's 288 GB); ids int64 ``[capacity]``.
"""
from __future__ import annotations

import json
import os
import struct
import time
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import _lib

_MAGIC = b"AMDRECIX1"
INDEX_TYPES = ("Flat", "IVF", "IVFPQ", "HNSW")


def _encode_id(x):
    """JSON form of one arbitrary ad id (the reference pickles id_map, faiss_retrieval.py:208-218; this format never
    unpickles): [tag, value] with tag i / f / s / n.  Other types are refused at save time."""
    if isinstance(x, (bool, np.bool_)):
        raise TypeError("ad ids of type bool cannot be saved; use int or str")
    if isinstance(x, (int, np.integer)):
        return ["i", int(x)]
    if isinstance(x, (float, np.floating)):
        return ["f", float(x)]
    if isinstance(x, str):
        return ["s", x]
    if x is None:
        return ["n", None]
    raise TypeError(f"ad ids of type {type(x).__name__} cannot be saved; use int, float, str or None")


def _decode_id(e):
    if isinstance(e, str):            # files written by the round-1 format (ids stringified)
        return e
    tag, v = e
    return {"i": int, "f": float, "s": str, "n": lambda _: None}[tag](v)


class _Handle:
    """The attributes of a faiss index object that the reference touches
    (``index.ntotal``, ``index.is_trained``, ``index.nprobe``: faiss_retrieval.py:90, :127, :150)."""

    def __init__(self, owner):
        self._o = owner

    @property
    def ntotal(self):
        return self._o._n

    @property
    def is_trained(self):
        return self._o._trained

    @property
    def nprobe(self):
        if self._o.index_type != "IVF":
            raise AttributeError("nprobe")
        return self._o.nprobe

    @nprobe.setter
    def nprobe(self, v):
        self._o.nprobe = int(v)


class FAISSIndex:
    def __init__(self, dimension: int, index_type: str = "IVF", nlist: int = 100, nprobe: int = 10,
                 use_gpu: bool = False, device=None, verbose: bool = False, prefilter: str = "bf16"):
        """``use_gpu`` is accepted for signature compatibility; the index always lives on the
        HIP device (``device`` or the current one) - there is no CPU engine.
        ``prefilter`` (Flat only): "bf16" keeps a bf16 copy of the corpus next to the fp32 one and searches with
        amdrec_flat_search_mixed (bf16 MFMA filter, fp32 re-score, certified exact); "fp32" = amdrec_flat_search."""
        if prefilter not in ("bf16", "fp32"):
            raise ValueError("prefilter must be 'bf16' or 'fp32'")
        self.prefilter = prefilter
        self.dimension = int(dimension)
        self.index_type = index_type
        self.nlist = int(nlist)
        self.nprobe = int(nprobe)
        self.use_gpu = use_gpu
        self.verbose = verbose
        self.device = torch.device(device if device is not None else "cuda")
        self._create_index()

    # -- construction ---------------------------------------------------------------
    def _create_index(self):
        if self.index_type not in INDEX_TYPES:
            raise ValueError(f"Unknown index type: {self.index_type}")          # :73
        if self.index_type in ("IVFPQ", "HNSW"):
            raise NotImplementedError(
                f"{self.index_type} is outside the MI355X hot path (SURVEY.md §2 #12): use 'Flat' or 'IVF'")
        if self.dimension % 4 or not (4 <= self.dimension <= 2048):
            raise ValueError("dimension must be a multiple of 4 in [4, 2048]")
        _lib.load()
        self._xb = torch.empty((0, self.dimension), dtype=torch.float32, device=self.device)
        self._ids = torch.empty((0,), dtype=torch.int64, device=self.device)
        self._n = 0
        # bf16 shadow of the corpus + its largest row norm (the mixed search's error bound)
        self._mixed = self.index_type == "Flat" and self.prefilter == "bf16" and self.dimension % 8 == 0
        self._xb16 = torch.empty((0, self.dimension), dtype=torch.bfloat16, device=self.device)
        self._maxnorm = torch.zeros(2, dtype=torch.float32, device=self.device)     # [max row norm, max row rounding-error norm]
        self._identity = True          # ids == arange(n): remap is the identity
        self._host_ids: Optional[list] = None   # only for non-integer ids
        self._trained = self.index_type == "Flat"
        self._ivf = None               # set by train() for IVF
        self.index = _Handle(self)
        self._log(f"Created {self.index_type} index with dimension {self.dimension}")

    def _log(self, msg):
        if self.verbose:
            print(msg)

    # -- helpers --------------------------------------------------------------------
    def _to_device_f32(self, a) -> torch.Tensor:
        """fp32 device COPY of the input (``astype('float32')`` at :114 / :146 copies)."""
        if isinstance(a, torch.Tensor):
            t = a.detach().to(device=self.device, dtype=torch.float32, copy=True)
        else:
            t = torch.from_numpy(np.ascontiguousarray(np.asarray(a), dtype=np.float32)).to(self.device)
        if t.dim() != 2 or t.shape[1] != self.dimension:
            raise ValueError(f"expected [n, {self.dimension}] embeddings, got {tuple(t.shape)}")
        return t.contiguous()

    def _normalize_(self, t: torch.Tensor) -> torch.Tensor:
        lib = _lib.load()
        if t.shape[0] == 0:
            return t
        _lib.check(lib.amdrec_l2_normalize(_lib.ptr(t), t.stride(0), _lib.ptr(t), t.stride(0), t.shape[0],
                                           self.dimension, _lib.stream_ptr(self.device)))
        return t

    def _reserve(self, n):
        cap = self._xb.shape[0]
        if n <= cap:
            return
        new_cap = max(n, int(cap * 1.5), 1024)
        xb = torch.empty((new_cap, self.dimension), dtype=torch.float32, device=self.device)
        ids = torch.empty((new_cap,), dtype=torch.int64, device=self.device)
        if self._n:
            xb[:self._n].copy_(self._xb[:self._n])
            ids[:self._n].copy_(self._ids[:self._n])
        self._xb, self._ids = xb, ids
        if self._mixed:
            xb16 = torch.empty((new_cap, self.dimension), dtype=torch.bfloat16, device=self.device)
            if self._n:
                xb16[:self._n].copy_(self._xb16[:self._n])
            self._xb16 = xb16

    def _shadow_rows(self, lo, hi):
        """(Re)build rows [lo, hi) of the bf16 shadow from the fp32 rows and fold their norms into _maxnorm."""
        if not self._mixed or hi <= lo:
            return
        lib = _lib.load()
        x, y = self._xb[lo:hi], self._xb16[lo:hi]
        _lib.check(lib.amdrec_bf16_rows(_lib.ptr(x), hi - lo, x.stride(0), self.dimension, _lib.ptr(y), y.stride(0),
                                        _lib.ptr(self._maxnorm), _lib.stream_ptr(self.device)))

    # -- reference API ----------------------------------------------------------------
    def train(self, embeddings):
        """faiss_retrieval.py:83-95: trains the coarse quantizer of IVF; no-op for Flat."""
        if self._trained:
            return
        t0 = time.time()
        self._log(f"Training index on {len(embeddings)} samples...")
        from . import ivf
        self._ivf = ivf.IVFState.train(self._to_device_f32(embeddings), self.nlist)
        self._trained = True
        self._log(f"Index trained in {time.time() - t0:.2f}s")

    def set_trained_centroids(self, centroids):
        """Install an already trained coarse quantizer (IVF): ``centroids`` [nlist, dimension], unit rows.  This is how
        the ranks of a sharded index share ONE quantizer (amdrec.sharded.share_ivf_centroids) -
        faiss would do the same with ``index_ivf.quantizer`` handed to every shard."""
        if self.index_type != "IVF":
            raise ValueError("only an IVF index has a coarse quantizer")
        if self._n:
            raise ValueError("set the centroids before adding vectors")
        from . import ivf
        c = self._to_device_f32(centroids)
        if c.shape[0] != self.nlist:
            raise ValueError(f"expected {self.nlist} centroids, got {c.shape[0]}")
        self._ivf = ivf.IVFState(c)
        self._trained = True

    @property
    def centroids(self):
        """The trained coarse quantizer [nlist, dimension] (device tensor), or None."""
        return None if self._ivf is None else self._ivf.centroids

    def add(self, embeddings, ad_ids: Optional[List] = None):
        """faiss_retrieval.py:97-127."""
        if not self._trained:
            self.train(embeddings)                                   # :107-108 (un-normalised input)
        t0 = time.time()
        # fp32 copy of the input straight into index storage, renormalised there (:114-118): the
        # caller's array is never modified and no second device copy is made
        if isinstance(embeddings, torch.Tensor):
            src = embeddings.detach()
        else:
            src = torch.from_numpy(np.ascontiguousarray(np.asarray(embeddings), dtype=np.float32))
        if src.dim() != 2 or src.shape[1] != self.dimension:
            raise ValueError(f"expected [n, {self.dimension}] embeddings, got {tuple(src.shape)}")
        m = src.shape[0]
        self._reserve(self._n + m)
        x = self._xb[self._n:self._n + m]
        x.copy_(src)                                                 # casts + moves to the device
        self._normalize_(x)
        self._shadow_rows(self._n, self._n + m)
        if ad_ids is None:                                           # :121-122
            new_ids = torch.arange(self._n, self._n + m, dtype=torch.int64, device=self.device)
            if self._host_ids is not None:
                self._host_ids.extend(range(self._n, self._n + m))
        else:
            if len(ad_ids) != m:
                raise ValueError("len(ad_ids) != len(embeddings)")
            try:
                arr = np.asarray(ad_ids)
                if arr.dtype.kind not in "iu":
                    raise TypeError
                new_ids = torch.from_numpy(arr.astype(np.int64)).to(self.device)
                ident = bool(m == 0 or (arr[0] == self._n and np.array_equal(arr, np.arange(self._n, self._n + m))))
                self._identity = self._identity and ident
                if self._host_ids is not None:
                    self._host_ids.extend(arr.tolist())
            except TypeError:
                # arbitrary Python objects as ids: keep them on the host (plumbing, not compute)
                if self._host_ids is None:
                    self._host_ids = self._ids[:self._n].tolist()
                self._host_ids.extend(list(ad_ids))
                self._identity = False
                new_ids = torch.full((m,), -1, dtype=torch.int64, device=self.device)
        self._ids[self._n:self._n + m].copy_(new_ids)                # :123
        if self._ivf is not None:
            self._ivf.append(x, self._n)
        self._n += m
        self._log(f"Added embeddings in {time.time() - t0:.2f}s")
        self._log(f"Total index size: {self._n}")

    @property
    def id_map(self) -> list:
        if self._host_ids is not None:
            return self._host_ids
        return self._ids[:self._n].tolist()

    def search_device(self, queries: torch.Tensor, k: int, normalize: bool = True,
                      return_positions: bool = False, pos_offset: int = 0):
        """Device-to-device search, asynchronous on the current stream.
        -> (ids int64 [nq,k], scores float32 [nq,k]) on the device.  ``return_positions``: corpus
        positions (+ ``pos_offset``, the shard's first global row) instead of ids, -1 = unfilled."""
        q = _lib.require_gpu(queries, "queries")
        if q.dim() != 2 or q.shape[1] != self.dimension:
            raise ValueError(f"expected [nq, {self.dimension}] queries, got {tuple(q.shape)}")
        if normalize:
            # faiss.normalize_L2 on the wrapper's copy (:146-147): ONE out-of-place launch (a float32 contiguous input is
            # read where it lies; the caller's tensor is never written) instead of a copy launch + an in-place one
            src = q.to(dtype=torch.float32).contiguous()
            q = torch.empty_like(src)
            if src.shape[0]:
                _lib.check(_lib.load().amdrec_l2_normalize(_lib.ptr(src), src.stride(0), _lib.ptr(q), q.stride(0), src.shape[0],
                                                           self.dimension, _lib.stream_ptr(self.device)))
        else:
            q = q.contiguous()
        nq = q.shape[0]
        scores = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        pos = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        if self.index_type == "IVF":
            self._ivf.search(self._xb, self._n, q, k, self.nprobe, scores, pos,
                             pos_offset=pos_offset if return_positions else 0)
        elif self._mixed:
            flat_search_mixed(self._xb, self._xb16, self._maxnorm, self._n, q, k, scores, pos,
                              pos_offset=pos_offset if return_positions else 0)
        else:
            flat_search(self._xb, self._n, q, k, scores, pos, pos_offset=pos_offset if return_positions else 0)
        if return_positions or self._identity:
            # identity map: id == position for filled slots; unfilled (-1) slots map to
            # id_map[-1] in the reference (:159) - reproduce that too
            if return_positions:
                return pos, scores
            if self._n and (k > self._n or self.index_type == "IVF"):   # only then can a slot be unfilled
                pos = torch.where(pos < 0, pos + self._n, pos)
            return pos, scores
        lib = _lib.load()
        ids = torch.empty_like(pos)
        _lib.check(lib.amdrec_remap_ids(_lib.ptr(pos), _lib.ptr(self._ids), self._n, _lib.ptr(ids), pos.numel(),
                                        _lib.stream_ptr(self.device)))
        return ids, scores

    def search(self, query_embeddings, k: int = 100, return_distances: bool = True):
        """faiss_retrieval.py:129-166.  numpy in, numpy out: (ad_ids, distances)."""
        q = self._to_device_f32(query_embeddings)
        t0 = time.time()
        if self._host_ids is not None:
            pos, scores = self.search_device(q, k, normalize=True, return_positions=True)
            idm = np.asarray(self._host_ids, dtype=object)
            ad_ids = idm[pos.cpu().numpy()]                          # pos == -1 -> id_map[-1]
        else:
            ids, scores = self.search_device(q, k, normalize=True)
            ad_ids = ids.cpu().numpy()
        distances = scores.cpu().numpy()
        self._log(f"Search completed in {(time.time() - t0) * 1000:.2f}ms for {len(q)} queries")
        if return_distances:
            return ad_ids, distances
        return ad_ids

    def batch_search(self, query_embeddings, k: int = 100, batch_size: int = 1000):
        """faiss_retrieval.py:168-194."""
        all_ids, all_d = [], []
        for i in range(0, len(query_embeddings), batch_size):
            ids, d = self.search(query_embeddings[i:i + batch_size], k)
            all_ids.append(ids)
            all_d.append(d)
        return np.vstack(all_ids), np.vstack(all_d)

    # -- persistence: own format (faiss' write_index binary is unreadable without faiss) --
    def save(self, filepath: str):
        """faiss_retrieval.py:196-221: index file + ``<path>.metadata`` sidecar.  The index file is
        ``AMDRECIX1 | u64 header_len | json header | raw arrays``; the sidecar is JSON with the
        reference's metadata fields (dimension, index_type, nlist, nprobe) - never pickle."""
        d = os.path.dirname(os.path.abspath(filepath))
        os.makedirs(d, exist_ok=True)
        arrays = [("xb", self._xb[:self._n].cpu().numpy()), ("ids", self._ids[:self._n].cpu().numpy())]
        if self._ivf is not None:
            arrays += self._ivf.export_arrays()
        header = {"dimension": self.dimension, "index_type": self.index_type, "nlist": self.nlist,
                  "nprobe": self.nprobe, "ntotal": self._n, "identity_ids": self._identity,
                  "arrays": [{"name": n, "dtype": str(a.dtype), "shape": list(a.shape)} for n, a in arrays]}
        if self._host_ids is not None:
            header["host_ids"] = [_encode_id(x) for x in self._host_ids]     # typed: ids round-trip as what they were
        hj = json.dumps(header).encode()
        with open(filepath, "wb") as f:
            f.write(_MAGIC)
            f.write(struct.pack("<Q", len(hj)))
            f.write(hj)
            for _, a in arrays:
                f.write(np.ascontiguousarray(a).tobytes())
        with open(filepath + ".metadata", "w") as f:
            json.dump({k: header[k] for k in ("dimension", "index_type", "nlist", "nprobe", "ntotal")}, f)
        self._log(f"Index saved to {filepath}")

    def load(self, filepath: str):
        """faiss_retrieval.py:223-245."""
        with open(filepath, "rb") as f:
            if f.read(len(_MAGIC)) != _MAGIC:
                raise ValueError(f"{filepath} is not an amdrec index file")
            (hl,) = struct.unpack("<Q", f.read(8))
            header = json.loads(f.read(hl).decode())
            arrays = {}
            for spec in header["arrays"]:
                dt = np.dtype(spec["dtype"])
                cnt = int(np.prod(spec["shape"])) if spec["shape"] else 1
                arrays[spec["name"]] = np.frombuffer(f.read(cnt * dt.itemsize), dtype=dt).reshape(spec["shape"])
        self.dimension = header["dimension"]
        self.index_type = header["index_type"]
        self.nlist = header["nlist"]
        self.nprobe = header["nprobe"]
        self._create_index()
        n = header["ntotal"]
        self._reserve(n)
        self._xb[:n].copy_(torch.from_numpy(arrays["xb"].copy()))
        self._ids[:n].copy_(torch.from_numpy(arrays["ids"].copy()))
        self._n = n
        self._shadow_rows(0, n)
        self._identity = header["identity_ids"]
        hid = header.get("host_ids")
        self._host_ids = None if hid is None else [_decode_id(x) for x in hid]
        if self.index_type == "IVF":
            from . import ivf
            self._ivf = ivf.IVFState.from_arrays(arrays, self.device)
            self._trained = True
        self._log(f"Index loaded from {filepath}")
        self._log(f"Index size: {self._n}")

    def get_stats(self):
        """faiss_retrieval.py:247-256."""
        return {"index_type": self.index_type, "dimension": self.dimension, "num_vectors": self._n,
                "is_trained": self._trained, "nlist": self.nlist, "nprobe": self.nprobe}


def benchmark_faiss_index(dimension: int = 256, num_vectors: int = 1000000, num_queries: int = 100, k: int = 100,
                          device="cuda", seed: int = 1234):
    """The reference's only benchmark (faiss_retrieval.py:372-436): random corpus, add + search timings per
    index type.  The Flat and IVF arms are built here (IVFPQ / HNSW are outside the hot path); vectors and
    queries are drawn on the device (randn, as :390-391, seeded here)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    vectors = torch.randn((num_vectors, dimension), generator=g, device=device)
    queries = torch.randn((num_queries, dimension), generator=g, device=device)
    results = {}
    for index_type, config in (("Flat", {}), ("IVF", {"nlist": 100, "nprobe": 10})):
        idx = FAISSIndex(dimension, index_type=index_type, device=device, **config)
        torch.cuda.synchronize()
        t0 = time.time()
        idx.add(vectors)
        torch.cuda.synchronize()
        add_time = time.time() - t0
        idx.search_device(queries, k)                                  # warm-up (lists, workspace)
        torch.cuda.synchronize()
        t0 = time.time()
        idx.search_device(queries, k)
        torch.cuda.synchronize()
        ms = (time.time() - t0) * 1000
        results[index_type] = {"add_time": add_time, "search_time_ms": ms, "per_query_ms": ms / num_queries}
    return results


def flat_search(xb: torch.Tensor, n: int, q: torch.Tensor, k: int, out_scores: torch.Tensor,
                out_pos: torch.Tensor, pos_offset: int = 0, n_fixup: Optional[torch.Tensor] = None):
    """amdrec_flat_search on device tensors (rows of xb[:n] and q already L2-normalised)."""
    lib = _lib.load()
    dev = q.device
    if q.shape[0] == 0:
        return
    nbytes = _lib.C.c_size_t(0)
    _lib.check(lib.amdrec_flat_search_workspace(q.shape[0], n, k, _lib.C.byref(nbytes)))
    ws = _lib.WORKSPACE.get(nbytes.value, dev)
    _lib.check(lib.amdrec_flat_search(
        _lib.ptr(xb), n, xb.stride(0) if xb.dim() == 2 and xb.shape[0] > 0 else xb.shape[-1], xb.shape[-1],
        _lib.ptr(q), q.shape[0], q.stride(0), k, pos_offset, _lib.ptr(out_scores), _lib.ptr(out_pos),
        _lib.ptr(ws), ws.numel(), _lib.ptr(n_fixup), _lib.stream_ptr(dev)))


def flat_search_mixed(xb: torch.Tensor, xb16: torch.Tensor, max_norm: torch.Tensor, n: int, q: torch.Tensor, k: int,
                      out_scores: torch.Tensor, out_pos: torch.Tensor, pos_offset: int = 0,
                      n_fixup: Optional[torch.Tensor] = None):
    """amdrec_flat_search_mixed on device tensors: xb16 = amdrec_bf16_rows(xb), max_norm = its largest row norm."""
    lib = _lib.load()
    dev = q.device
    if q.shape[0] == 0:
        return
    d = xb.shape[-1]
    nbytes = _lib.C.c_size_t(0)
    _lib.check(lib.amdrec_flat_search_mixed_workspace(q.shape[0], n, k, d, _lib.C.byref(nbytes)))
    ws = _lib.WORKSPACE.get(nbytes.value, dev)
    has = xb.dim() == 2 and xb.shape[0] > 0
    _lib.check(lib.amdrec_flat_search_mixed(
        _lib.ptr(xb), n, xb.stride(0) if has else d, d, _lib.ptr(xb16), xb16.stride(0) if has else d,
        _lib.ptr(max_norm), _lib.ptr(q), q.shape[0], q.stride(0), k, pos_offset, _lib.ptr(out_scores),
        _lib.ptr(out_pos), _lib.ptr(ws), ws.numel(), _lib.ptr(n_fixup), _lib.stream_ptr(dev)))
