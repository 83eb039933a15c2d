"""IVF-Flat state for ``FAISSIndex(index_type='IVF')`` (faiss_retrieval.py:50-55): coarse centroids,
list assignment, list-contiguous copy of the corpus, and the search driver over libamdrec's
``amdrec_flat_search`` (coarse probes) + ``amdrec_ivf_scan`` + ``amdrec_ivf_select``.

Training (k-means) and assignment run at index-BUILD time with plain torch matmuls on the device
(offline plumbing; the hot path - search - is hand-written HIP).  faiss' own k-means (its
sub-sampling, seeding and iteration details) is not reproducible offline, so, as SURVEY.md §8c says,
IVF parity is: (i) the scan is exact over the probed lists given this build's centroids and
assignments (tested against the oracle), (ii) recall@k against the Flat result is reported.
Trainer: spherical Lloyd, max-inner-product assignment (the quantizer is IndexFlatIP), <= 256
training points per centroid, 10 iterations, seed 1234 (faiss' documented defaults).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch

from . import _lib

NITER = 10
MAX_POINTS_PER_CENTROID = 256
SEED = 1234
POOL_BYTES = 2 << 30           # candidate-pool workspace per query chunk
GROUPED_MIN_QUERIES = 16       # batches at least this large scan list-major (every list read once per 64 queries)
QTILE = 64                     # queries per grouped-scan tile (ShapeIvf::BP)


def _normalize(x):
    return x / x.norm(dim=1, keepdim=True).clamp_min(1e-30)


def _assign(x: torch.Tensor, cent: torch.Tensor, chunk: int = 1 << 16) -> torch.Tensor:
    out = torch.empty(x.shape[0], dtype=torch.int64, device=x.device)
    ct = cent.t().contiguous()
    for s in range(0, x.shape[0], chunk):
        out[s:s + chunk] = torch.argmax(x[s:s + chunk] @ ct, dim=1)
    return out


class IVFState:
    def __init__(self, centroids: torch.Tensor):
        self.centroids = centroids.contiguous()                    # [nlist, d], unit rows
        self.nlist, self.dim = centroids.shape
        self.device = centroids.device
        self.assign = torch.empty(0, dtype=torch.int64, device=self.device)   # list of every position
        self._lists = None                                          # (xs, spos, list_off, list_len, max_len)

    # -- build ----------------------------------------------------------------------------
    @classmethod
    def train(cls, x: torch.Tensor, nlist: int) -> "IVFState":
        """x: fp32 device copy of the training embeddings (un-normalised, as FAISSIndex.add passes
        them to train(), faiss_retrieval.py:107-108)."""
        x = _normalize(x.float())
        n = x.shape[0]
        if n < nlist:
            raise ValueError(f"need at least nlist={nlist} training vectors, got {n}")
        g = torch.Generator(device="cpu")
        g.manual_seed(SEED)
        if n > MAX_POINTS_PER_CENTROID * nlist:
            sel = torch.randperm(n, generator=g)[:MAX_POINTS_PER_CENTROID * nlist].to(x.device)
            x = x[sel]
            n = x.shape[0]
        cent = x[torch.randperm(n, generator=g)[:nlist].to(x.device)].clone()
        for _ in range(NITER):
            a = _assign(x, cent)
            order = torch.argsort(a, stable=True)
            counts = torch.bincount(a, minlength=nlist)
            try:                                                     # deterministic segmented sum
                sums = torch.segment_reduce(x[order], "sum", lengths=counts, axis=0)
            except Exception:                                        # pragma: no cover
                sums = torch.zeros_like(cent).index_add_(0, a, x)
            nonempty = counts > 0
            cent = torch.where(nonempty[:, None], _normalize(sums), cent)
        return cls(cent)

    def append(self, x_normalised: torch.Tensor, start: int):
        assert start == self.assign.shape[0]
        self.assign = torch.cat([self.assign, _assign(x_normalised, self.centroids)])
        self._lists = None

    def _build_lists(self, xb: torch.Tensor, n: int):
        if self._lists is None or self._lists[5] != n:
            a = self.assign[:n]
            order = torch.argsort(a, stable=True)                   # rows of a list keep insertion order
            counts = torch.bincount(a, minlength=self.nlist)
            off = torch.zeros(self.nlist + 1, dtype=torch.int64, device=self.device)
            off[1:] = torch.cumsum(counts, 0)
            xs = xb[:n][order].contiguous()
            self._lists = (xs, order.contiguous(), off, counts.to(torch.int64), int(counts.max().item()) if n else 0, n)
        return self._lists

    # -- search ---------------------------------------------------------------------------
    def search(self, xb: torch.Tensor, n: int, q: torch.Tensor, k: int, nprobe: int, out_scores: torch.Tensor,
               out_pos: torch.Tensor, pos_offset: int = 0):
        from .index import flat_search
        lib = _lib.load()
        nq = q.shape[0]
        if nq == 0:
            return
        if n == 0:
            out_scores.fill_(float("-inf"))
            out_pos.fill_(-1)
            return
        xs, spos, off, lens, max_len, _ = self._build_lists(xb, n)
        nprobe = max(1, int(nprobe))
        # 1. coarse quantizer: nprobe best centroids by inner product (IndexFlatIP quantizer)
        cs = torch.empty((nq, nprobe), dtype=torch.float32, device=self.device)
        probes = torch.empty((nq, nprobe), dtype=torch.int64, device=self.device)
        flat_search(self.centroids, self.nlist, q, nprobe, cs, probes)
        # 2. pool layout (tiny [nq, nprobe] integer plumbing)
        plen = torch.where(probes >= 0, lens[probes.clamp_min(0)], torch.zeros_like(probes))
        base = (torch.cumsum(plen, 1) - plen).contiguous()
        n_pool = plen.sum(1).contiguous()
        pool_ld = max(1, nprobe * max_len)
        chunk = max(1, min(nq, 65535, POOL_BYTES // (pool_ld * 8)))
        grouped = nq >= GROUPED_MIN_QUERIES
        if grouped:                                   # one launch needs pairs/64 + nlist <= 65535 query tiles
            chunk = max(1, min(chunk, ((65535 - self.nlist) * QTILE) // nprobe))
        ws = _lib.WORKSPACE.get(chunk * pool_ld * 8, self.device)
        st = _lib.stream_ptr(self.device)
        for s in range(0, nq, chunk):
            m = min(chunk, nq - s)
            if grouped:
                # sort this chunk's (query, probe) pairs by list: integer plumbing on [m * nprobe] elements
                pl = probes[s:s + m].reshape(-1)
                ls = torch.where(pl < 0, torch.full_like(pl, self.nlist), pl)
                order = torch.argsort(ls, stable=True)
                pair_q = (order // nprobe).contiguous()
                pair_p = (order % nprobe).contiguous()
                gcount = torch.bincount(ls, minlength=self.nlist + 1)[:self.nlist]
                goff = torch.zeros(self.nlist + 1, dtype=torch.int64, device=self.device)
                goff[1:] = torch.cumsum(gcount, 0)
                qtp = torch.zeros(self.nlist + 1, dtype=torch.int64, device=self.device)
                qtp[1:] = torch.cumsum((gcount + QTILE - 1) // QTILE, 0)
                bound = (m * nprobe) // QTILE + self.nlist
                _lib.check(lib.amdrec_ivf_scan_grouped(
                    _lib.ptr(xs), xs.stride(0), self.dim, _lib.ptr(spos), _lib.ptr(off), self.nlist, max_len,
                    _lib.ptr(q[s:]), q.stride(0), _lib.ptr(goff), _lib.ptr(qtp), bound, _lib.ptr(pair_q),
                    _lib.ptr(pair_p), _lib.ptr(base[s:]), nprobe, _lib.ptr(ws), pool_ld, pos_offset, st))
            else:
                _lib.check(lib.amdrec_ivf_scan(_lib.ptr(xs), xs.stride(0), self.dim, _lib.ptr(spos), _lib.ptr(off),
                                               _lib.ptr(q[s:]), m, q.stride(0), _lib.ptr(probes[s:]),
                                               _lib.ptr(base[s:]), nprobe, _lib.ptr(ws), pool_ld, pos_offset, st))
            _lib.check(lib.amdrec_ivf_select(_lib.ptr(ws), pool_ld, _lib.ptr(n_pool[s:]), m, k,
                                             _lib.ptr(out_scores[s:]), _lib.ptr(out_pos[s:]), st))

    # -- persistence ----------------------------------------------------------------------
    def export_arrays(self):
        return [("ivf_centroids", self.centroids.cpu().numpy()), ("ivf_assign", self.assign.cpu().numpy())]

    @classmethod
    def from_arrays(cls, arrays, device) -> "IVFState":
        st = cls(torch.from_numpy(np.array(arrays["ivf_centroids"])).to(device))
        st.assign = torch.from_numpy(np.array(arrays["ivf_assign"])).to(device)
        return st
