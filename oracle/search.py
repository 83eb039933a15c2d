"""Oracle for the retrieval index (test infrastructure).

The arithmetic lives in faiss (faiss-cpu>=1.7.4, requirements.txt:8), which is absent
from /root/reference and from this image: PARITY UNPINNED at the faiss boundary.  What is
restated is (i) the reference wrapper's own steps and (ii) the published semantics of the
two index types on the path:

* FAISSIndex.add     faiss_retrieval.py:97-127  fp32 copy -> L2 renorm -> append; ids default arange
* FAISSIndex.search  faiss_retrieval.py:129-166 fp32 copy -> L2 renorm -> index.search -> id remap
                                                 -> returns (ad_ids, distances)
* faiss.normalize_L2: x /= ||x||_2 per row, rows with zero norm left untouched
* IndexFlatIP.search: exact inner product, k largest, sorted descending, (D float32, I int64);
  slots that cannot be filled (k > ntotal) come back as I=-1, D=-inf (upstream behaviour)
* IndexIVFFlat(METRIC_INNER_PRODUCT) with IndexFlatIP quantizer (faiss_retrieval.py:50-55):
  assign each vector to the centroid of maximum inner product; at query time scan the
  ``nprobe`` lists whose centroids have the largest inner product with the query.

Tie rule of this build (documented, deterministic): equal scores -> lower position first.
"""
from __future__ import annotations

import numpy as np


def normalize_l2(x):
    """faiss.normalize_L2 semantics on a copy."""
    x = np.array(x, dtype=np.float32, copy=True)
    n = np.sqrt((x * x).sum(axis=1, keepdims=True, dtype=np.float32))
    np.divide(x, n, out=x, where=n > 0)
    return x


def topk_desc(scores, k):
    """Exact k largest per row, sorted (score desc, position asc). -> (D, I)."""
    nq, n = scores.shape
    kk = min(k, n)
    D = np.full((nq, k), -np.inf, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        s = scores[q]
        if kk < n:
            kth = np.partition(s, n - kk)[n - kk]
            cand = np.nonzero(s >= kth)[0]
        else:
            cand = np.arange(n)
        order = np.lexsort((cand, -s[cand].astype(np.float64)))[:kk]
        I[q, :kk] = cand[order]
        D[q, :kk] = s[cand[order]]
    return D, I


def flat_ip_search(xb, xq, k, chunk=1 << 16, dtype=np.float32):
    """IndexFlatIP.search on already-normalised float32 data -> (D[nq,k], I[nq,k])."""
    xb = np.asarray(xb, dtype=np.float32)
    xq = np.asarray(xq, dtype=np.float32)
    nq, n = xq.shape[0], xb.shape[0]
    if n <= chunk:
        return topk_desc((xq.astype(dtype) @ xb.astype(dtype).T).astype(np.float32), k)
    # chunked: keep a running candidate set, then one exact final selection
    bestD = np.full((nq, 0), -np.inf, dtype=np.float32)
    bestI = np.zeros((nq, 0), dtype=np.int64)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        sc = (xq.astype(dtype) @ xb[s:e].astype(dtype).T).astype(np.float32)
        d, i = topk_desc(sc, min(k, e - s))
        i = np.where(i >= 0, i + s, -1)
        bestD = np.concatenate([bestD, d], axis=1)
        bestI = np.concatenate([bestI, i], axis=1)
        if bestD.shape[1] > 4 * k:
            bestD, bestI = _merge(bestD, bestI, k)
    return _merge(bestD, bestI, k)


def _merge(D, I, k):
    """Merge candidate lists: (score desc, id asc), invalid (-1) last."""
    nq = D.shape[0]
    oD = np.full((nq, k), -np.inf, dtype=np.float32)
    oI = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        valid = I[q] >= 0
        d, i = D[q][valid], I[q][valid]
        order = np.lexsort((i, -d.astype(np.float64)))[:k]
        oD[q, :len(order)] = d[order]
        oI[q, :len(order)] = i[order]
    return oD, oI


def merge_shards(Ds, Is, offsets, k):
    """Cross-shard merge: per-shard (D, local I) + row offsets -> global top-k."""
    D = np.concatenate(Ds, axis=1)
    I = np.concatenate([np.where(i >= 0, i + off, -1) for i, off in zip(Is, offsets)], axis=1)
    return _merge(D, I, k)


def merge_partial(Ds, Is, offsets, k):
    """Merge of SHORT per-shard lists (each [nq, list_k], list_k <= k): -> (D, I, inexact[nq] bool).  A shard's list is its
    best list_k rows in the search's total order (score desc, lower position first), so a row it did not send is strictly
    behind its last entry.  The merged top-k is proven exact when no FULL list's last entry lies strictly AHEAD of the
    merged k-th entry in that order; a last entry that IS the merged k-th, or ties with its score at a higher position,
    proves the list was cut at or below the boundary (score ties alone are not failures).  A merged list shorter than k
    with a full shard list counts as not proven."""
    D, I = merge_shards(Ds, Is, offsets, k)
    nq = D.shape[0]
    inexact = np.zeros(nq, dtype=bool)
    for q in range(nq):
        have_k = I.shape[1] >= k and I[q, k - 1] >= 0
        for d, i, off in zip(Ds, Is, offsets):
            full = i[q, -1] >= 0 and not np.isnan(d[q, -1])
            if not full:
                continue
            ahead = have_k and (d[q, -1] > D[q, k - 1] or (d[q, -1] == D[q, k - 1] and i[q, -1] + off < I[q, k - 1]))
            if not have_k or ahead:
                inexact[q] = True
    return D, I, inexact


class FlatIndex:
    """FAISSIndex(index_type='Flat') restated: add / search with the wrapper's steps."""

    def __init__(self, dimension):
        self.dimension = dimension
        self.xb = np.zeros((0, dimension), dtype=np.float32)
        self.id_map = []

    @property
    def ntotal(self):
        return self.xb.shape[0]

    def add(self, embeddings, ad_ids=None):
        x = normalize_l2(embeddings)                       # :114-115
        if ad_ids is None:                                 # :121-122
            ad_ids = list(range(len(self.id_map), len(self.id_map) + len(x)))
        self.xb = np.concatenate([self.xb, x], axis=0)
        self.id_map.extend(ad_ids)                         # :123

    def search(self, queries, k=100):
        q = normalize_l2(queries)                          # :146-147
        D, I = flat_ip_search(self.xb, q, k)               # :155
        idm = np.asarray(self.id_map, dtype=np.int64)
        ids = idm[I]                                       # :159-160 (I == -1 -> id_map[-1], as upstream)
        return ids, D                                      # :164-166 order (ids, distances)


def kmeans_ip(x, nlist, niter=10, seed=1234):
    """Spherical Lloyd iterations with max-inner-product assignment (the build's own
    trainer; faiss' exact k-means is not reproducible offline - SURVEY.md §8c)."""
    rng = np.random.default_rng(seed)
    x = np.asarray(x, dtype=np.float32)
    cent = x[rng.choice(len(x), nlist, replace=False)].copy()
    for _ in range(niter):
        a = assign_ip(x, cent)
        for c in range(nlist):
            m = x[a == c]
            if len(m):
                cent[c] = m.mean(axis=0)
        cent = normalize_l2(cent)
    return cent


def assign_ip(x, cent, chunk=1 << 16):
    out = np.empty(len(x), dtype=np.int64)
    for s in range(0, len(x), chunk):
        out[s:s + chunk] = np.argmax(x[s:s + chunk] @ cent.T, axis=1)
    return out


def ivf_search(xb, assign, cent, xq, k, nprobe, probes=None):
    """IndexIVFFlat(IP).search given centroids + assignments: exact scan of probed lists.
    ``probes`` [nq, nprobe] overrides the coarse selection (to test the scan in isolation)."""
    nq = xq.shape[0]
    if probes is None:
        coarse = xq @ cent.T
        _, probe = topk_desc(coarse.astype(np.float32), nprobe)
    else:
        probe = np.asarray(probes)
    D = np.full((nq, k), -np.inf, dtype=np.float32)
    I = np.full((nq, k), -1, dtype=np.int64)
    # inverted lists: positions grouped by centroid, insertion order inside a list (one stable sort: 10M rows x 4096
    # lists must not cost 4096 passes over the assignment)
    assign = np.asarray(assign)
    order = np.argsort(assign, kind="stable")
    bounds = np.searchsorted(assign[order], np.arange(cent.shape[0] + 1))
    lists = [order[bounds[c]:bounds[c + 1]] for c in range(cent.shape[0])]
    for q in range(nq):
        rows = np.concatenate([lists[c] for c in probe[q] if c >= 0]) if nprobe else np.zeros(0, int)
        if len(rows) == 0:
            continue
        # float64 dot rounded to fp32: a row's score must not depend on which other rows are scanned with it (a BLAS
        # matvec blocks by matrix shape; the sharded-vs-unsharded tests compare score BITS)
        s = (xb[rows].astype(np.float64) @ xq[q].astype(np.float64)).astype(np.float32)
        order = np.lexsort((rows, -s.astype(np.float64)))[:k]
        D[q, :len(order)] = s[order]
        I[q, :len(order)] = rows[order]
    return D, I


# ---- tolerance-aware comparison (SURVEY.md §8a "Parity rules") ----------------------

def check_topk(D_ref, I_ref, D_got, I_got, tau=1e-5, score_tol=1e-6, scores_of=None):
    """Assert a top-k result equals the oracle's up to near-ties at the k-th score.

    * scores compared by sorted value: |dD| <= score_tol
    * ids must equal the oracle's except elements whose oracle score lies within ``tau``
      of the oracle's k-th score (any member of that near-tie class is acceptable)
    * each returned id's score must be the true score of that id (if ``scores_of`` given:
      callable (q, ids) -> float32 scores)
    """
    D_ref, I_ref = np.asarray(D_ref), np.asarray(I_ref)
    D_got, I_got = np.asarray(D_got), np.asarray(I_got)
    assert D_ref.shape == D_got.shape and I_ref.shape == I_got.shape
    fin = np.isfinite(D_ref)
    assert np.array_equal(fin, np.isfinite(D_got)), "filled-slot pattern differs"
    if fin.any():
        err = np.abs(D_ref[fin] - D_got[fin]).max()
        assert err <= score_tol, f"sorted score mismatch {err}"
    for q in range(D_ref.shape[0]):
        kq = int(fin[q].sum())
        if kq == 0:
            continue
        assert np.all(np.diff(D_got[q, :kq]) <= 0), "scores not sorted descending"
        assert len(set(I_got[q, :kq].tolist())) == kq, "duplicate ids in result"
        kth = D_ref[q, kq - 1]
        ref_set = set(I_ref[q, :kq].tolist())
        got_set = set(I_got[q, :kq].tolist())
        for miss in ref_set - got_set:
            j = int(np.nonzero(I_ref[q, :kq] == miss)[0][0])
            assert D_ref[q, j] - kth <= tau, f"q{q}: id {miss} (score {D_ref[q, j]}) missing"
        for extra in got_set - ref_set:
            j = int(np.nonzero(I_got[q, :kq] == extra)[0][0])
            assert kth - D_got[q, j] <= tau, f"q{q}: id {extra} (score {D_got[q, j]}) spurious"
        if scores_of is not None:
            true = scores_of(q, I_got[q, :kq])
            assert np.abs(true - D_got[q, :kq]).max() <= score_tol, "score does not belong to id"
