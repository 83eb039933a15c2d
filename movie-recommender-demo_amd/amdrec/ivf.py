"""IVF-Flat state for ``FAISSIndex(index_type='IVF')`` (faiss_retrieval.py:50-55): coarse centroids,
list assignment, list-contiguous copy of the corpus, and the search driver over libamdrec's
``amdrec_flat_search`` (coarse probes) + ``amdrec_ivf_group`` + ``amdrec_ivf_scan[_grouped]`` + ``amdrec_ivf_select``.

Build and search are hand-written HIP: assignment = ``amdrec_ivf_assign`` (fp32-MFMA GEMM with an arg-max epilogue),
training = ``amdrec_ivf_kmeans_step`` x NITER (assignment + order-independent fixed-point centroid sums: training is
bit-reproducible), and a search call launches only libamdrec kernels (no ATen op, no host synchronisation: it can be
captured in a HIP graph).  What stays in torch is build-time data plumbing: drawing the training sample and the one
stable sort that lays the corpus out list-contiguously after an ``add``.
faiss' own k-means (its sub-sampling, seeding and iteration details) is not reproducible offline, so, as SURVEY.md
section 8c says, IVF parity is: (i) the scan is exact over the probed lists given this build's centroids and
assignments (tested against the oracle), (ii) recall@k against the Flat result is reported.
Trainer: spherical Lloyd, max-inner-product assignment (the quantizer is IndexFlatIP), <= 256
training points per centroid, 10 iterations, seed 1234 (faiss' documented defaults).
Search = coarse probes -> [nearest eighth of the probes: grouped scan -> select -> tau] -> [other probes: grouped scan keeping
score >= tau] -> select (exact: tau is the k-th score over a subset of the candidates, hence a lower bound of the final one).
Sharding: the centroids are SHARED by all ranks (``amdrec.sharded.share_ivf_centroids``); a rank
files its own rows under them and holds its slice of every list, so the union of the ranks' scans is exactly the
unsharded scan and the merged top-k is bit-identical to the unsharded IVF result.
"""
from __future__ import annotations

import os
from typing import Optional

import numpy as np
import torch

from . import _lib

NITER = 10
MAX_POINTS_PER_CENTROID = 256
SEED = 1234
POOL_BYTES = 2 << 30           # candidate-pool workspace per query chunk
GROUPED_MIN_QUERIES = 16       # batches at least this large MAY scan list-major (every list read once per query tile) ...
GROUPED_MIN_PAIRS_PER_LIST = 3.0   # ... if a probed list is shared by at least this many of the batch's queries on average
QTILE = 64                     # queries per grouped-scan tile (ShapeIvf::BP) ...
QTILE_SPARSE = 32              # ... or 32 (ShapeIvf32) when fewer than SPARSE_PAIRS_PER_LIST queries probe a list on average
SPARSE_PAIRS_PER_LIST = 24
MAX_QUERY_TILES = 65535        # grid limit of one amdrec_ivf_scan_grouped launch (pairs / tile + nlist query tiles)
SELECT_SLICE_KEYS = 4096        # amdrec_ivf_select_split: at least this many pool keys per slice,
SELECT_MAX_SLICES = 64          # at most this many slices per query
MIXED_SCAN = True              # second phase of the two-phase scan on a bf16 shadow of the lists (prefilter) + fp32 re-score of
                               # the nominated rows (csrc/ivf.hip EpiIvfPrefilter); the shadow costs half the lists' bytes again.
                               # AMDREC_IVF_MIXED=0 / 1 in the environment forces it off / on (A/B runs)
MIXED_MIN_FIRST_ROWS_PER_K = 32  # ... only when the first phase scans at least this many rows per wanted result: the prefilter
                               # pays when tau is selective (10M ads / 4096 lists, k = 500: 19.5k first-phase rows, second phase
                               # 0.77 -> 0.43 ms); when it is not, every row that passes costs a random 1 KB fp32 row read (a
                               # rank of 8 over the same index, k = 128 against 2440 first-phase rows of weakly separated
                               # 305-row lists: 0.64 -> 0.60 ms, profiles/r04_ivf_mixed_ab.log) - not worth the shadow's bytes
TWO_PHASE_MIN_PROBES = 16      # from here on the scan is split: nearest probes unfiltered, the rest filtered by their k-th score (at 10 probes it loses: 0.78 vs 0.41 ms at 64 queries, nlist 100)


def _normalize(x):
    return x / x.norm(dim=1, keepdim=True).clamp_min(1e-30)


def _assign(x: torch.Tensor, cent: torch.Tensor) -> torch.Tensor:
    """amdrec_ivf_assign: arg max_c <x, c> per row (ties -> lower centroid)."""
    lib = _lib.load()
    x = x.contiguous()
    n = x.shape[0]
    out = torch.empty(n, dtype=torch.int64, device=x.device)
    if n == 0:
        return out
    ws = _lib.WORKSPACE.get(n * 8 + 256, x.device)
    _lib.check(lib.amdrec_ivf_assign(_lib.ptr(x), n, x.stride(0), x.shape[1], _lib.ptr(cent), cent.shape[0],
                                     cent.stride(0), _lib.ptr(out), None, _lib.ptr(ws), ws.numel(),
                                     _lib.stream_ptr(x.device)))
    return out


def first_phase_probes(nprobe: int) -> int:
    """Probes of the two-phase scan's unfiltered first phase: the nearest eighth (AMDREC_IVF_FIRST_DIV overrides the 8 for
    A/B runs)."""
    return max(2, nprobe // max(1, int(os.environ.get("AMDREC_IVF_FIRST_DIV", "8"))))


def use_mixed_scan(n: int, nlist: int, nprobe: int, k: int) -> bool:
    """Second phase of the two-phase scan on the bf16 shadow of the lists (prefilter + fp32 re-score of the rows that pass)?
    Only when the first phase is selective: it scans at least MIXED_MIN_FIRST_ROWS_PER_K rows per wanted result, so its k-th
    score leaves few rows of the second phase to re-score.  AMDREC_IVF_MIXED=0 / 1 forces the answer (A/B runs)."""
    force = os.environ.get("AMDREC_IVF_MIXED")
    if force in ("0", "1"):
        return force == "1"
    return MIXED_SCAN and first_phase_probes(nprobe) * (n / max(1, nlist)) >= MIXED_MIN_FIRST_ROWS_PER_K * k


def use_grouped_scan(nq: int, nprobe: int, nlist: int) -> bool:
    """List-major (grouped) scan or one workgroup set per (query, probe) pair?  The grouped scan reads a probed list once
    per query tile, the pair scan once per probing query - but at the HBM rate whatever the lists' lengths (their rows are
    split over workgroups), where the grouped kernel's (list, query tile, 256 rows) workgroups run ~3x slower than their
    bytes on short or thinly shared lists (1M ads, nlist 4096, nprobe 64, 64 queries: 1.6 queries per probed list -
    0.60 ms grouped, HBM time of the pairs' bytes 0.2).  So: grouped from GROUPED_MIN_QUERIES queries on AND only when
    the batch's (query, probe) pairs outnumber the lists they can fall on GROUPED_MIN_PAIRS_PER_LIST to one."""
    pairs = nq * nprobe
    return nq >= GROUPED_MIN_QUERIES and pairs >= GROUPED_MIN_PAIRS_PER_LIST * min(nlist, pairs)


def grouped_chunk_limit(nlist: int, nprobe: int) -> int:
    """Largest query chunk whose grouped scan fits one launch: a launch walks at most pairs / tile + nlist query tiles
    (every list can end in a partial tile) and the tile is re-picked per (chunk, probe-column range) - a phase or a tail
    chunk may fall under the sparse threshold - so the limit is taken with the SMALLEST tile.  A quantizer with more
    lists than the grid has tiles cannot take the grouped path at all."""
    room = MAX_QUERY_TILES - nlist
    if room < 1:
        raise ValueError(f"nlist = {nlist} is beyond the grouped IVF scan's limit ({MAX_QUERY_TILES - 1} lists)")
    return max(1, (room * min(QTILE, QTILE_SPARSE)) // max(1, nprobe))


class IVFState:
    def __init__(self, centroids: torch.Tensor):
        self.centroids = centroids.contiguous()                    # [nlist, d], unit rows
        self.nlist, self.dim = centroids.shape
        self.device = centroids.device
        self.assign = torch.empty(0, dtype=torch.int64, device=self.device)   # list of every position
        self._lists = None                                          # (xs, spos, list_off, list_len, max_len)
        self._top_rows = None                                       # host: rows of the p longest lists (pool_rows_bound)
        self._nlist_count = None                                    # device: nlist per query (the coarse select's pool sizes)
        self._sel_scratch = None                                    # device: amdrec_ivf_select_split's partial lists and tickets
        self._sel_tickets = None

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
        cent = x[torch.randperm(n, generator=g)[:nlist].to(x.device)].clone().contiguous()
        x = x.contiguous()
        lib = _lib.load()
        nbytes = _lib.C.c_size_t(0)
        _lib.check(lib.amdrec_ivf_kmeans_workspace(n, x.shape[1], nlist, _lib.C.byref(nbytes)))
        ws = _lib.WORKSPACE.get(nbytes.value, x.device)
        for _ in range(NITER):                                       # Lloyd iterations, all on libamdrec kernels
            _lib.check(lib.amdrec_ivf_kmeans_step(_lib.ptr(x), n, x.stride(0), x.shape[1], _lib.ptr(cent), nlist,
                                                  cent.stride(0), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device)))
        return cls(cent)

    def append(self, x_normalised: torch.Tensor, start: int):
        assert start == self.assign.shape[0]
        self.assign = torch.cat([self.assign, _assign(x_normalised, self.centroids)])
        self._lists = None
        self._top_rows = None

    def _build_lists(self, xb: torch.Tensor, n: int):
        if self._lists is None or self._lists[5] != n:
            a = self.assign[:n]
            order = torch.argsort(a, stable=True)                   # rows of a list keep insertion order
            counts = torch.bincount(a, minlength=self.nlist)
            off = torch.zeros(self.nlist + 1, dtype=torch.int64, device=self.device)
            off[1:] = torch.cumsum(counts, 0)
            xs = xb[:n][order].contiguous()
            self._lists = (xs, order.contiguous(), off, counts.to(torch.int64), int(counts.max().item()) if n else 0, n)
            self._shadow = None                                     # bf16 copy of xs + its {M, D}: made on first use
            # rows of the p longest lists, p = 1 .. nlist: the tight host-side bound of a query's candidate pool
            # (its nprobe probed lists cannot hold more than the nprobe longest; once per list rebuild, like max above)
            self._top_rows = np.cumsum(np.sort(counts.cpu().numpy())[::-1].astype(np.int64))
        return self._lists

    def _list_shadow(self):
        """bf16 (round-to-nearest) copy of the list-contiguous corpus and max_norm = {largest row norm, largest row
        rounding-error norm} (amdrec_bf16_rows): the operands of the second-phase prefilter."""
        if getattr(self, "_shadow", None) is None:
            xs = self._lists[0]
            n, d = xs.shape
            xs16 = torch.empty((n, d), dtype=torch.bfloat16, device=self.device)
            mx = torch.zeros(2, dtype=torch.float32, device=self.device)
            if n:
                _lib.check(_lib.load().amdrec_bf16_rows(_lib.ptr(xs), n, xs.stride(0), d, _lib.ptr(xs16), d, _lib.ptr(mx),
                                                        _lib.stream_ptr(self.device)))
            self._shadow = (xs16, mx)
        return self._shadow

    def pool_rows_bound(self, nprobe: int) -> int:
        """Upper bound of the rows a query's ``nprobe`` probed lists hold = the ``nprobe`` longest lists' rows.  Round 2
        sized the pool as nprobe x the longest list (9767 x 64 keys = 5 MB per query at 10M / 4096: 512 queries needed two
        chunks of the 2 GB pool); list lengths spread 0 .. 4x the mean, so this is ~1.6x tighter."""
        if self._top_rows is None or len(self._top_rows) == 0:
            return 1
        return max(1, int(self._top_rows[min(int(nprobe), len(self._top_rows)) - 1]))

    def _select(self, lib, pool, pool_ld: int, n_pool, m: int, k: int, out_scores, out_pos, stream_ptr):
        """The k best keys of each query's pool.  Few queries with large pools (one request against nlist 100 / nprobe 10:
        100 000 keys) are selected by several workgroups per query (``amdrec_ivf_select_split``), everything else by one."""
        slices = min(SELECT_MAX_SLICES, pool_ld // SELECT_SLICE_KEYS, 1024 // max(1, m))
        if slices < 4:
            _lib.check(lib.amdrec_ivf_select(_lib.ptr(pool), pool_ld, _lib.ptr(n_pool), m, k, _lib.ptr(out_scores),
                                             _lib.ptr(out_pos), stream_ptr))
            return
        need = m * slices * k * 8
        if self._sel_scratch is None or self._sel_scratch.numel() < need:
            self._sel_scratch = torch.empty(need, dtype=torch.uint8, device=self.device)
        if self._sel_tickets is None or self._sel_tickets.numel() < m:
            self._sel_tickets = torch.zeros(max(m, 1024), dtype=torch.int32, device=self.device)   # the kernel leaves them zero
        _lib.check(lib.amdrec_ivf_select_split(_lib.ptr(pool), pool_ld, _lib.ptr(n_pool), m, k, slices, _lib.ptr(out_scores),
                                               _lib.ptr(out_pos), _lib.ptr(self._sel_scratch), self._sel_scratch.numel(),
                                               _lib.ptr(self._sel_tickets), stream_ptr))

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
        # 1. coarse quantizer: nprobe best centroids by inner product (IndexFlatIP quantizer) = a dense (score, centroid)
        #    key table + the pool select (step 4's kernel); runs below, once the workspace is sized
        cs = torch.empty((nq, nprobe), dtype=torch.float32, device=self.device)
        probes = torch.empty((nq, nprobe), dtype=torch.int64, device=self.device)
        # 2. pool layout + (query, probe) pairs grouped by list: amdrec_ivf_group (kernels, no host sync)
        base = torch.empty((nq, nprobe), dtype=torch.int64, device=self.device)
        n_pool = torch.empty((nq,), dtype=torch.int64, device=self.device)
        pool_ld = self.pool_rows_bound(nprobe)
        chunk = max(1, min(nq, 65535, POOL_BYTES // (pool_ld * 8)))
        grouped = use_grouped_scan(nq, nprobe, self.nlist)
        qtile = QTILE_SPARSE if min(chunk, nq) * nprobe < SPARSE_PAIRS_PER_LIST * self.nlist else QTILE
        if grouped:
            chunk = min(chunk, grouped_chunk_limit(self.nlist, nprobe))
        # one workspace: [candidate pool | grouping scratch | pair arrays and offsets]
        pool_bytes = (chunk * pool_ld * 8 + 255) // 256 * 256
        grp_bytes = ((self.nlist + 1) * 4 + 255) // 256 * 256 + (chunk * nprobe * 4 + 255) // 256 * 256
        arr_bytes = 2 * chunk * nprobe * 8 + 2 * (self.nlist + 1) * 8
        coarse_ld = (self.nlist + 1) // 2 * 2
        coarse = nq * coarse_ld * 8 <= POOL_BYTES and nprobe <= _lib.MAX_K and nq < (1 << 24)
        wsall = _lib.WORKSPACE.get(max(pool_bytes + grp_bytes + arr_bytes + 256, nq * coarse_ld * 8 if coarse else 0),
                                   self.device)
        if coarse:                                     # the key table lives in the pool's memory: the scans come after it
            if self._nlist_count is None or self._nlist_count.numel() < nq:
                self._nlist_count = torch.full((max(nq, 512),), self.nlist, dtype=torch.int64, device=self.device)
            _lib.check(lib.amdrec_ivf_coarse_keys(_lib.ptr(self.centroids), self.nlist, self.centroids.stride(0), self.dim,
                                                  _lib.ptr(q), nq, q.stride(0), _lib.ptr(wsall), coarse_ld,
                                                  _lib.stream_ptr(self.device)))
            _lib.check(lib.amdrec_ivf_select(_lib.ptr(wsall), coarse_ld, _lib.ptr(self._nlist_count), nq, nprobe,
                                             _lib.ptr(cs), _lib.ptr(probes), _lib.stream_ptr(self.device)))
        else:
            flat_search(self.centroids, self.nlist, q, nprobe, cs, probes)
        ws = wsall[:pool_bytes]
        grp = wsall[pool_bytes:pool_bytes + grp_bytes]
        arr = wsall[pool_bytes + grp_bytes:pool_bytes + grp_bytes + arr_bytes].view(torch.int64)
        pair_q, pair_p = arr[:chunk * nprobe], arr[chunk * nprobe:2 * chunk * nprobe]
        goff = arr[2 * chunk * nprobe:2 * chunk * nprobe + self.nlist + 1]
        qtp = arr[2 * chunk * nprobe + self.nlist + 1:]
        st = lambda: _lib.stream_ptr(self.device)      # noqa: E731  (per call: check() ends the call's device scope)

        def group_and_scan(s, m, col0, ncol, tau=None, fill=None, n_out=None):
            """(query, probe) pairs of probe columns [col0, col0 + ncol) grouped by list, then the grouped scan of those
            lists: unfiltered into the per-probe slots of the pool, or (tau, fill) filtered and appended."""
            qt = QTILE_SPARSE if m * ncol < SPARSE_PAIRS_PER_LIST * self.nlist else QTILE
            if os.environ.get("AMDREC_IVF_QTILE"):                       # A/B runs: "32", "64", or "first,second" per phase
                f = os.environ["AMDREC_IVF_QTILE"].split(",")
                qt = int(f[0] if (tau is None or len(f) == 1) else f[1])
            pv = probes[s:, col0:]
            _lib.check(lib.amdrec_ivf_group(_lib.ptr(pv), nprobe, m, ncol, self.nlist, _lib.ptr(lens), _lib.ptr(base[s:]),
                                            _lib.ptr(n_out), _lib.ptr(pair_q), _lib.ptr(pair_p), _lib.ptr(goff),
                                            _lib.ptr(qtp), qt, _lib.ptr(grp), grp.numel(), st()))
            if tau is not None and mixed:
                # bf16 prefilter: tau_lo = tau - eps_q, nominate on the bf16 shadow, re-score the nominated rows in fp32
                _lib.check(lib.amdrec_ivf_filter_bounds(_lib.ptr(q[s:]), m, q.stride(0), self.dim, _lib.ptr(q16[s:]), q16.stride(0),
                                                        _lib.ptr(mx16), _lib.ptr(tau), tau.stride(0), _lib.ptr(tau_lo[s:]), st()))
                _lib.check(lib.amdrec_ivf_scan_grouped_mixed(
                    _lib.ptr(xs), xs.stride(0), _lib.ptr(xs16), xs16.stride(0), self.dim, _lib.ptr(spos), _lib.ptr(off),
                    self.nlist, max_len, _lib.ptr(q[s:]), q.stride(0), _lib.ptr(q16[s:]), q16.stride(0), _lib.ptr(goff),
                    _lib.ptr(qtp), (m * ncol) // qt + self.nlist, qt, _lib.ptr(pair_q), _lib.ptr(ws), pool_ld, pos_offset,
                    _lib.ptr(tau), tau.stride(0), _lib.ptr(tau_lo[s:]), _lib.ptr(fill), st()))
                return
            _lib.check(lib.amdrec_ivf_scan_grouped(
                _lib.ptr(xs), xs.stride(0), self.dim, _lib.ptr(spos), _lib.ptr(off), self.nlist, max_len,
                _lib.ptr(q[s:]), q.stride(0), _lib.ptr(goff), _lib.ptr(qtp), (m * ncol) // qt + self.nlist, qt,
                _lib.ptr(pair_q), _lib.ptr(pair_p), _lib.ptr(base[s:]), ncol, _lib.ptr(ws), pool_ld, pos_offset,
                _lib.ptr(tau), 0 if tau is None else tau.stride(0), _lib.ptr(fill), st()))

        # Exact two-phase scan (batches that take the grouped scan, >= TWO_PHASE_MIN_PROBES probes): the nearest eighth of
        # the probes unfiltered -> select -> tau = that subset's k-th score, a LOWER bound of the final k-th score -> the other
        # probes keep only rows with score >= tau, appended behind the first phase's keys -> the final select.  Same
        # result as one unfiltered scan of every probe; the pool that is written and selected from shrinks several-fold.
        two_phase = grouped and nprobe >= TWO_PHASE_MIN_PROBES
        n_first = first_phase_probes(nprobe)
        mixed = two_phase and self.dim % 8 == 0 and use_mixed_scan(n, self.nlist, nprobe, k)
        if mixed:
            xs16, mx16 = self._list_shadow()
            q16 = torch.empty((nq, self.dim), dtype=torch.bfloat16, device=self.device)
            _lib.check(lib.amdrec_bf16_rows(_lib.ptr(q), nq, q.stride(0), self.dim, _lib.ptr(q16), q16.stride(0), None, st()))
            tau_lo = torch.empty((nq,), dtype=torch.float32, device=self.device)
        scratch_n = torch.empty((chunk,), dtype=torch.int64, device=self.device) if two_phase else None
        for s in range(0, nq, chunk):
            m = min(chunk, nq - s)
            if two_phase:
                group_and_scan(s, m, 0, n_first, n_out=n_pool[s:])
                self._select(lib, ws, pool_ld, n_pool[s:], m, k, out_scores[s:], out_pos[s:], st())
                group_and_scan(s, m, n_first, nprobe - n_first, tau=out_scores[s:, k - 1], fill=n_pool[s:], n_out=scratch_n)
            elif grouped:
                group_and_scan(s, m, 0, nprobe, n_out=n_pool[s:])
            else:
                _lib.check(lib.amdrec_ivf_group(_lib.ptr(probes[s:]), nprobe, m, nprobe, self.nlist, _lib.ptr(lens),
                                                _lib.ptr(base[s:]), _lib.ptr(n_pool[s:]), _lib.ptr(pair_q), _lib.ptr(pair_p),
                                                _lib.ptr(goff), _lib.ptr(qtp), qtile, _lib.ptr(grp), grp.numel(), st()))
                _lib.check(lib.amdrec_ivf_scan(_lib.ptr(xs), xs.stride(0), self.dim, _lib.ptr(spos), _lib.ptr(off),
                                               _lib.ptr(q[s:]), m, q.stride(0), _lib.ptr(probes[s:]),
                                               _lib.ptr(base[s:]), nprobe, _lib.ptr(ws), pool_ld, pos_offset, st()))
            self._select(lib, ws, pool_ld, n_pool[s:], m, k, out_scores[s:], out_pos[s:], st())

    # -- persistence ----------------------------------------------------------------------
    def export_arrays(self):
        return [("ivf_centroids", self.centroids.cpu().numpy()), ("ivf_assign", self.assign.cpu().numpy())]

    @classmethod
    def from_arrays(cls, arrays, device) -> "IVFState":
        st = cls(torch.from_numpy(np.array(arrays["ivf_centroids"])).to(device))
        st.assign = torch.from_numpy(np.array(arrays["ivf_assign"])).to(device)
        return st
