"""Corpus-sharded serving across the GPUs of one node (absent in the reference,
which is single-process / device 0 only, faiss_retrieval.py:76-78).

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).  Rank r holds corpus
rows [offset_r, offset_r + n_r); weights and the ad-feature table are replicated.  One step:

  1. every rank encodes the GLOBAL user batch (the tower is 0.26 GFLOP per 512 users - cheaper
     than broadcasting embeddings) and searches ITS shard for all users -> [B_g, k] scores and
     global positions;
  2. ONE collective on a packed buffer [scores f32 | global positions i32] (8 B per candidate): an all-to-all in
     which every rank sends each peer only the lists of that peer's users (B*k*8 bytes in and out per rank whatever
     the world size: 2 MB per 512 users x 500), or - when the batch does not divide - an all-gather of everything
     (world x as much); the payload is latency-bound on xGMI, so no ring-sized bucketing is needed;
  3. rank r merges the world's lists for ITS contiguous slice of users (G*k -> k, exact, same
     order rule as the single-GPU search: the result is bit-identical to an unsharded search);
  4. rank r ranks its own users' candidates (data parallel) and selects their top-k.

Short lists (``shard_k``): a randomly sharded corpus puts k/G +- sqrt(k/G) of a user's top-k rows on each shard, so a
shard need not produce (re-score exactly, sort, send) its own top-k: it sends its best ``short_list_k(k, G)`` rows
(128 instead of 500 at G = 8) and the merge PROVES the result exact - no shard's list ends strictly ahead of the merged
k-th entry in the search's (score, position) order (``amdrec_topk_merge_partial``; a tie in score alone is not a failure:
duplicate ads are routine) - or counts the query as inexact, in which case THAT batch is repeated with full lists.  Short
lists are switched off for the recommender only when the failures are not occasional (``SHORT_LIST_MAX_FAIL_FRAC`` of a
batch's queries, or ``SHORT_LIST_MAX_REPEAT_FRAC`` of the recent batches: a corpus sharded by topic defeats the premise);
``short_list_stats()`` reports the hit rate.

The compute steps go through an ``engine`` so that the orchestration (slicing, packing, the
collective, offsets) can be exercised on CPU under gloo with a test engine; the default
engine is the HIP one and refuses CPU tensors.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.distributed as dist

from . import _lib


def user_slice(n_users: int, rank: int, world: int):
    """Contiguous block of users owned by ``rank`` (ceil division, last ranks may be short/empty)."""
    per = (n_users + world - 1) // world
    q0 = min(rank * per, n_users)
    return q0, min(per, n_users - q0)


def short_list_k(k: int, world: int) -> int:
    """Per-shard list length for a randomly sharded corpus: mean k/G + 6 sigma of Binomial(k, 1/G) + 8, rounded up to
    a multiple of 32; k itself when that saves less than a fifth (small worlds, small k)."""
    if world <= 1:
        return k
    m = k / world
    kq = int(-(-(m + 6.0 * (m * (1.0 - 1.0 / world)) ** 0.5 + 8.0) // 32) * 32)
    return k if kq > 0.8 * k else kq


def packed_layout(n_users: int, k: int):
    """Byte layout of one rank's all-gather chunk: [scores f32 [B,k] | global positions i32 [B,k]]
    (corpus positions are < 2^31: amdrec_flat_search refuses larger shards).  -> (offset of positions, bytes)"""
    s_bytes = n_users * k * 4
    return s_bytes, 2 * s_bytes


STAGES = ("tower", "search", "pack", "exchange", "merge", "proof_wait", "ranker", "select")


class StageTimer:
    """Per-stage times of ShardedRecommender steps, so that a multi-GPU run localises its own problems (the first
    8-GPU run is also the first debug run).  Device stages are bracketed by HIP events on the launch stream - a
    collective issued through torch.distributed makes that stream wait for it, so the event behind it is the
    collective's completion -, CPU engines (the gloo tests) by host clocks; ``proof_wait`` is host wall time (the
    host blocks there while the ranker runs).  An event pair idles the stream ~10 us, so a timer is attached for a
    few diagnostic steps, never inside a timed region.  ``report()`` -> mean ms per step and stage."""

    def __init__(self, device):
        self.cuda = torch.device(device).type == "cuda"
        self.marks, self.host, self.steps = [], {}, 0

    def mark(self, name):
        if self.cuda:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.marks.append((name, ev))
        else:
            import time
            self.marks.append((name, time.perf_counter()))
        if name == "start":
            self.steps += 1

    def add_host(self, name, seconds):
        self.host[name] = self.host.get(name, 0.0) + seconds

    def report(self):
        if self.cuda:
            torch.cuda.synchronize()
        tot = {}
        for (_, a), (nb, b) in zip(self.marks, self.marks[1:]):
            if nb == "start":
                continue
            tot[nb] = tot.get(nb, 0.0) + (a.elapsed_time(b) if self.cuda else (b - a) * 1e3)
        for k, v in self.host.items():
            tot[k] = tot.get(k, 0.0) + v * 1e3
        n = max(self.steps, 1)
        return {k: round(tot.get(k, 0.0) / n, 4) for k in STAGES}


class HipEngine:
    """The product engine: libamdrec kernels behind an AdRecommenderInference."""

    def __init__(self, rec, shard_offset: int):
        self.rec = rec
        self.shard_offset = int(shard_offset)

    def local_search(self, uc, un, k, mark=None):
        emb = self.rec.two_tower_model.user_tower.encode(uc, un, check_indices=False, renormalize=True)
        if mark is not None:
            mark("tower")
        pos, scores = self.rec.faiss_index.search_device(emb, k, normalize=False, return_positions=True,
                                                         pos_offset=self.shard_offset)
        return scores, pos

    def merge(self, gathered: torch.Tensor, world: int, n_users: int, k: int, q0: int, nq: int, k_out: Optional[int] = None,
              inexact: Optional[torch.Tensor] = None):
        """``world`` lists of ``k`` entries per user -> top ``k_out`` (default k).  With k < k_out (short lists) ``inexact``
        (device int32[1]) is incremented per query whose result could not be proven exact."""
        lib = _lib.load()
        k_out = k if k_out is None else k_out
        s_bytes, chunk = packed_layout(n_users, k)
        dev = gathered.device
        out_s = torch.empty((nq, k_out), dtype=torch.float32, device=dev)
        out_p = torch.empty((nq, k_out), dtype=torch.int64, device=dev)
        base = gathered.data_ptr()
        if k_out == k:
            _lib.check(lib.amdrec_topk_merge(_lib.C.c_void_p(base), _lib.C.c_void_p(base + s_bytes), world, chunk,
                                             q0, nq, k, _lib.ptr(out_s), _lib.ptr(out_p), _lib.stream_ptr(dev)))
        else:
            _lib.check(lib.amdrec_topk_merge_partial(_lib.C.c_void_p(base), _lib.C.c_void_p(base + s_bytes), world, k, chunk,
                                                     q0, nq, k_out, _lib.ptr(out_s), _lib.ptr(out_p), _lib.ptr(inexact),
                                                     _lib.stream_ptr(dev)))
        return out_s, out_p

    def rank(self, uc, un, cand_pos, top_k, mark=None):
        return self.rec._stage2(uc, un, cand_pos, top_k, False, ids_are_positions=True, mark=mark)


def share_ivf_centroids(index, train_rows, rank: int, world: int, group=None, src: int = 0):
    """One coarse quantizer for all ranks of a row-sharded IVF index: rank ``src`` trains k-means
    on ``train_rows`` (its own shard's rows are an unbiased sample of a randomly sharded corpus) and broadcasts the
    [nlist, d] centroids (4 MB at 4096 x 256: one small collective at BUILD time); every rank installs them and then
    ``add``s its rows, which files them under the shared centroids - each rank ends up with its slice of EVERY list.
    At query time all ranks compute identical probes (same centroids, same queries), scan their slices, and the usual
    top-k exchange + merge returns exactly the unsharded IVF result."""
    dev = index.device
    err = None
    if rank == src:
        try:
            index.train(train_rows)
            cent = index.centroids.clone()
        except Exception as e:                  # the peers are about to wait in a collective: tell them first
            err = e
            cent = torch.empty((index.nlist, index.dimension), dtype=torch.float32, device=dev)
    else:
        cent = torch.empty((index.nlist, index.dimension), dtype=torch.float32, device=dev)
    if world > 1:
        gloo = dist.get_backend(group) == "gloo"
        ok = torch.tensor([0 if err is not None else 1], dtype=torch.int32, device="cpu" if gloo else dev)
        dist.broadcast(ok, src=src, group=group)                    # status first: every rank raises together
        if int(ok.item()) != 1:
            if err is not None:
                raise err
            raise RuntimeError(f"share_ivf_centroids: training failed on rank {src}")
        if cent.is_cuda and gloo:
            h = cent.cpu()
            dist.broadcast(h, src=src, group=group)
            cent.copy_(h)
        else:
            dist.broadcast(cent, src=src, group=group)
    elif err is not None:
        raise err
    if rank != src:
        index.set_trained_centroids(cent)
    return cent


def all_gather_bytes(out: torch.Tensor, inp: torch.Tensor, group=None):
    """all_gather_into_tensor; device tensors under a gloo group (rehearsal of the multi-rank
    path on a box with fewer GPUs than ranks) are staged through the host."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        h_in, h_out = inp.cpu(), torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h_out, h_in, group=group)
        out.copy_(h_out)
    else:
        dist.all_gather_into_tensor(out, inp, group=group)


def all_to_all_bytes(out: torch.Tensor, inp: torch.Tensor, group=None):
    """all_to_all_single with equal splits; device tensors under a gloo group are staged through the host."""
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        h_in, h_out = inp.cpu(), torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(h_out, h_in, group=group)
        out.copy_(h_out)
    else:
        dist.all_to_all_single(out, inp, group=group)


SHORT_LIST_MAX_FAIL_FRAC = 0.05      # one batch with more than this share of unproven queries: the premise is false
SHORT_LIST_MAX_REPEAT_FRAC = 0.25    # ... or more than this share of the last SHORT_LIST_WINDOW batches repeated
SHORT_LIST_WINDOW = 16


class ShardedRecommender:
    def __init__(self, rec, rank: int, world: int, shard_offset: int, group: Optional[dist.ProcessGroup] = None,
                 engine=None, exchange: str = "auto", shard_k="auto"):
        """``exchange``: "all_to_all" - every rank sends each peer only the candidate lists of THAT peer's users
        (B*k*8 bytes in and out per rank, independent of the world size; needs n_users % world == 0);
        "all_gather" - every rank receives every list (world x as much); "auto" picks all_to_all when it applies.
        ``shard_k``: entries per shard list - "auto" = ``short_list_k(stage1_k, world)``, an int, or None = stage1_k
        (full lists, no proof needed)."""
        if exchange not in ("auto", "all_to_all", "all_gather"):
            raise ValueError("exchange must be 'auto', 'all_to_all' or 'all_gather'")
        if not (shard_k is None or shard_k == "auto" or (isinstance(shard_k, int) and shard_k >= 1)):
            raise ValueError("shard_k must be 'auto', None or a positive int")
        self.rank, self.world, self.group, self.exchange = rank, world, group, exchange
        self.shard_k = shard_k
        self.engine = engine if engine is not None else HipEngine(rec, shard_offset)
        self._inexact = None                 # device int32[1]: queries of THIS rank's users not proven exact, accumulated
        self._recent = []                    # 1 per verified short-list batch that had to be repeated, else 0 (window)
        self.stats = {"batches": 0, "queries": 0, "unproven_queries": 0, "repeated_batches": 0, "switched_off": False}
        self.last_exchange = None            # {"kind", "bytes_per_rank", "list_k"} of the most recent step
        self._side = None                    # side stream + pinned flag of the asynchronous proof read (_proof_begin)
        self._host_flag = None
        self.timer = None                    # a StageTimer while a caller wants per-stage times (bench.py's N > 1 line)
        import inspect
        self._engine_marks = ("mark" in inspect.signature(self.engine.local_search).parameters
                              and "mark" in inspect.signature(self.engine.rank).parameters)

    def _mark(self, name):
        if self.timer is not None:
            self.timer.mark(name)

    def list_k(self, stage1_k: int) -> int:
        if self.shard_k is None or self.world <= 1:
            return stage1_k
        kq = short_list_k(stage1_k, self.world) if self.shard_k == "auto" else min(int(self.shard_k), stage1_k)
        return stage1_k if kq * self.world < stage1_k else kq

    def inexact_count(self, reset: bool = True) -> int:
        """Queries (over ALL ranks) whose short-list merge was not proven exact since the last reset: one small
        all-reduce + a host read.  Collective: every rank must call it."""
        if self._inexact is None:
            return 0
        t = self._inexact.clone()
        if self.world > 1:
            if t.is_cuda and dist.get_backend(self.group) == "gloo":
                h = t.cpu()
                dist.all_reduce(h, group=self.group)
                t = h
            else:
                dist.all_reduce(t, group=self.group)
        if reset:
            self._inexact.zero_()
        return int(t.item())

    @torch.no_grad()
    def recommend_device(self, user_categorical, user_numerical, top_k: int = 10, stage1_k: int = 500,
                         verify: bool = True):
        """Global user batch (identical on every rank) -> this rank's users' results:
        dict(ad_ids [nq,top_k], scores [3,nq,top_k], user_offset q0, candidate_ids, candidate_scores).
        With short lists and ``verify`` (default) the proof of exactness is checked on the host before returning (one tiny
        all-reduce + a device sync) and a failed batch is recomputed with full lists; ``verify=False`` leaves the check
        to the caller (``inexact_count()``, e.g. once per reporting interval)."""
        kq = self.list_k(stage1_k)
        if not (kq < stage1_k and verify):
            return self._step(user_categorical, user_numerical, top_k, stage1_k, kq)
        # The proof is read BETWEEN the merge and the ranker launch, on a side stream: the host blocks only until the 4-byte
        # all-reduce and its copy have landed while the ranker (2.7 ms) runs, so the next step's launches queue up behind it
        # as in the unverified mode.  (Reading the counter after the step - one .item() on the launch stream - parks the
        # host until the ranker is done and exposes the next step's ~0.3 ms of launch work: 6 % in the 2-rank rehearsal.)
        proof = []
        out = self._step(user_categorical, user_numerical, top_k, stage1_k, kq, after_merge=lambda: proof.append(self._proof_begin()))
        import time
        t0 = time.perf_counter()
        bad = self._proof_end(proof[0])                                       # identical on every rank (all-reduce)
        if self.timer is not None:
            self.timer.add_host("proof_wait", time.perf_counter() - t0)
        B = int(user_categorical.shape[0])
        st = self.stats
        st["batches"] += 1
        st["queries"] += B
        st["unproven_queries"] += bad
        self._recent = (self._recent + [1 if bad else 0])[-SHORT_LIST_WINDOW:]
        if bad:
            # per-batch fallback: only this batch pays the second exchange.  What it costs: the short-list step's ranker
            # pass (already enqueued when the proof is read: ~2.7 ms of GPU time per 512 users) is discarded and the whole
            # step, local search included, runs again with full lists - acceptable for the rare unlucky batch (6 sigma),
            # which is why a corpus where it is NOT rare loses its short lists below.
            st["repeated_batches"] += 1
            out = self._step(user_categorical, user_numerical, top_k, stage1_k, stage1_k)
        # The decision to give short lists up for good needs more than an occasional unlucky query; it is evaluated on
        # EVERY verified batch (every rank sees the same counts: same decision), so the statistics and the decision stay
        # consistent: a streak of repeats at or below the per-batch threshold switches them off as soon as the window says so.
        often = len(self._recent) >= 4 and sum(self._recent) > SHORT_LIST_MAX_REPEAT_FRAC * len(self._recent)
        if bad > SHORT_LIST_MAX_FAIL_FRAC * B or often:
            self.shard_k = None                                           # this corpus is not randomly sharded
            st["switched_off"] = True
        return out

    def _proof_begin(self):
        """Start reading the proof counter (queries of this step not proven exact, summed over the ranks) right behind the
        merge: clone + reset on the launch stream, all-reduce, copy to pinned host memory on a side stream.  -> handle for
        ``_proof_end``.  Under a gloo group with device tensors (rehearsal) and for CPU engines the read is synchronous."""
        if self._inexact is None:
            return ("value", 0)
        t = self._inexact.clone()
        self._inexact.zero_()
        if not t.is_cuda or self.world <= 1 or dist.get_backend(self.group) == "gloo" or os.environ.get("AMDREC_SYNC_PROOF") == "1":
            return ("sync", t)
        work = dist.all_reduce(t, group=self.group, async_op=True)            # waits for the merge, not for what follows
        if self._side is None:
            self._side = torch.cuda.Stream(device=t.device)
            self._host_flag = torch.empty(1, dtype=torch.int32, pin_memory=True)
        with torch.cuda.stream(self._side):
            work.wait()                                                       # the SIDE stream waits for the collective
            t.record_stream(self._side)
            self._host_flag.copy_(t, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self._side)
        return ("async", ev, t)

    def _proof_end(self, handle) -> int:
        if handle[0] == "value":
            return handle[1]
        if handle[0] == "async":
            handle[1].synchronize()                                           # all-reduce + 4-byte copy; the ranker keeps running
            return int(self._host_flag.item())
        t = handle[1]
        if self.world > 1:
            if t.is_cuda and dist.get_backend(self.group) == "gloo":
                h = t.cpu()
                dist.all_reduce(h, group=self.group)
                t = h
            else:
                dist.all_reduce(t, group=self.group)
        return int(t.item())

    def short_list_stats(self):
        """Verified short-list batches so far: queries proven exact on the first exchange / all queries (hit rate),
        batches repeated with full lists, and whether short lists were switched off."""
        st = dict(self.stats)
        st["hit_rate"] = 1.0 - st["unproven_queries"] / st["queries"] if st["queries"] else None
        return st

    def _step(self, uc, un, top_k: int, k_out: int, k: int, after_merge=None):
        """One exchange with lists of k entries per shard, merged to k_out candidates per user.  ``after_merge``: called once
        the merge is enqueued, before the ranker is (the proof read of the verified mode)."""
        B = uc.shape[0]
        self._mark("start")
        if self._engine_marks:
            scores, pos = self.engine.local_search(uc, un, k, mark=self._mark)    # [B,k] each; marks "tower" inside
        else:
            scores, pos = self.engine.local_search(uc, un, k)
        self._mark("search")
        q0, nq = user_slice(B, self.rank, self.world)
        inexact = None
        if k < k_out:
            if self._inexact is None:
                self._inexact = torch.zeros(1, dtype=torch.int32, device=scores.device)
            inexact = self._inexact
        if self.exchange == "all_to_all" and B % self.world:
            raise ValueError("exchange='all_to_all' needs n_users divisible by the world size")
        if self.exchange != "all_gather" and B % self.world == 0 and self.world > 1:
            # destination d receives [scores of d's users | positions of d's users] from every rank
            s_bytes, chunk = packed_layout(nq, k)
            send = torch.empty(chunk * self.world, dtype=torch.uint8, device=scores.device)
            sv = send.view(self.world, chunk)
            sv[:, :s_bytes].view(torch.float32).copy_(scores.reshape(self.world, nq * k))
            sv[:, s_bytes:].view(torch.int32).copy_(pos.reshape(self.world, nq * k))    # int64 -> int32 on the wire
            recv = torch.empty_like(send)
            self._mark("pack")
            all_to_all_bytes(recv, send, self.group)                          # ONE collective per step
            self._mark("exchange")
            self.last_exchange = {"kind": "all_to_all", "bytes_per_rank": int(send.numel()), "list_k": k}
            cand_scores, cand_pos = self._merge(recv, nq, k, 0, nq, k_out, inexact)
        else:
            s_bytes, chunk = packed_layout(B, k)
            buf = torch.empty(chunk, dtype=torch.uint8, device=scores.device)
            buf[:B * k * 4].view(torch.float32).copy_(scores.reshape(-1))
            buf[s_bytes:].view(torch.int32).copy_(pos.reshape(-1))        # int64 -> int32 on the wire
            gathered = torch.empty(chunk * self.world, dtype=torch.uint8, device=scores.device)
            self._mark("pack")
            all_gather_bytes(gathered, buf, self.group)                       # ONE collective per step
            self._mark("exchange")
            self.last_exchange = {"kind": "all_gather", "bytes_per_rank": int(buf.numel()), "list_k": k}
            cand_scores, cand_pos = self._merge(gathered, B, k, q0, nq, k_out, inexact)
        self._mark("merge")
        if after_merge is not None:
            after_merge()
        if self._engine_marks:
            out = self.engine.rank(uc[q0:q0 + nq], un[q0:q0 + nq], cand_pos, top_k, mark=self._mark)   # marks "ranker" inside
        else:
            out = self.engine.rank(uc[q0:q0 + nq], un[q0:q0 + nq], cand_pos, top_k)
        self._mark("select")
        out["candidate_scores"] = cand_scores
        out["user_offset"] = q0
        return out

    def _merge(self, buf, n_users, k, q0, nq, k_out, inexact):
        if k == k_out:
            return self.engine.merge(buf, self.world, n_users, k, q0, nq)
        return self.engine.merge(buf, self.world, n_users, k, q0, nq, k_out, inexact)

    @torch.no_grad()
    def recommend_all(self, user_categorical, user_numerical, top_k: int = 10, stage1_k: int = 500):
        """Same, then all-gather the final [nq, top_k] results so every rank holds all users
        (~60 KB at 512 x 10).  Requires n_users % world == 0."""
        out = self.recommend_device(user_categorical, user_numerical, top_k, stage1_k)
        B = user_categorical.shape[0]
        if B % self.world:
            raise ValueError("recommend_all needs n_users divisible by the world size")
        ids = torch.empty((B, top_k), dtype=torch.int64, device=out["ad_ids"].device)
        all_gather_bytes(ids, out["ad_ids"].contiguous(), self.group)
        sc = out["scores"]                                                    # [T, nq, top_k]
        T = sc.shape[0]
        sc_all = torch.empty((self.world, T, B // self.world, top_k), dtype=sc.dtype, device=sc.device)
        all_gather_bytes(sc_all, sc.contiguous(), self.group)
        return {"ad_ids": ids, "scores": sc_all.permute(1, 0, 2, 3).reshape(T, B, top_k), "tasks": out["tasks"]}
