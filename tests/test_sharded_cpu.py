"""world_size-2 gloo test of the sharded serving orchestration (amdrec.sharded) on CPU.

The compute engine is injected: here it is the CPU oracle (test infrastructure), so what is
under test is the product's slicing / packing / all-gather / offset logic - the code path the
8-GPU RCCL run takes - not the kernels (those are covered by the -m gpu tests)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from amdrec import synth
from amdrec.sharded import ShardedRecommender, packed_layout, user_slice
from tests import cases

N_ADS, B, K1, TOPK = 3001, 7, 50, 5


class OracleEngine:
    def __init__(self, tt_sd, rk_sd, corpus, offset, ad_table):
        self.tt_sd, self.rk_sd, self.offset, self.ad_table = tt_sd, rk_sd, offset, ad_table
        self.index = oracle.search.FlatIndex(corpus.shape[1])
        self.index.add(corpus)

    def local_search(self, uc, un, k):
        emb = oracle.towers.user_tower(self.tt_sd, uc.numpy(), un.numpy())
        q = oracle.search.normalize_l2(emb)
        D, I = oracle.search.flat_ip_search(self.index.xb, q, k)
        I = np.where(I >= 0, I + self.offset, -1)
        return torch.from_numpy(D), torch.from_numpy(I)

    def merge(self, gathered, world, n_users, k, q0, nq, k_out=None, inexact=None):
        s_bytes, chunk = packed_layout(n_users, k)
        raw = gathered.numpy()
        Ds, Is = [], []
        for g in range(world):
            c = raw[g * chunk:(g + 1) * chunk]
            Ds.append(c[:n_users * k * 4].view(np.float32).reshape(n_users, k)[q0:q0 + nq])
            Is.append(c[s_bytes:].view(np.int32).reshape(n_users, k)[q0:q0 + nq].astype(np.int64))
        if k_out is None or k_out == k:
            D, I = oracle.search.merge_shards(Ds, Is, [0] * world, k)
        else:                                                   # short lists: the CPU statement of amdrec_topk_merge_partial
            D, I, bad = oracle.search.merge_partial(Ds, Is, [0] * world, k_out)
            inexact += int(bad.sum())
        return torch.from_numpy(D), torch.from_numpy(I)

    def rank(self, uc, un, cand_pos, top_k):
        ids_out, sc_out = [], []
        for b in range(uc.shape[0]):
            ids = cand_pos[b].numpy()
            lg = oracle.ranker.forward(self.rk_sd, np.repeat(uc[b:b + 1].numpy(), len(ids), 0), self.ad_table[ids],
                                       np.repeat(un[b:b + 1].numpy(), len(ids), 0))
            top = oracle.pipeline.select_top(lg["ctr"], top_k)
            ids_out.append(ids[top])
            sc_out.append(np.stack([oracle.pipeline.sigmoid(lg[t][top]) for t in oracle.ranker.TASKS]))
        return {"ad_ids": torch.from_numpy(np.stack(ids_out)) if ids_out else torch.zeros((0, top_k), dtype=torch.int64),
                "scores": torch.from_numpy(np.stack(sc_out, axis=1)) if sc_out else torch.zeros((3, 0, top_k)),
                "tasks": list(oracle.ranker.TASKS), "candidate_ids": cand_pos}


def _inputs(B=B):
    user, ad, nnum = cases.small_dims()
    tt_sd = synth.two_tower_state(user, ad, nnum, seed=41)
    rk_sd = synth.ranker_state(user, ad, nnum, seed=42, cross_scale=1.0 / 16)
    ad_table = synth.ad_features(ad, N_ADS, seed=43)
    corpus = oracle.towers.ad_tower(tt_sd, ad_table)
    uc, un = synth.user_batch(user, nnum, B, seed=44)
    return tt_sd, rk_sd, ad_table, corpus, uc, un


def _dup_corpus(corpus, ad_table):
    """Every ad twice, the copy on the other shard of a 2-way split (bit-identical embeddings: score ties everywhere)."""
    h = N_ADS // 2
    return np.concatenate([corpus[:h], corpus[:h]]), np.concatenate([ad_table[:h], ad_table[:h]])


class FlakyEngine(OracleEngine):
    """Reports one unproven query in the first short-list merge on rank 0 (an unlucky 6-sigma query, injected)."""
    calls = 0

    def merge(self, gathered, world, n_users, k, q0, nq, k_out=None, inexact=None):
        out = super().merge(gathered, world, n_users, k, q0, nq, k_out, inexact)
        if inexact is not None and self.offset == 0 and self.calls == 0:
            inexact += 1
        self.calls += 1
        return out


def _worker(rank, world, port, q, B=B, exchange="auto", shard_k=None, sort_corpus=False, dup=False, flaky=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tt_sd, rk_sd, ad_table, corpus, uc, un = _inputs(B)
        if dup:
            corpus, ad_table = _dup_corpus(corpus, ad_table)
        n_ads = len(corpus)
        per = (n_ads + world - 1) // world
        lo, hi = rank * per, min(n_ads, (rank + 1) * per)
        if sort_corpus:          # shard 0 holds user 0's best rows: the premise of short lists (random sharding) is false
            emb = oracle.search.normalize_l2(oracle.towers.user_tower(tt_sd, uc, un))
            order = np.argsort(-(oracle.search.normalize_l2(corpus) @ emb[0]), kind="stable")
            corpus, ad_table = corpus[order], ad_table[order]
        eng = (FlakyEngine if flaky else OracleEngine)(tt_sd, rk_sd, corpus[lo:hi], lo, ad_table)
        sr = ShardedRecommender(None, rank, world, lo, engine=eng, exchange=exchange, shard_k=shard_k)
        used_k = sr.list_k(K1)
        out = sr.recommend_device(torch.from_numpy(uc), torch.from_numpy(un), TOPK, K1)
        q.put((rank, out["user_offset"], out["ad_ids"].numpy(), out["scores"].numpy(),
               out["candidate_ids"].numpy(), out["candidate_scores"].numpy(), used_k, sr.shard_k, sr.short_list_stats(),
               sr.last_exchange))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.timeout(300)
@pytest.mark.parametrize("B,exchange", [(7, "auto"),            # 7 users / 2 ranks: ragged slices -> all-gather
                                        (8, "auto"),            # divisible -> all-to-all (each peer gets its users' lists)
                                        (8, "all_gather")])
def test_two_rank_sharded_pipeline_equals_unsharded_oracle(B, exchange):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, B, exchange)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    tt_sd, rk_sd, ad_table, corpus, uc, un = _inputs(B)
    oidx = oracle.search.FlatIndex(256)
    oidx.add(corpus)
    ref = oracle.pipeline.recommend(tt_sd, rk_sd, oidx, ad_table, uc, un, TOPK, K1)
    covered = []
    for rank, q0, ids, sc, cand, cs, *_ in res:
        assert (q0, len(ids)) == user_slice(B, rank, world)
        for j in range(len(ids)):
            r = ref[q0 + j]
            # sharding must not change anything: merge of exact shard top-k == exact global top-k
            assert np.array_equal(cand[j], r["candidate_ids"])
            assert np.array_equal(cs[j], r["candidate_scores"])
            assert ids[j].tolist() == r["ad_ids"]
            for ti, t in enumerate(oracle.ranker.TASKS):
                assert np.allclose(sc[ti, j], r["scores"][t], atol=1e-6)
            covered.append(q0 + j)
    assert covered == list(range(B))


@pytest.mark.timeout(300)
@pytest.mark.parametrize("shard_k,sort_corpus,exchange", [(40, False, "auto"), (40, False, "all_gather"), (26, True, "auto")])
def test_short_shard_lists_are_proven_exact_or_repeated(shard_k, sort_corpus, exchange):
    """Short lists (amdrec.sharded, amdrec_topk_merge_partial): 40 of 50 entries per shard on a randomly ordered corpus
    -> proven exact, no repeat; 26 entries on a corpus sorted by user 0's score (all of user 0's best rows on shard 0)
    -> the proof fails for 1 of 8 queries (> SHORT_LIST_MAX_FAIL_FRAC), the step is repeated with full lists and short
    lists are switched off.  Either way the result is the unsharded oracle's, bit for bit."""
    world, Bn = 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, Bn, exchange, shard_k, sort_corpus)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    tt_sd, rk_sd, ad_table, corpus, uc, un = _inputs(Bn)
    if sort_corpus:
        emb = oracle.search.normalize_l2(oracle.towers.user_tower(tt_sd, uc, un))
        order = np.argsort(-(oracle.search.normalize_l2(corpus) @ emb[0]), kind="stable")
        corpus, ad_table = corpus[order], ad_table[order]
    oidx = oracle.search.FlatIndex(256)
    oidx.add(corpus)
    ref = oracle.pipeline.recommend(tt_sd, rk_sd, oidx, ad_table, uc, un, TOPK, K1)
    for rank, q0, ids, sc, cand, cs, used_k, shard_k_after, stats, _ in res:
        assert used_k == shard_k
        assert shard_k_after == (None if sort_corpus else shard_k)
        assert stats["batches"] == 1 and stats["repeated_batches"] == int(sort_corpus)
        assert stats["switched_off"] == sort_corpus and (stats["hit_rate"] == 1.0) == (not sort_corpus)
        for j in range(len(ids)):
            r = ref[q0 + j]
            assert np.array_equal(cand[j], r["candidate_ids"])
            assert np.array_equal(cs[j], r["candidate_scores"])
            assert ids[j].tolist() == r["ad_ids"]


@pytest.mark.timeout(300)
def test_four_rank_short_lists_all_to_all():
    """world 4 (two users per rank, all-to-all): rank > 1 offsets, packing and the short-list merge; whether the 24-entry
    lists are proven or the step is repeated with full lists, the result is the unsharded oracle's."""
    world, Bn = 4, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, Bn, "all_to_all", 24, False)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    tt_sd, rk_sd, ad_table, corpus, uc, un = _inputs(Bn)
    oidx = oracle.search.FlatIndex(256)
    oidx.add(corpus)
    ref = oracle.pipeline.recommend(tt_sd, rk_sd, oidx, ad_table, uc, un, TOPK, K1)
    covered = []
    for rank, q0, ids, sc, cand, cs, used_k, shard_k_after, stats, _ in res:
        assert (q0, len(ids)) == user_slice(Bn, rank, world) and used_k == 24 and shard_k_after in (24, None)
        for j in range(len(ids)):
            r = ref[q0 + j]
            assert np.array_equal(cand[j], r["candidate_ids"]) and np.array_equal(cs[j], r["candidate_scores"])
            assert ids[j].tolist() == r["ad_ids"]
            covered.append(q0 + j)
    assert covered == list(range(Bn))
    assert len({r[7] for r in res}) == 1                              # every rank took the same decision


def test_short_list_length_and_merge_rule():
    from amdrec.sharded import short_list_k
    assert [short_list_k(500, w) for w in (1, 2, 4, 8)] == [500, 352, 192, 128]
    assert short_list_k(50, 2) == 50                                   # saves less than a fifth: full lists
    # the merge rule on hand-made lists: shard 0 sends (9, 8), shard 1 sends (7, 1) for k = 3
    D0, I0 = np.array([[9., 8.]], dtype=np.float32), np.array([[0, 1]])
    D1, I1 = np.array([[7., 1.]], dtype=np.float32), np.array([[5, 6]])
    D, I, bad = oracle.search.merge_partial([D0, D1], [I0, I1], [0, 0], 3)
    assert I.tolist() == [[0, 1, 5]] and bad.tolist() == [True]        # shard 0's last (8) >= merged 3rd (7): not proven
    D, I, bad = oracle.search.merge_partial([D0, D1], [I0, I1], [0, 0], 2)
    assert I.tolist() == [[0, 1]] and bad.tolist() == [False]          # shard 0's last entry IS the merged 2nd: what it
    #                                                                    did not send is strictly behind it -> proven
    D0b = np.array([[9., 6.]], dtype=np.float32)
    D, I, bad = oracle.search.merge_partial([D0b, D1], [I0, I1], [0, 0], 2)
    assert I.tolist() == [[0, 5]] and bad.tolist() == [False]          # both lasts (6, 1) below the merged 2nd (7)
    D1s, I1s = np.array([[7., -np.inf]], dtype=np.float32), np.array([[5, -1]])   # a shard with one row: not full
    D, I, bad = oracle.search.merge_partial([D0b, D1s], [I0, I1s], [0, 0], 3)
    assert I.tolist() == [[0, 5, 1]] and bad.tolist() == [False]       # shard 0 full, its last (6) IS the merged 3rd
    # score ties at the boundary are decided by position, like the merge itself: a last entry that ties with the merged
    # k-th at a HIGHER position is behind it (proven); at a LOWER position it is ahead (the shard may hold more such rows)
    Dt0, It0 = np.array([[9., 7.]], dtype=np.float32), np.array([[0, 9]])
    Dt1, It1 = np.array([[7., 7.]], dtype=np.float32), np.array([[5, 6]])
    D, I, bad = oracle.search.merge_partial([Dt0, Dt1], [It0, It1], [0, 0], 3)
    assert I.tolist() == [[0, 5, 6]] and bad.tolist() == [False]       # lasts: (7, pos 9) behind, (7, pos 6) IS the 3rd
    D, I, bad = oracle.search.merge_partial([Dt0, Dt1], [It0, It1], [0, 0], 2)
    assert I.tolist() == [[0, 5]] and bad.tolist() == [False]          # both lasts tie with (7, pos 5) from behind
    Du0, Iu0 = np.array([[9., 7.]], dtype=np.float32), np.array([[0, 3]])
    D, I, bad = oracle.search.merge_partial([Du0, Dt1], [Iu0, It1], [0, 0], 3)
    assert I.tolist() == [[0, 3, 5]] and bad.tolist() == [True]        # shard 0 ends at (7, pos 3), AHEAD of the merged
    #                                                                    3rd (7, pos 5): it may hold an unsent (7, pos 4)


def test_user_slice_and_layout():
    assert [user_slice(7, r, 2) for r in range(2)] == [(0, 4), (4, 3)]
    assert [user_slice(3, r, 4) for r in range(4)] == [(0, 1), (1, 1), (2, 1), (3, 0)]
    assert [user_slice(4096, r, 8) for r in range(8)] == [(512 * r, 512) for r in range(8)]
    s, c = packed_layout(7, 5)
    assert s == 7 * 5 * 4 and c == 2 * s


# ---- sharded IVF with SHARED centroids --------------------------------------------------------
class _FakeIvfIndex:
    """The attributes amdrec.sharded.share_ivf_centroids touches, with the oracle's k-means behind train()."""

    def __init__(self, dim, nlist):
        self.dimension, self.nlist, self.device = dim, nlist, torch.device("cpu")
        self._cent = None

    def train(self, rows):
        self._cent = torch.from_numpy(oracle.search.kmeans_ip(oracle.search.normalize_l2(np.asarray(rows)), self.nlist))

    @property
    def centroids(self):
        return self._cent

    def set_trained_centroids(self, c):
        self._cent = c.clone()


class IvfOracleEngine(OracleEngine):
    def __init__(self, *a, cent=None, nprobe=4, **k):
        super().__init__(*a, **k)
        self.cent, self.nprobe = cent, nprobe
        self.assign = oracle.search.assign_ip(self.index.xb, cent)           # this rank's rows under the SHARED centroids

    def local_search(self, uc, un, k):
        emb = oracle.towers.user_tower(self.tt_sd, uc.numpy(), un.numpy())
        q = oracle.search.normalize_l2(emb)
        D, I = oracle.search.ivf_search(self.index.xb, self.assign, self.cent, q, k, self.nprobe)
        return torch.from_numpy(D), torch.from_numpy(np.where(I >= 0, I + self.offset, -1))


def _ivf_worker(rank, world, port, q):
    from amdrec.sharded import share_ivf_centroids
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tt_sd, rk_sd, ad_table, corpus, uc, un = _inputs(8)
        per = (N_ADS + world - 1) // world
        lo, hi = rank * per, min(N_ADS, (rank + 1) * per)
        idx = _FakeIvfIndex(256, 16)
        cent = share_ivf_centroids(idx, corpus[lo:hi], rank, world)            # rank 0 trains, everyone installs
        assert torch.equal(idx.centroids, cent)
        eng = IvfOracleEngine(tt_sd, rk_sd, corpus[lo:hi], lo, ad_table, cent=cent.numpy(), nprobe=4)
        sr = ShardedRecommender(None, rank, world, lo, engine=eng)
        out = sr.recommend_device(torch.from_numpy(uc), torch.from_numpy(un), TOPK, K1)
        q.put((rank, out["user_offset"], cent.numpy(), out["candidate_ids"].numpy(), out["candidate_scores"].numpy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_ivf_with_shared_centroids_equals_unsharded_ivf():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ivf_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][2], res[1][2])                                # one quantizer on both ranks
    cent = res[0][2]
    tt_sd, rk_sd, ad_table, corpus, uc, un = _inputs(8)
    xb = oracle.search.normalize_l2(corpus)
    emb = oracle.search.normalize_l2(oracle.towers.user_tower(tt_sd, uc, un))
    rD, rI = oracle.search.ivf_search(xb, oracle.search.assign_ip(xb, cent), cent, emb, K1, 4)
    for rank, q0, _, cand, cs in res:
        for j in range(len(cand)):
            # union of the ranks' slices of the probed lists == the unsharded lists: bit-identical result
            assert np.array_equal(cand[j], rI[q0 + j]) and np.array_equal(cs[j], rD[q0 + j])


@pytest.mark.timeout(300)
@pytest.mark.parametrize("shard_k,repeats", [(26, 0), (25, 1)])
def test_short_lists_on_a_corpus_of_duplicated_ads(shard_k, repeats):
    """real catalogs hold ads with bit-identical embeddings (the AdTower embeds categorical features only,
    training_pipeline.py:523), so score ties at the k-th place are routine.  Corpus = every ad twice, the copy on the other
    shard: the merged top-50 is 25 pairs and the 50th entry ties with the 49th.  26-entry lists end below the boundary ->
    proven, no repeat, although the k-th score is tied; 25-entry lists: shard 1's last entry IS the merged 50th (fine),
    shard 0's last is its twin at the lower position, i.e. ahead of it -> not provable (shard 0 might hold a third copy)
    -> THAT batch is repeated with full lists.  Bit-equal to the unsharded oracle either way."""
    world, Bn = 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, Bn, "auto", shard_k, False, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    tt_sd, rk_sd, ad_table, corpus, uc, un = _inputs(Bn)
    corpus, ad_table = _dup_corpus(corpus, ad_table)
    oidx = oracle.search.FlatIndex(256)
    oidx.add(corpus)
    ref = oracle.pipeline.recommend(tt_sd, rk_sd, oidx, ad_table, uc, un, TOPK, K1)
    h = N_ADS // 2
    for rank, q0, ids, sc, cand, cs, used_k, shard_k_after, stats, exch in res:
        assert used_k == shard_k and stats["batches"] == 1 and stats["repeated_batches"] == repeats
        assert exch["kind"] == "all_to_all" and exch["list_k"] == (K1 if repeats else shard_k)
        assert exch["bytes_per_rank"] == Bn * exch["list_k"] * 8
        for j in range(len(ids)):
            r = ref[q0 + j]
            assert np.array_equal(cand[j], r["candidate_ids"]) and np.array_equal(cs[j], r["candidate_scores"])
            assert ids[j].tolist() == r["ad_ids"]
            assert np.array_equal(cand[j][0::2] + h, cand[j][1::2])          # pairs: an ad and its twin on the other shard
            assert cs[j][K1 - 1] == cs[j][K1 - 2]                            # the k-th score IS tied


@pytest.mark.timeout(300)
def test_an_occasional_unproven_query_repeats_its_batch_but_keeps_short_lists():
    """1 unproven query of 32 (3 % < SHORT_LIST_MAX_FAIL_FRAC): the batch is repeated with full lists, the recommender
    keeps its short lists (a tie / an unlucky query must not be a permanent switch)."""
    world, Bn = 2, 32
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, Bn, "auto", 40, False, False, True)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=280) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    tt_sd, rk_sd, ad_table, corpus, uc, un = _inputs(Bn)
    oidx = oracle.search.FlatIndex(256)
    oidx.add(corpus)
    ref = oracle.pipeline.recommend(tt_sd, rk_sd, oidx, ad_table, uc, un, TOPK, K1)
    for rank, q0, ids, sc, cand, cs, used_k, shard_k_after, stats, exch in res:
        assert stats["repeated_batches"] == 1 and stats["unproven_queries"] == 1 and not stats["switched_off"]
        assert shard_k_after == 40 and abs(stats["hit_rate"] - (1 - 1 / 32)) < 1e-9
        assert exch["list_k"] == K1                                           # the repeat ran with full lists
        for j in range(len(ids)):
            r = ref[q0 + j]
            assert np.array_equal(cand[j], r["candidate_ids"]) and ids[j].tolist() == r["ad_ids"]


def _centroid_fail_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from amdrec.sharded import share_ivf_centroids

        class Idx:
            device, nlist, dimension = torch.device("cpu"), 8, 4

            def train(self, rows):
                raise ValueError("Number of training points (3) should be at least as large as number of clusters (8)")

            def set_trained_centroids(self, c):
                raise AssertionError("must not be reached")
        try:
            share_ivf_centroids(Idx(), torch.zeros(3, 4), rank, world)
            q.put((rank, "returned"))
        except Exception as e:                                       # noqa: BLE001
            q.put((rank, type(e).__name__))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_failed_centroid_training_raises_on_every_rank_instead_of_hanging():
    """only rank 0 trains; if that raises, the peers (already waiting in the broadcast) must raise too."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_centroid_fail_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res == {0: "ValueError", 1: "RuntimeError"}


class MarkingEngine(OracleEngine):
    """OracleEngine that accepts the StageTimer's mark callback like the HIP engine does (tower / ranker boundaries)."""

    def local_search(self, uc, un, k, mark=None):
        if mark is not None:
            mark("tower")                               # (the oracle encodes and searches in one call: tower ~ 0 here)
        return super().local_search(uc, un, k)

    def rank(self, uc, un, cand_pos, top_k, mark=None):
        out = super().rank(uc, un, cand_pos, top_k)
        if mark is not None:
            mark("ranker")
        return out


def _timer_worker(rank, world, port, q):
    from amdrec.sharded import STAGES, StageTimer
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        tt_sd, rk_sd, ad_table, corpus, uc, un = _inputs(8)
        per = (len(corpus) + world - 1) // world
        lo, hi = rank * per, min(len(corpus), (rank + 1) * per)
        reports = []
        for eng_cls in (MarkingEngine, OracleEngine):    # an engine without the callback must still yield every key
            sr = ShardedRecommender(None, rank, world, lo, engine=eng_cls(tt_sd, rk_sd, corpus[lo:hi], lo, ad_table), shard_k=40)
            plain = sr.recommend_device(torch.from_numpy(uc), torch.from_numpy(un), TOPK, K1)
            sr.timer = StageTimer("cpu")
            for _ in range(2):
                timed = sr.recommend_device(torch.from_numpy(uc), torch.from_numpy(un), TOPK, K1)
            rep = sr.timer.report()
            sr.timer = None
            assert torch.equal(plain["ad_ids"], timed["ad_ids"])           # the timer changes nothing
            reports.append((rep, sr.timer is None, tuple(STAGES), dist.get_backend(), dist.get_world_size()))
        q.put((rank, reports))
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_stage_timer_reports_every_stage_at_world_2():
    """the multi-GPU bench line carries rank 0's per-stage times {tower, search, pack, exchange, merge,
    proof_wait, ranker, select}; here the same StageTimer runs under gloo at world 2 on the CPU engines."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_timer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, reports in res:
        for rep, detached, stages, backend, ws in reports:
            assert set(rep) == set(stages) == {"tower", "search", "pack", "exchange", "merge", "proof_wait", "ranker", "select"}
            assert all(v >= 0 for v in rep.values()) and detached
            assert rep["search"] > 0 and rep["exchange"] > 0 and rep["merge"] > 0 and rep["proof_wait"] > 0
            assert backend == "gloo" and ws == 2
        assert reports[0][0]["ranker"] > 0               # the marking engine separates the ranker from the selection
