"""GPU: the cross-shard merge kernel and the sharded recommender (world of 1 rank under RCCL,
plus several shards emulated on one device) give bit-identical results to the unsharded search."""
import os

import numpy as np
import pytest
import torch

import oracle
from amdrec import synth

pytestmark = pytest.mark.gpu


def test_merge_of_shard_topk_is_bit_identical_to_unsharded_search():
    from amdrec import _lib
    from amdrec.index import FAISSIndex
    from amdrec.sharded import HipEngine, packed_layout
    n, nq, k, G = 40_000, 37, 500, 4
    xb, xq = synth.unit_corpus(n, 256, seed=1), synth.unit_corpus(nq, 256, seed=2)
    full = FAISSIndex(256, index_type="Flat")
    full.add(xb)
    q = torch.from_numpy(xq).cuda()
    ref_pos, ref_sc = full.search_device(q, k, return_positions=True)
    s_bytes, chunk = packed_layout(nq, k)
    gathered = torch.empty(chunk * G, dtype=torch.uint8, device="cuda")
    per = (n + G - 1) // G
    for g in range(G):
        lo, hi = g * per, min(n, (g + 1) * per)
        sh = FAISSIndex(256, index_type="Flat")
        sh.add(xb[lo:hi])
        pos, sc = sh.search_device(q, k, return_positions=True, pos_offset=lo)
        c = gathered[g * chunk:(g + 1) * chunk]
        c[:nq * k * 4].view(torch.float32).copy_(sc.reshape(-1))
        c[s_bytes:].view(torch.int32).copy_(pos.reshape(-1))
    eng = HipEngine(None, 0)
    for q0, m in ((0, nq), (5, 11), (36, 1)):
        sc, pos = eng.merge(gathered, G, nq, k, q0, m)
        assert torch.equal(pos, ref_pos[q0:q0 + m]) and torch.equal(sc, ref_sc[q0:q0 + m])
    # shards smaller than k: unfilled (-1) slots are ignored by the merge
    tiny = torch.empty(chunk * 2, dtype=torch.uint8, device="cuda")
    for g, (lo, hi) in enumerate(((0, 300), (300, 700))):
        sh = FAISSIndex(256, index_type="Flat")
        sh.add(xb[lo:hi])
        pos, sc = sh.search_device(q, k, return_positions=True, pos_offset=lo)
        c = tiny[g * chunk:(g + 1) * chunk]
        c[:nq * k * 4].view(torch.float32).copy_(sc.reshape(-1))
        c[s_bytes:].view(torch.int32).copy_(pos.reshape(-1))
    sm = FAISSIndex(256, index_type="Flat")
    sm.add(xb[:700])
    rp, rs = sm.search_device(q, k, return_positions=True)
    sc, pos = eng.merge(tiny, 2, nq, k, 0, nq)
    assert torch.equal(pos, rp) and torch.equal(sc, rs)


@pytest.mark.parametrize("list_k,clustered", [(128, False), (96, False), (128, True)])
def test_short_list_merge_is_proven_exact_or_flagged(list_k, clustered):
    """amdrec_topk_merge_partial over 8 emulated shards of a 200k-row corpus, top-500: every shard sends its best
    list_k rows.  Random sharding: the merge equals the unsharded HIP search bit for bit, equals the oracle's merge rule
    (oracle.search.merge_partial) and counts nothing inexact.  Clustered: the rows nearest to queries 0..3 all sit on
    shard 2, whose list is cut off above the merged 500th score -> exactly those queries are counted."""
    from amdrec.index import FAISSIndex
    from amdrec.sharded import HipEngine, packed_layout
    n, nq, k, G = 200_000, 24, 500, 8
    xb, xq = synth.unit_corpus(n, 256, seed=11), synth.unit_corpus(nq, 256, seed=12)
    per = n // G
    if clustered:
        rng = np.random.default_rng(5)
        for qi in range(4):                                   # 400 near-copies of each of the first four queries on shard 2
            rows = xq[qi][None, :] + 0.05 * rng.standard_normal((400, 256)).astype(np.float32)
            xb[2 * per + 400 * qi:2 * per + 400 * (qi + 1)] = rows / np.linalg.norm(rows, axis=1, keepdims=True)
    full = FAISSIndex(256, index_type="Flat")
    full.add(xb)
    q = torch.from_numpy(xq).cuda()
    ref_pos, ref_sc = full.search_device(q, k, return_positions=True)
    s_bytes, chunk = packed_layout(nq, list_k)
    gathered = torch.empty(chunk * G, dtype=torch.uint8, device="cuda")
    Ds, Is = [], []
    for g in range(G):
        lo, hi = g * per, (g + 1) * per
        sh = FAISSIndex(256, index_type="Flat")
        sh.add(xb[lo:hi])
        pos, sc = sh.search_device(q, list_k, return_positions=True, pos_offset=lo)
        c = gathered[g * chunk:(g + 1) * chunk]
        c[:nq * list_k * 4].view(torch.float32).copy_(sc.reshape(-1))
        c[s_bytes:].view(torch.int32).copy_(pos.reshape(-1))
        Ds.append(sc.cpu().numpy())
        Is.append(pos.cpu().numpy())
    oD, oI, bad = oracle.search.merge_partial(Ds, Is, [0] * G, k)
    inexact = torch.zeros(1, dtype=torch.int32, device="cuda")
    eng = HipEngine(None, 0)
    sc, pos = eng.merge(gathered, G, nq, list_k, 0, nq, k, inexact)
    assert np.array_equal(pos.cpu().numpy(), oI) and np.array_equal(sc.cpu().numpy(), oD)
    assert int(inexact.item()) == int(bad.sum())
    if clustered:
        assert bad[:4].all() and not bad[4:].any()
        ok = slice(4, nq)
    else:
        # 8 shards x 96 leaves less slack (62.5 +- 7.4 per shard): whatever is not proven must at least be flagged
        ok = ~bad
        assert list_k == 96 or not bad.any()
    ok = torch.from_numpy(np.arange(nq)[ok]).cuda()
    assert torch.equal(pos[ok], ref_pos[ok]) and torch.equal(sc[ok], ref_sc[ok])
    # a sub-range of the users, as a rank merging only its own slice does; the counter accumulates
    sc2, pos2 = eng.merge(gathered, G, nq, list_k, 3, 9, k, inexact)
    assert np.array_equal(pos2.cpu().numpy(), oI[3:12]) and int(inexact.item()) == int(bad.sum()) + int(bad[3:12].sum())


@pytest.mark.parametrize("list_k", [250, 251, 320])
def test_short_list_merge_on_duplicated_ads_decides_score_ties_by_position(list_k):
    """a corpus of duplicated ads (every row twice, the copy on the other shard) makes the merged k-th score a
    tie for every query.  amdrec_topk_merge_partial must decide like the merge itself, by (score, position): 251+ entries
    per shard end below the boundary -> proven although the k-th score is tied; 250 entries: shard 1's last entry IS the
    merged 500th (fine), shard 0's last is its twin at the lower position = ahead of it -> counted for every query.
    Result and count equal oracle.search.merge_partial; a proven result equals the unsharded HIP search bit for bit."""
    from amdrec.index import FAISSIndex
    from amdrec.sharded import HipEngine, packed_layout
    h, nq, k, G = 30_000, 12, 500, 2
    half = synth.unit_corpus(h, 256, seed=21)
    xb, xq = np.concatenate([half, half]), synth.unit_corpus(nq, 256, seed=22)
    full = FAISSIndex(256, index_type="Flat")
    full.add(xb)
    q = torch.from_numpy(xq).cuda()
    ref_pos, ref_sc = full.search_device(q, k, return_positions=True)
    # premise: every returned row comes with its twin at the same score, and the k-th boundary splits no pair (two DIFFERENT
    # rows may tie exactly too - then the four interleave by position - so adjacency is only required at the boundary)
    rp, rs = ref_pos.cpu().numpy(), ref_sc.cpu().numpy()
    srt = np.sort(rp % h, axis=1)
    assert np.array_equal(srt[:, 0::2], srt[:, 1::2])
    for qi in range(nq):
        by_pos = dict(zip(rp[qi].tolist(), rs[qi].tolist()))
        assert all(by_pos[p_] == by_pos[p_ + h] for p_ in rp[qi] if p_ < h)
    assert torch.equal(ref_pos[:, k - 2] + h, ref_pos[:, k - 1]) and torch.equal(ref_sc[:, k - 1], ref_sc[:, k - 2])
    s_bytes, chunk = packed_layout(nq, list_k)
    gathered = torch.empty(chunk * G, dtype=torch.uint8, device="cuda")
    Ds, Is = [], []
    for g in range(G):
        sh = FAISSIndex(256, index_type="Flat")
        sh.add(xb[g * h:(g + 1) * h])
        pos, sc = sh.search_device(q, list_k, return_positions=True, pos_offset=g * h)
        c = gathered[g * chunk:(g + 1) * chunk]
        c[:nq * list_k * 4].view(torch.float32).copy_(sc.reshape(-1))
        c[s_bytes:].view(torch.int32).copy_(pos.reshape(-1))
        Ds.append(sc.cpu().numpy())
        Is.append(pos.cpu().numpy())
    oD, oI, bad = oracle.search.merge_partial(Ds, Is, [0] * G, k)
    inexact = torch.zeros(1, dtype=torch.int32, device="cuda")
    sc, pos = HipEngine(None, 0).merge(gathered, G, nq, list_k, 0, nq, k, inexact)
    assert np.array_equal(pos.cpu().numpy(), oI) and np.array_equal(sc.cpu().numpy(), oD)
    assert int(inexact.item()) == int(bad.sum()) == (nq if list_k == 250 else 0)
    assert torch.equal(pos, ref_pos) and torch.equal(sc, ref_sc)      # (here even the unproven merge happens to be right)


def test_bench_rehearsal_two_ranks_on_one_gpu_prints_n_gpus_2():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts its two ranks itself; under the rehearsal
    switch both run on cuda:0 over gloo.  One JSON line, n_gpus 2, the exchange described in config."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["AMDREC_BENCH_REHEARSE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--ads", "200000"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["scaling"] == "weak" and doc["value"] > 0
    cfg = doc["config"]
    assert cfg["world"] == 2 and cfg["backend"] == "gloo" and cfg["exchange"] == "all_to_all"
    assert cfg["bytes_per_rank"] == 1024 * cfg["shard_lists"]["list_k_timed"] * 8
    assert cfg["users_per_step"] == 1024 and cfg["shard_lists"]["inexact_in_timed_steps"] == 0
    # the N > 1 line localises its own problems
    mg = doc["multi_gpu"]
    assert mg["backend"] == "gloo" and mg["world_size"] == 2
    assert set(mg["stages_ms_rank0"]) == {"tower", "search", "pack", "exchange", "merge", "proof_wait", "ranker", "select"}
    assert all(mg["stages_ms_rank0"][k] > 0 for k in ("tower", "search", "exchange", "merge", "ranker", "select"))
    assert mg["step_ms_max_over_ranks"] >= mg["step_ms_min_over_ranks"] > 0
    assert mg["verified"]["value"] > 0 and "unverified" in mg
    assert doc["user_batches"] == 8 and doc["same_batch"]["value"] > 0 and doc["warmup_steps_run"] >= 20


def test_sharded_recommender_single_rank_rccl_matches_pipeline():
    import torch.distributed as dist
    from amdrec.sharded import ShardedRecommender
    from tests.test_pipeline_gpu import _setup
    rec, _, (user, ad, nnum) = _setup(6000, 1.0 / 16)
    uc, un = synth.user_batch(user, nnum, 6, seed=3)
    uc, un = torch.from_numpy(uc).cuda(), torch.from_numpy(un).cuda()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ.setdefault("MASTER_PORT", "29541")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sr = ShardedRecommender(rec, 0, 1, shard_offset=0)
        a = sr.recommend_device(uc, un, 10, 500)
        allr = sr.recommend_all(uc, un, 10, 500)
        b = rec.recommend_device(uc, un, 10, 500)
        torch.cuda.synchronize()
        assert torch.equal(a["ad_ids"], b["ad_ids"]) and torch.equal(a["scores"], b["scores"])
        assert torch.equal(allr["ad_ids"], b["ad_ids"]) and torch.equal(allr["scores"], b["scores"])
        # the RCCL entry points of the exchange on packed uint8 device buffers (world 1: both are a copy)
        from amdrec.sharded import all_gather_bytes, all_to_all_bytes
        src = torch.randint(0, 255, (4096,), dtype=torch.uint8, device="cuda")
        d1, d2 = torch.empty_like(src), torch.empty_like(src)
        all_to_all_bytes(d1, src)
        all_gather_bytes(d2, src)
        torch.cuda.synchronize()
        assert torch.equal(d1, src) and torch.equal(d2, src)
        # the asynchronous proof read of the verified short-list mode (all-reduce with async_op, side stream, pinned flag)
        # over real RCCL: with one rank the sum is the rank's own counter.  (`world = 2` only selects the code path; the
        # collective runs on the 1-rank group.)
        sr2 = ShardedRecommender(rec, 0, 1, shard_offset=0)
        sr2.world = 2
        for want in (3, 0, 41):
            sr2._inexact = torch.tensor([want], dtype=torch.int32, device="cuda")
            busy = torch.randn(4096, 4096, device="cuda") @ torch.randn(4096, 4096, device="cuda")     # work queued behind it
            h = sr2._proof_begin()
            busy = busy @ busy                                                                          # ... and after it
            assert h[0] == "async" and sr2._proof_end(h) == want
            assert int(sr2._inexact.item()) == 0                                                        # read-and-reset
        del busy
    finally:
        dist.destroy_process_group()
