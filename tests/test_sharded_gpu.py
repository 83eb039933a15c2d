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
    finally:
        dist.destroy_process_group()
