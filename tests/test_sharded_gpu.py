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
