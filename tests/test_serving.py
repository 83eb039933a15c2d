"""CPU test of the micro-batcher (host-side plumbing; the batch function is injected)."""
import threading

from amdrec.serving import MicroBatcher


def test_concurrent_requests_are_coalesced_and_routed_back():
    def batch_fn(users):
        return [{"ad_ids": [u["id"] * 10 + j for j in range(3)]} for u in users]
    mb = MicroBatcher(batch_fn, max_batch=16, max_wait_ms=50)
    out = {}

    def call(i):
        out[i] = mb.recommend_ads({"id": i})
    threads = [threading.Thread(target=call, args=(i,)) for i in range(40)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(10)
    mb.close()
    assert all(out[i]["ad_ids"] == [i * 10, i * 10 + 1, i * 10 + 2] for i in range(40))
    assert sum(mb.batches) == 40 and max(mb.batches) <= 16 and len(mb.batches) < 40     # coalesced


def test_errors_reach_every_caller_and_the_worker_survives():
    calls = {"n": 0}

    def batch_fn(users):
        calls["n"] += 1
        if calls["n"] == 1:
            raise ValueError("boom")
        return [{"ok": True} for _ in users]
    mb = MicroBatcher(batch_fn, max_batch=4, max_wait_ms=1)
    try:
        mb.recommend_ads({"id": 0})
        assert False
    except ValueError:
        pass
    assert mb.recommend_ads({"id": 1}) == {"ok": True}
    mb.close()
