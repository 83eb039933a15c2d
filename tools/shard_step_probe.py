"""Dev tool: the COMPUTE of one rank's step at the weak-scaling shape of a G-rank run, on one GPU, without the collective:
tower for the global batch (512 G users) -> search of a 1M/G-row shard with short lists -> pack -> merge of G lists (the
rank's own list replicated at shifted positions stands in for the peers') -> ranker for the rank's 512 users -> top-10.
usage: python tools/shard_step_probe.py G [reps]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "movie-recommender-demo_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from amdrec import synth  # noqa: E402
from amdrec.index import FAISSIndex  # noqa: E402
from amdrec.pipeline import AdRecommenderInference  # noqa: E402
from amdrec.sharded import HipEngine, packed_layout, short_list_k  # noqa: E402

G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
tt, rk, _, dims = bench.build_models(dev)
user, ad, nnum = dims
rows = bench.N_ADS // G
index = FAISSIndex(bench.DIM, index_type="Flat", device=dev)
index.add(bench.device_corpus(bench.N_ADS, bench.DIM, dev, row0=0, rows=rows))
ad_table = torch.from_numpy(synth.ad_features(ad, bench.N_ADS, seed=99)).to(dev)
rec = AdRecommenderInference(two_tower_model=tt, transformer_ranker=rk, faiss_index=index, ad_features=ad_table)
B = 512 * G
uc, un = synth.user_batch(user, nnum, B, seed=2024)
uc, un = torch.from_numpy(uc).to(dev), torch.from_numpy(un).to(dev)
eng = HipEngine(rec, 0)
k, kq = bench.STAGE1_K, short_list_k(bench.STAGE1_K, G) if G > 1 else bench.STAGE1_K
nq = 512
inexact = torch.zeros(1, dtype=torch.int32, device=dev)


def step():
    scores, pos = eng.local_search(uc, un, kq)
    s_bytes, chunk = packed_layout(nq, kq)
    buf = torch.empty(chunk * G, dtype=torch.uint8, device=dev)
    bv = buf.view(G, chunk)
    for g in range(G):                                   # stand-in for the all-to-all: G lists of this rank's 512 users
        bv[g, :s_bytes].view(torch.float32).copy_(scores[:nq].reshape(-1))
        bv[g, s_bytes:].view(torch.int32).copy_((pos[:nq] + g * rows).reshape(-1))
    if kq < k:
        cs, cp = eng.merge(buf, G, nq, kq, 0, nq, k, inexact)
    else:
        cs, cp = eng.merge(buf, G, nq, kq, 0, nq)
    return eng.rank(uc[:nq], un[:nq], cp, bench.TOP_K)


for _ in range(5):
    step()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
print(f"G={G} list_k={kq} per-rank compute {ms:.3f} ms/step ({2 * G - 2} stand-in copies included) -> {512 * G / ms * 1e3:.0f} recs/s if the exchange were free")
