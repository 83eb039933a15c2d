"""Dev tool: per-kernel time of one recommend_device call at a given batch (library profiling hooks)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "movie-recommender-demo_amd")]
import torch
import bench
from amdrec import _lib, synth
from amdrec.index import FAISSIndex
from amdrec.pipeline import AdRecommenderInference
dev = torch.device("cuda:0")
tt, rk, _, (user, ad, nnum) = bench.build_models(dev)
idx = FAISSIndex(256, index_type="Flat", device=dev)
idx.add(bench.device_corpus(1_000_000, 256, dev))
table = torch.from_numpy(synth.ad_features(ad, 1_000_000, seed=99)).to(dev)
rec = AdRecommenderInference(two_tower_model=tt, transformer_ranker=rk, faiss_index=idx, ad_features=table)
uc, un = synth.user_batch(user, nnum, 512, seed=1)
uc, un = torch.from_numpy(uc).to(dev), torch.from_numpy(un).to(dev)
for B in [int(a) for a in sys.argv[1:]] or [64]:
    for _ in range(3):
        rec.recommend_device(uc[:B], un[:B], 10, 500)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        rec.recommend_device(uc[:B], un[:B], 10, 500)
    e1.record(); torch.cuda.synchronize()
    p = _lib.profile_report(); _lib.profile_enable(False)
    print(B, round(e0.elapsed_time(e1) / 10, 3), {k: round(v["total_ms"] / 10, 3) for k, v in sorted(p.items())}, flush=True)
