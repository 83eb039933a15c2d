"""Dev tool: end-to-end recommend_device latency by batch size with the ranker's small batches on (a) the fp32-MFMA
small-shape kernels (x3_min_rows = 8193, many launches) and (b) the row-owner kernel (x3_min_rows = 1, one launch)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "movie-recommender-demo_amd")]
import torch
import bench
from amdrec import synth
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
rows = []
for B in (1, 2, 4, 8, 16, 32, 64, 128, 512):
    row = {"B": B, "ranker_rows": B * 500}
    for name, mr in (("small_shapes_ms", 8193), ("rowowner_ms", 1)):
        rk.x3_min_rows = mr
        for _ in range(3):
            rec.recommend_device(uc[:B], un[:B], 10, 500)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            rec.recommend_device(uc[:B], un[:B], 10, 500)
        e1.record(); torch.cuda.synchronize()
        row[name] = round(e0.elapsed_time(e1) / 20, 3)
    rows.append(row)
    print(json.dumps(row), flush=True)
