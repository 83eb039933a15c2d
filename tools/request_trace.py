"""Dev tool: N single requests (recommend_device, B users, 1M ads Flat, the bench's models) back to back, for a rocprofv3 kernel
trace of the request's launches and the gaps between them.  usage: rocprofv3 --kernel-trace ... -- python3 tools/request_trace.py [B] [N]"""
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

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
tt, rk, _, (user, ad, nnum) = bench.build_models(dev)
index = FAISSIndex(bench.DIM, index_type="Flat", device=dev)
index.add(bench.device_corpus(bench.N_ADS, bench.DIM, dev))
ad_table = torch.from_numpy(synth.ad_features(ad, bench.N_ADS, seed=99)).to(dev)
rec = AdRecommenderInference(two_tower_model=tt, transformer_ranker=rk, faiss_index=index, ad_features=ad_table)
uc, un = synth.user_batch(user, nnum, B, seed=5)
uc, un = torch.from_numpy(uc).to(dev), torch.from_numpy(un).to(dev)
for _ in range(10):
    rec.recommend_device(uc, un)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(N):
    rec.recommend_device(uc, un)
e1.record()
torch.cuda.synchronize()
print(f"B={B}: {e0.elapsed_time(e1) / N:.4f} ms per request (back to back, device-resident inputs)")
# one request at a time (the host waits for each): what a caller that needs the result sees
import time  # noqa: E402
t0 = time.perf_counter()
for _ in range(N):
    rec.recommend_device(uc, un)
    torch.cuda.synchronize()
print(f"B={B}: {(time.perf_counter() - t0) / N * 1e3:.4f} ms per request (synchronised after each)")
