"""Dev tool: search-only loop at a given batch for rocprofv3 (per-kernel durations of the search stage).
usage: python tools/search_only.py B [reps] [n_ads] [k]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "movie-recommender-demo_amd"))
import torch
from amdrec.index import FAISSIndex

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000
K = int(sys.argv[4]) if len(sys.argv) > 4 else 500
pref = os.environ.get("PREFILTER", "bf16")
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(1)
idx = FAISSIndex(256, index_type="Flat", prefilter=pref)
for s in range(0, n, 250_000):
    idx.add(torch.randn((min(250_000, n - s), 256), generator=g, device=dev))
q = torch.randn((B, 256), generator=g, device=dev)
q = q / q.norm(dim=1, keepdim=True)
for _ in range(3):
    idx.search_device(q, K, normalize=False)
torch.cuda.synchronize()
from amdrec import _lib
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):                      # whole call, no per-launch events inside (they idle the stream ~10 us each)
    idx.search_device(q, K, normalize=False)
e1.record(); torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(reps):                      # per-kernel breakdown
    idx.search_device(q, K, normalize=False)
torch.cuda.synchronize()
prof = _lib.profile_report()
parts = " ".join(f"{k.replace('search_', '')}={v['total_ms'] / v['launches']:.4f}" for k, v in sorted(prof.items()) if v["launches"])
print(f"B={B} k={K} prefilter={pref} lib={os.path.basename(_lib.LIB_PATH)} ms/search={e0.elapsed_time(e1) / reps:.4f}  {parts}")
