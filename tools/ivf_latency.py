"""Dev tool: IVF-Flat search latency by batch size (the reference's default index: nlist 100, nprobe 10), per-kernel breakdown.
usage: python tools/ivf_latency.py [n_ads] [nlist] [nprobe]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "movie-recommender-demo_amd"))
import torch  # noqa: E402
from amdrec import _lib  # noqa: E402
from amdrec.index import FAISSIndex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
nlist = int(sys.argv[2]) if len(sys.argv) > 2 else 100
nprobe = int(sys.argv[3]) if len(sys.argv) > 3 else 10
dev = torch.device("cuda:0")
g = torch.Generator(device=dev)
g.manual_seed(1)
idx = FAISSIndex(256, index_type="IVF", nlist=nlist, nprobe=nprobe)
x = torch.randn((n, 256), generator=g, device=dev)
idx.add(x)
for B in (1, 8, 64, 512):
    q = torch.randn((B, 256), generator=g, device=dev)
    for _ in range(3):
        idx.search_device(q, 500)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        idx.search_device(q, 500)
    e1.record()
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(5):
        idx.search_device(q, 500)
    rep = _lib.profile_report()
    _lib.profile_enable(False)
    parts = " ".join(f"{k}={v['total_ms'] / 5:.4f}x{v['launches'] // 5}" for k, v in sorted(rep.items()))
    print(f"IVF({nlist},{nprobe}) n={n} B={B}: {e0.elapsed_time(e1) / 20:.4f} ms/search  {parts}", flush=True)
