"""Dev tool: wall time of FAISSIndex.add (normalise + bf16 shadow) for N device-resident rows, Flat.  usage: python tools/index_build_time.py [N]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "movie-recommender-demo_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from amdrec import _lib  # noqa: E402
from amdrec.index import FAISSIndex  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
x = bench.device_corpus(n, bench.DIM, dev)
for rep in range(2):
    idx = FAISSIndex(bench.DIM, index_type="Flat", device=dev)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    t0 = time.perf_counter()
    idx.add(x)
    q = torch.randn(8, bench.DIM, device=dev)
    idx.search_device(q, 10)                        # (the shadow is built lazily at the first search)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rep_ = _lib.profile_report()
    _lib.profile_enable(False)
    print(f"N={n} add + first search: {dt * 1e3:.1f} ms", {k: round(v['total_ms'], 2) for k, v in rep_.items() if v['total_ms'] > 0.5})
    del idx
