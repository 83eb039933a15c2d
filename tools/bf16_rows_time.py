"""Dev tool: amdrec_bf16_rows (fp32 rows -> bf16 shadow + the two norm maxima) timed with events on N x 256 rows."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "movie-recommender-demo_amd"))
import torch  # noqa: E402
import bench  # noqa: E402
from amdrec import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
dev = torch.device("cuda", 0)
x = bench.device_corpus(n, bench.DIM, dev)
out = torch.empty((n, bench.DIM), dtype=torch.bfloat16, device=dev)
mx = torch.zeros(2, dtype=torch.float32, device=dev)
lib = _lib.load()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for rep in range(4):
    mx.zero_()
    e0.record()
    _lib.check(lib.amdrec_bf16_rows(_lib.ptr(x), n, x.stride(0), bench.DIM, _lib.ptr(out), out.stride(0), _lib.ptr(mx), _lib.stream_ptr(dev)))
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(f"{os.environ.get('AMDREC_LIB_PATH', 'product')}: bf16_rows N={n}: {ms:.2f} ms = {n * bench.DIM * 6 / ms / 1e6:.0f} GB/s, maxima {mx.tolist()}")
