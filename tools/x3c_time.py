"""Launch time of the ranker's row-owner kernels on small passes: the column-split kernel (csrc/rowowner16c.hpp) against
the 64-row workgroup shape of the 16-row kernel.  AMDREC_LIB_PATH = a diagnostic build (-DAMDREC_X3C_DBG=<bits>, see the
header) gives elimination runs; with bit 16 the kernel's cycle stamps are printed.
    python tools/x3c_time.py [rows ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "movie-recommender-demo_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from amdrec import _lib                       # noqa: E402
from amdrec.ranker import TransformerRanker   # noqa: E402
from tests import cases                       # noqa: E402

user, ad, nnum, sd, _ = cases.ranker_case("demo", "randn")
m = TransformerRanker(dict(user), dict(ad), nnum)
m.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
m.x3_variant = 16
m = m.cuda().eval()
lib = _lib.load()
stamps = "cdbg16" in os.environ.get("AMDREC_LIB_PATH", "")
for rows in [int(a) for a in sys.argv[1:]] or [500, 1000, 2000, 4096, 8192]:
    X = torch.randn(rows, 256, device="cuda")
    for cs in (-1, 0):
        m.x3_cs_max_rows = cs if rows <= 4096 or cs == -1 else 1 << 20
        params, tasks = m._pack(X.device)
        logits = torch.zeros((3 + 64, max(rows, 512)), device="cuda")
        ws = torch.empty(((rows + 127) // 128) * 128 * 1024, dtype=torch.uint8, device="cuda")

        def run():
            _lib.check(lib.amdrec_ranker_x3_prefix(C.byref(params), _lib.ptr(X), X.stride(0), rows, -1, None, 0, _lib.ptr(logits),
                                                   logits.stride(0), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(X.device)))
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if os.environ.get("FLUSH") == "1":                 # the serving pipeline's situation: a 1 GB scan precedes every pass
            big = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
            tot = 0.0
            for _ in range(20):
                big.add_(1.0)
                a.record()
                run()
                b.record()
                torch.cuda.synchronize()
                tot += a.elapsed_time(b)
            us = tot / 20 * 1e3
        else:
            a.record()
            for _ in range(50):
                run()
            b.record()
            torch.cuda.synchronize()
            us = a.elapsed_time(b) / 50 * 1e3
        line = f"rows={rows} cs_max_rows={m.x3_cs_max_rows}: {us:.1f} us per launch"
        if stamps and m.x3_cs_max_rows >= 0:
            n_w = (rows + 15) // 16 * 4
            d = logits.view(-1)[3 * logits.stride(0):3 * logits.stride(0) + 4 * n_w].view(n_w, 4).cpu().numpy()
            line += (f" | s_memtime ticks (10 ns) per wave, mean: kernel {d[:, 0].mean():.0f}, vmcnt+lgkm waits {d[:, 1].mean():.0f}, "
                     f"barriers {d[:, 2].mean():.0f}")
        print(line, flush=True)
