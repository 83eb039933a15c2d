"""Dev tool: the COMPUTE of one rank's step at the weak-scaling shape of a G-rank run, on one GPU, without the collective:
tower for the global batch (512 G users) -> search of a 1M/G-row shard with short lists -> pack -> merge of G lists (the
rank's own list replicated at shifted positions stands in for the peers') -> ranker for the rank's 512 users -> top-10.
usage: python tools/shard_step_probe.py G [reps] [--index ivf [--ads N] [--nlist L] [--nprobe P]]
--index ivf: configs[4] at the shape a rank of G sees - a (N / G)-row shard of the clustered N-ad corpus (default 10M) filed
under ONE coarse quantizer of nlist centroids (default 4096, trained on this shard as rank 0 would), 512 G queries, nprobe 64,
short per-shard lists; prints the search's per-kernel breakdown and the list scan's bytes against HBM."""
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

import argparse  # noqa: E402
ap = argparse.ArgumentParser()
ap.add_argument("G", type=int, nargs="?", default=8)
ap.add_argument("reps", type=int, nargs="?", default=10)
ap.add_argument("--index", choices=["flat", "ivf"], default="flat")
ap.add_argument("--ads", type=int, default=None)
ap.add_argument("--nlist", type=int, default=4096)
ap.add_argument("--nprobe", type=int, default=64)
A = ap.parse_args()
G, reps = A.G, A.reps
dev = torch.device("cuda", 0)
tt, rk, _, dims = bench.build_models(dev)
user, ad, nnum = dims
n_ads = A.ads or (10_000_000 if A.index == "ivf" else bench.N_ADS)
rows = n_ads // G
if A.index == "ivf":
    shard = bench.device_corpus(n_ads, bench.DIM, dev, row0=0, rows=rows, kind="clustered")
    index = FAISSIndex(bench.DIM, index_type="IVF", nlist=A.nlist, nprobe=A.nprobe, device=dev)
    index.train(shard)                                   # rank 0 trains on its shard; the peers would receive the centroids
    index.add(shard)
    del shard
else:
    index = FAISSIndex(bench.DIM, index_type="Flat", device=dev)
    index.add(bench.device_corpus(n_ads, bench.DIM, dev, row0=0, rows=rows))
ad_table = torch.from_numpy(synth.ad_features(ad, rows * G if A.index == "ivf" else bench.N_ADS, seed=99)).to(dev)
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
print(f"G={G} index={A.index} ads={n_ads} list_k={kq} per-rank compute {ms:.3f} ms/step ({2 * G - 2} stand-in copies included) -> {512 * G / ms * 1e3:.0f} recs/s if the exchange were free")
# the search alone, whole call and per kernel
from amdrec import _lib  # noqa: E402
emb = rec.two_tower_model.user_tower.encode(uc, un, check_indices=False, renormalize=True)
for _ in range(3):
    index.search_device(emb, kq, normalize=False, return_positions=True)
torch.cuda.synchronize()
e0.record()
for _ in range(reps):
    index.search_device(emb, kq, normalize=False, return_positions=True)
e1.record()
torch.cuda.synchronize()
sms = e0.elapsed_time(e1) / reps
_lib.profile_enable(True)
for _ in range(reps):
    index.search_device(emb, kq, normalize=False, return_positions=True)
rep = _lib.profile_report()
_lib.profile_enable(False)
parts = " ".join(f"{k}={v['total_ms'] / reps:.4f}x{v['launches'] // reps}" for k, v in sorted(rep.items()))
print(f"search of {B} queries x {rows} rows, k={kq}: {sms:.4f} ms  {parts}")
if A.index == "ivf":
    st = index._ivf
    _, _, _, lens, max_len, _ = st._build_lists(index._xb, index._n)
    cs = torch.empty((B, A.nprobe), dtype=torch.float32, device=dev)
    probes = torch.empty((B, A.nprobe), dtype=torch.int64, device=dev)
    from amdrec.index import flat_search  # noqa: E402
    flat_search(st.centroids, st.nlist, emb.contiguous(), A.nprobe, cs, probes)
    cnt = torch.bincount(probes.reshape(-1), minlength=st.nlist)
    lf = lens.float()
    once = int((lens * (cnt > 0)).sum().item()) * bench.DIM * 4            # every probed list once
    for qt in (32, 64):
        tiles = int((((cnt + qt - 1) // qt) * lens).sum().item()) * bench.DIM * 4
        print(f"  list bytes if read once per {qt}-query tile: {tiles / 1e9:.3f} GB (once per probed list: {once / 1e9:.3f} GB)")
    nf = max(2, A.nprobe // 8)                             # the two-phase scan: nearest nf probes unfiltered (32-query tiles here)
    for name, pr, qt in (("phase 1", probes[:, :nf], 32), ("phase 2", probes[:, nf:], 64)):
        c = torch.bincount(pr.reshape(-1), minlength=st.nlist)
        qtiles = (c + qt - 1) // qt
        for bq in (128, 256):
            wgs = int((qtiles * ((lens + bq - 1) // bq)).sum().item())
            print(f"  {name}: {int(qtiles.sum())} (list, {qt}-query tile) groups, {wgs} row tiles of {bq}; "
                  f"fp32-MFMA time of those tiles {wgs * 8 * (bq // 32) * (qt // 32) * 16 * 64 / 4 / 256 / 2.4e9 * 1e3:.3f} ms")
    sc = sum(v["total_ms"] for k_, v in rep.items() if k_.startswith("ivf_scan")) / reps
    print(f"  lists: len min/mean/max {int(lens.min())}/{lf.mean():.1f}/{max_len}; queries per probed list mean {cnt.float().mean():.1f} max {int(cnt.max())}; "
          f"scan {sc:.4f} ms = {once / sc / 1e6 / 8000:.3f} of HBM peak on the once-per-list bytes")
