#!/usr/bin/env python3
"""End-to-end recommendations/s on the BASELINE.json workload.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (one child process
per GPU, spawned before this process imports torch or touches the GPU; the children are fresh interpreters, nothing
re-execs); under torchrun the ranks are torchrun's.  A world that differs from --gpus exits non-zero.

Workload (BASELINE.json metric / configs[2]): synthetic corpus of 1 000 000 ads,
d = 256 (randn rows L2-normalised, seed 1234, the reference benchmark's distribution
faiss_retrieval.py:390), per-ad categorical table ad_cat[1M,20], seeded random-init two-tower
and ranker weights with the reference's architecture; one STEP = one batch of 512 users per GPU
through the whole hot path with inputs resident in HBM:
    UserTower -> L2 renorm -> exact inner-product top-500 -> TransformerRanker on the 500
    candidates of every user -> top-10 by CTR logit (+ sigmoid of the 3 tasks).
N > 1: the corpus is row-sharded over the ranks (1M/N rows each), every rank searches its shard
for the global batch (512*N users), the per-shard lists (short, proven exact by the merge:
amdrec.sharded) are exchanged with one RCCL all-to-all, each rank merges and ranks its own 512
users ("scaling": "weak").
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "movie-recommender-demo_amd"))


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-users", type=int, default=192)
    ap.add_argument("--sweep", action="store_true", help="(kept for old command lines: the latency sweep is in the line by default)")
    ap.add_argument("--no-search-sweep", action="store_true",
                    help="skip the search-only B sweep that fills the line's `search.sweep` (< 1 s)")
    ap.add_argument("--corpus", choices=["random", "clustered"], default=None,
                    help="synthetic corpus: randn unit rows (the reference benchmark's, default for flat) or a "
                         "clustered one (default for ivf: an inverted file needs structure to exploit)")
    ap.add_argument("--full-lists", action="store_true",
                    help="N > 1: every shard sends its full top-500 (default: short lists with a proof of exactness)")
    ap.add_argument("--ads", type=int, default=N_ADS, help="corpus size (configs[3]: 10000000)")
    ap.add_argument("--index", choices=["flat", "ivf"], default="flat", help="configs[4]: ivf")
    ap.add_argument("--nlist", type=int, default=4096, help="IVF lists of the coarse quantizer shared by all ranks (whole corpus)")
    ap.add_argument("--nprobe", type=int, default=64, help="IVF probes per query (every rank scans its slice of each probed list)")
    ap.add_argument("--no-strict-fp32", action="store_true",
                    help="skip the second figure (`strict_fp32`: 5 steps with the ranker on the fp32-MFMA engine)")
    ap.add_argument("--ranker-engine", choices=["default", "f16x3", "fp32", "bf16x6"], default="default",
                    help="TransformerRanker.gemm_engine for the WHOLE run (default: the drop-in's default, f16x3); "
                         "`fp32` = the strict fp32-MFMA engine as the headline (profiles/r04_bench_kernel_stats_fp32.csv)")
    ap.add_argument("--user-batches", type=int, default=8,
                    help="distinct seeded user batches rotated through the timed loop (a serving benchmark never sees the "
                         "same users twice in a row); the one-batch replay is reported beside it as `same_batch`")
    ap.add_argument("--no-latency-sweep", action="store_true",
                    help="skip `e2e_latency_by_batch` (one recommend_device call at B = 1 / 8 / 64 / 512, N = 1 only)")
    ap.add_argument("--dry-run", action="store_true",
                    help="form the world (launcher, rendezvous, world-size checks), print a stub line with n_gpus and "
                         "exit without touching a GPU: the CPU test of the N > 1 launch path")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    return args


def launch_ranks(args, argv):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start N ranks of this same script, one per GPU, and wait.
    Runs BEFORE torch is imported: this process never initialises the GPU, and each child is a fresh interpreter
    (subprocess: fork + exec of a process that holds no HIP state), so nothing that owns a GPU context ever execs.
    stdout / stderr are inherited: rank 0's JSON line is this process's JSON line.  Exit code = the first failing
    rank's (the others are terminated by PID), 0 when every rank exits 0."""
    import subprocess
    import tempfile
    # Rendezvous of the self-launched world through a FILE store in a private temporary directory (init_method
    # file://...): probing a free TCP port and closing the socket before the children bind it left a window in which
    # another process could take the port and the run died in init_process_group.  Under torchrun the
    # rendezvous is torchrun's (MASTER_ADDR / MASTER_PORT).
    tmp = tempfile.mkdtemp(prefix="amdrec_bench_rdzv_")
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   AMDREC_BENCH_INIT_FILE=os.path.join(tmp, "store"), AMDREC_BENCH_SELF_LAUNCHED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this host driver
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc = r if r > 0 else 1
                for q in alive:                      # a rank died: the others would wait in a collective for ever
                    q.terminate()
        time.sleep(0.05)
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    return rc


N_ADS = 1_000_000
if __name__ == "__main__":
    _ARGS = parse_args()
    if _ARGS.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(_ARGS, sys.argv[1:]))

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

DIM = 256
USERS_PER_GPU = 512
STAGE1_K = 500
TOP_K = 10
FP32_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: fp32 MFMA == fp32 vector peak
BF16_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA
X6_PEAK_TFLOPS = BF16_PEAK_TFLOPS / 6   # "x6" kernels: 6 bf16 MFMA products per fp32 multiply-add (gemm_core.hpp)
X3_PEAK_TFLOPS = BF16_PEAK_TFLOPS / 3   # "x3" kernel: 3 fp16 MFMA products per fp32 multiply-add (rowowner.hpp); fp16 rate == bf16 rate
HBM_PEAK_GBS = 8000.0
MIN_WARMUP = 20               # untimed steps before the timed region whatever --warmup says (the
                              # power-limited ranker kernel is still settling its clock after 10 launches)


def _t(sd):
    return {k: torch.from_numpy(np.array(v)) for k, v in sd.items()}


def build_models(device, seed=7):
    from amdrec import synth
    from amdrec.ranker import TransformerRanker
    from amdrec.towers import TwoTowerModel
    user, ad, nnum = synth.demo_dims()
    tt_sd = synth.two_tower_state(user, ad, nnum, seed=seed)
    rk_sd = synth.ranker_state(user, ad, nnum, seed=seed + 1, cross_scale=1.0 / 16)
    tt = TwoTowerModel(dict(user), dict(ad), nnum)
    tt.load_state_dict(_t(tt_sd))
    rk = TransformerRanker(dict(user), dict(ad), nnum)
    rk.load_state_dict(_t(rk_sd))
    return tt.to(device).eval(), rk.to(device).eval(), (tt_sd, rk_sd), (user, ad, nnum)


def device_corpus(n, dim, device, seed=1234, row0=0, rows=None, kind="random"):
    """The benchmark corpus (amdrec.devsynth): any shard of the same (n, seed) corpus is bit-identical on every rank."""
    from amdrec import devsynth
    if kind == "clustered":
        return devsynth.device_clustered_corpus(n, dim, device, seed=seed, row0=row0, rows=rows)
    return devsynth.device_corpus(n, dim, device, seed=seed, row0=row0, rows=rows)


def pmc_traffic(tag, run_tags=()):
    """HBM bytes per launch of the kernel family behind a bench tag, from the committed PMC summary
    (profiles/rNN_pmc.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command, gfx950 x2 read correction applied - tools/pmc_summary.py).  None if no summary is committed, or if the
    newest summary was taken on a different set of kernels than the ones this run launched (a stale file must not
    be quoted: the summary lists `bench_tags`, compared with this run's profiling tags)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
    if not files:
        return None
    try:
        doc = json.load(open(files[-1]))
        k = doc["kernels"]
    except Exception:
        return None
    if sorted(doc.get("bench_tags", [])) != sorted(run_tags):
        return None
    epi, shape = tag.rsplit("_", 1)
    fam = [v for name, v in k.items() if name in (tag, f"{epi}_gather_{shape}") and "hbm_bytes_per_launch" in v]
    if not fam:
        return None
    n = sum(v["launches"] for v in fam)
    return {"hbm_bytes_per_launch": round(sum(v["hbm_bytes_per_launch"] * v["launches"] for v in fam) / n),
            "source": os.path.basename(files[-1])}


def _host_threads():
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count()
    try:
        threads = min(threads, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    return int(threads)


def torch_cpu_pipeline(tt_sd, rk_sd, dims, corpus_cpu, ad_table_cpu, uc, un, n_users, user_chunk=32, corpus_chunk=1 << 16):
    """SURVEY.md section 8(d)'s CPU baseline when faiss is absent: the reference's pipeline restated on PyTorch-CPU -
    the drop-in modules' ATen forwards (`autograd_forward`: the reference's own op sequence, two_tower_model.py:33-121,
    transformer_ranker.py:310-380, literal 8-head attention) in eval mode under no_grad for the towers and the ranker,
    IndexFlatIP restated as chunked `Q @ X^T` + `torch.topk` with a running merge (faiss_retrieval.py:146-155), the ranker
    fed user_chunk users x 500 candidate rows per forward (inference.py:241-255 builds 500-row batches; rows are
    independent), top-10 by CTR (inference.py:258-263).  -> (seconds, ad ids [n_users, TOP_K])."""
    from amdrec.ranker import TransformerRanker
    from amdrec.towers import TwoTowerModel
    user, ad, nnum = dims
    tt = TwoTowerModel(dict(user), dict(ad), nnum)
    tt.load_state_dict(_t(tt_sd))
    rk = TransformerRanker(dict(user), dict(ad), nnum)
    rk.load_state_dict(_t(rk_sd))
    tt.eval()
    rk.eval()
    X = torch.from_numpy(corpus_cpu)
    table = torch.from_numpy(ad_table_cpu)
    ucat, unum = torch.from_numpy(uc[:n_users]), torch.from_numpy(un[:n_users])
    # intra-op threads: ATen at one thread per host CPU is oversubscribed on a 128-CPU box (round 3: 26-30 recs/s, below the
    # survey's 8-core probe).  A short untimed calibration - one ranker chunk and one corpus chunk per candidate count - picks
    # the fastest setting for the timed run; the count used is reported as `cores`.
    cores = os.cpu_count() or 1
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    best = (None, float("inf"))
    with torch.no_grad():
        qprobe = torch.randn(min(n_users, 64), X.shape[1])
        nprobe_u = min(n_users, user_chunk)
        rows_p = torch.zeros(nprobe_u * STAGE1_K, dtype=torch.int64)
        _ = qprobe @ X[:corpus_chunk].T                      # first-touch / thread-pool start-up outside the calibration
        for nt in sorted({cores, min(cores, 64), min(cores, 32), min(cores, 16), min(cores, 8)}, reverse=True):
            torch.set_num_threads(nt)
            t0 = time.time()
            _ = torch.topk(qprobe @ X[:corpus_chunk].T, min(STAGE1_K, corpus_chunk), dim=1)
            _ = rk.autograd_forward(ucat[:nprobe_u].repeat_interleave(STAGE1_K, 0), table[rows_p],
                                    unum[:nprobe_u].repeat_interleave(STAGE1_K, 0))
            dt_ = time.time() - t0
            if dt_ < best[1]:
                best = (nt, dt_)
    torch.set_num_threads(best[0])
    t0 = time.time()
    with torch.no_grad():
        q = tt.user_tower.autograd_forward(ucat, unum)
        q = torch.nn.functional.normalize(q, p=2, dim=1)                  # faiss.normalize_L2 on the query copy (:146-147)
        best_s = torch.empty((n_users, 0))
        best_i = torch.empty((n_users, 0), dtype=torch.int64)
        for s in range(0, X.shape[0], corpus_chunk):                       # IndexFlatIP.search (:155)
            sc = q @ X[s:s + corpus_chunk].T
            v, i = torch.topk(sc, min(STAGE1_K, sc.shape[1]), dim=1)
            best_s, best_i = torch.cat([best_s, v], 1), torch.cat([best_i, i + s], 1)
            if best_s.shape[1] > 4 * STAGE1_K:
                v, j = torch.topk(best_s, STAGE1_K, dim=1)
                best_s, best_i = v, torch.gather(best_i, 1, j)
        v, j = torch.topk(best_s, STAGE1_K, dim=1)
        cand = torch.gather(best_i, 1, j)
        ids = torch.empty((n_users, TOP_K), dtype=torch.int64)
        for b0 in range(0, n_users, user_chunk):
            b1 = min(n_users, b0 + user_chunk)
            rows = cand[b0:b1].reshape(-1)
            pred = rk.autograd_forward(ucat[b0:b1].repeat_interleave(STAGE1_K, 0), table[rows],
                                       unum[b0:b1].repeat_interleave(STAGE1_K, 0))
            ctr = torch.sigmoid(pred["ctr"]).view(b1 - b0, STAGE1_K)
            _ = torch.sigmoid(pred["engagement"]), torch.sigmoid(pred["revenue"])
            top = torch.topk(ctr, TOP_K, dim=1).indices
            ids[b0:b1] = torch.gather(cand[b0:b1], 1, top)
    return time.time() - t0, ids.numpy()


def cpu_baseline(tt_sd, rk_sd, dims, corpus_cpu, ad_table_cpu, uc, un, n_users, user_chunk=32, torch_leg=True):
    """The reference's CPU path timed on this host's cores on a bounded sample of the same workload: the FIRST n_users
    users of the GPU step's own batch against the full corpus, twice:
      * the PyTorch-CPU restatement (torch_cpu_pipeline) - the reported baseline ("faiss unavailable
        - torch-CPU restatement"; `import faiss` is attempted, and IndexFlatIP timed through it if it is there);
      * the numpy oracle (oracle/), whose output doubles as the full-size parity check of the GPU step (parity_check) -
        the checker, reported beside it.
    The ranker legs are fed user_chunk users x 500 candidate rows per forward (the reference loops 500-row forwards,
    inference.py:310-317: the 8-core survey probe of that form gave 38 recs/s)."""
    import oracle
    idx = oracle.search.FlatIndex(DIM)
    idx.add(corpus_cpu)                                   # index build is not timed (nor is it on the GPU)

    def numpy_leg_run(limit=None):
        t0 = time.time()
        if limit:
            from threadpoolctl import threadpool_limits
            with threadpool_limits(limits=int(limit)):
                r = oracle.pipeline.recommend(tt_sd, rk_sd, idx, ad_table_cpu, uc[:n_users], un[:n_users], TOP_K, STAGE1_K,
                                              user_chunk=user_chunk)
        else:
            r = oracle.pipeline.recommend(tt_sd, rk_sd, idx, ad_table_cpu, uc[:n_users], un[:n_users], TOP_K, STAGE1_K,
                                          user_chunk=user_chunk)
        dt_ = time.time() - t0
        return r, {"value": round(n_users / dt_, 2), "unit": "recs/s", "cores": int(limit) if limit else _host_threads(),
                   "seconds": round(dt_, 1), "what": "oracle/ (numpy fp32 BLAS + per-query sort): the parity checker"}

    if not torch_leg:
        ref, numpy_leg = numpy_leg_run()
        return dict(numpy_leg, kind="port", sample=f"the first {n_users} users x {len(corpus_cpu)} ads through oracle/"), ref
    try:
        import faiss  # noqa: F401
        have_faiss = True
    except Exception:
        have_faiss = False
    faiss_leg = None
    if have_faiss:                                        # never the case on this image; kept so that a box with faiss uses it
        import faiss
        fi = faiss.IndexFlatIP(DIM)
        fi.add(corpus_cpu)
        emb = oracle.search.normalize_l2(oracle.towers.user_tower(tt_sd, uc[:n_users], un[:n_users]))
        t0 = time.time()
        fi.search(emb, STAGE1_K)
        faiss_leg = {"search_seconds": round(time.time() - t0, 3), "threads": faiss.omp_get_max_threads()}
    tdt, tids = torch_cpu_pipeline(tt_sd, rk_sd, dims, corpus_cpu, ad_table_cpu, uc, un, n_users, user_chunk)
    try:                                                  # the numpy oracle on the thread count the calibration picked for ATen
        ref, numpy_leg = numpy_leg_run(torch.get_num_threads())
    except Exception:
        ref, numpy_leg = numpy_leg_run()
    same = float(np.mean([set(tids[b].tolist()) == set(ref[b]["ad_ids"]) for b in range(n_users)]))
    tthreads = torch.get_num_threads()
    return {"value": round(n_users / tdt, 2), "unit": "recs/s", "cores": int(tthreads), "kind": "port",
            "sample": f"the first {n_users} users of the timed batch x {len(corpus_cpu)} ads end-to-end on PyTorch-CPU "
                      f"(torch.get_num_threads() = {tthreads} of {os.cpu_count()} host CPUs): the drop-in modules' ATen "
                      f"forwards (the reference's op sequence, eval mode, no_grad), IndexFlatIP restated as chunked Q@X^T + "
                      f"torch.topk, ranker fed {user_chunk} users x {STAGE1_K} rows per forward, {tdt:.1f}s; faiss "
                      + ("timed separately" if have_faiss else "unavailable - torch-CPU restatement"),
            "top10_equal_to_numpy_oracle_frac": round(same, 4), "faiss": faiss_leg, "numpy_oracle": numpy_leg}, ref


def parity_check(ref, out, n_users):
    """Full-size end-to-end check of the timed GPU step against the oracle run of cpu_baseline (same users, same
    corpus): stage-1 candidate sets (tolerance-aware), ranker logits on the common candidates
    against the STRICT rule |d| <= 1e-4 * max(1, |logit|), and the final top-10."""
    import oracle
    cand = out["candidate_ids"][:n_users].cpu().numpy()
    cs = out["candidate_scores"][:n_users].cpu().numpy()
    B_all = out["candidate_ids"].shape[0]
    logits = out["logits"].cpu().numpy().reshape(3, B_all, STAGE1_K)
    ids = out["ad_ids"][:n_users].cpu().numpy()
    topk_ok, worst, top10_same, top10_ok = True, 0.0, 0, True
    for b in range(n_users):
        try:
            oracle.search.check_topk(ref[b]["candidate_scores"][None], ref[b]["candidate_ids"][None], cs[b][None],
                                     cand[b][None], tau=1e-5, score_tol=1e-6)
        except AssertionError:
            topk_ok = False
        pos = {int(i): j for j, i in enumerate(ref[b]["candidate_ids"])}
        common = [j for j, i in enumerate(cand[b]) if int(i) in pos]
        sel = np.array([pos[int(cand[b][j])] for j in common])
        for ti, t in enumerate(oracle.ranker.TASKS):
            r = ref[b]["logits"][t][sel].astype(np.float64)
            err = np.abs(logits[ti, b][common].astype(np.float64) - r)
            worst = max(worst, float((err / (1e-4 * np.maximum(1.0, np.abs(r)))).max()))
        # top-10: the GPU's winners must be the winners of its OWN logits (selection exact), and equal the oracle's
        # unless a winner sits within the logit tolerance of the 10th place
        own = cand[b][oracle.pipeline.select_top(logits[0, b], TOP_K)]
        top10_ok = top10_ok and bool(np.array_equal(own, ids[b]))
        top10_same += int(set(ids[b].tolist()) == set(ref[b]["ad_ids"]))
    return {"users": n_users, "topk_set_ok": topk_ok, "max_logit_err_over_bound": round(worst, 5),
            "logit_bound": "1e-4*max(1,|logit|) (strict SURVEY 8a rule, no batch-scale escape)",
            "top10_selection_exact": top10_ok, "top10_equal_to_oracle_frac": round(top10_same / max(n_users, 1), 4)}


def _rendezvous():
    """init_process_group keywords: the self-launched world meets through the launcher's file store, torchrun's through
    MASTER_ADDR / MASTER_PORT (env://)."""
    f = os.environ.get("AMDREC_BENCH_INIT_FILE")
    if f:
        return {"init_method": "file://" + f}
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    return {}


def roofline_block(name, p, run_tags, steps):
    """The `roofline` object of one kernel tag from the library's per-launch HIP events (p = profile_report()[name] of a
    timed region that recorded events around this tag only): achieved = algorithmic FLOPs per launch / average launch
    time, against the peak of the engine the tag names (x3: fp16 MFMA / 3 products, x6: bf16 MFMA / 6, else fp32 MFMA)."""
    avg_ms = p["total_ms"] / p["launches"]
    achieved = p["flops"] / p["launches"] / (avg_ms * 1e-3) / 1e12
    tr = pmc_traffic(name, run_tags)
    x6 = name.endswith("_x6")       # fp32 GEMM computed as 6 bf16-MFMA products per MAC (exact 3-way split)
    x3 = name.endswith("_x3")       # fp32 chain computed as 3 fp16-MFMA products per MAC (two-plane split)
    peak = X3_PEAK_TFLOPS if x3 else (X6_PEAK_TFLOPS if x6 else FP32_PEAK_TFLOPS)
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
            "peak_note": ("fp32-equivalent FLOPs (2*M*N*K of every GEMM of the chain) against dense fp16 MFMA peak "
                          f"/ 3 products = executed-MFMA utilisation; the fp32 MFMA peak is {FP32_PEAK_TFLOPS}; "
                          "the peak is priced at 2.4 GHz - on realistic operands the chip holds ~2.0 GHz on this kernel "
                          "(power-limited: profiles/r02_x3_cycle_stamps_and_dvfs.log), where the bound is ~694") if x3
            else (("fp32-equivalent FLOPs (2*M*N*K) against dense bf16 MFMA peak / 6 products; "
                   f"the fp32 MFMA peak is {FP32_PEAK_TFLOPS}") if x6 else "2*M*N*K against the fp32 MFMA peak"),
            "traffic": tr["hbm_bytes_per_launch"] if tr else None,
            "traffic_source": tr["source"] if tr else None,
            "traffic_note": ("L2 memory-side requests (FETCH_SIZE x 2 + WRITE_SIZE; Infinity-Cache hits are counted): the "
                             "algorithmic bytes + the 8.6 MB weight stream once per XCD L2 and round of workgroups (it does "
                             "not fit a 4 MB L2): 8.6 MB x 8 x 7.8 = 0.54 GB, served on-die") if x3 and tr else None,
            "alg_bytes_per_launch": round(p["bytes"] / p["launches"]),
            "kernel": name, "launches_per_step": p["launches"] / steps,
            "avg_launch_ms": round(avg_ms, 4),
            "alg_gflop_per_launch": round(p["flops"] / p["launches"] / 1e9, 3)}


def ranker_family(prof, steps):
    """The ranker's GEMM tags of a fully event-timed pre-pass (strict fp32 engine: linear_* / residual_ln_* / cross_*):
    per tag ms per step, TFLOP/s and the fraction of the fp32-MFMA peak, plus all of them together."""
    fam = {k: v for k, v in prof.items() if k.startswith(("linear_", "residual_ln_", "cross_")) and v["total_ms"]}
    out = {k: {"ms_per_step": round(v["total_ms"] / steps, 3), "launches_per_step": v["launches"] / steps,
               "tflops": round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12, 2),
               "frac_of_fp32_mfma_peak": round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)}
           for k, v in sorted(fam.items())}
    if fam:
        ms, fl = sum(v["total_ms"] for v in fam.values()), sum(v["flops"] for v in fam.values())
        out["all"] = {"ms_per_step": round(ms / steps, 3), "tflops": round(fl / (ms * 1e-3) / 1e12, 2),
                      "frac_of_fp32_mfma_peak": round(fl / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4)}
    return out


def main():
    args = parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        # never report a line whose n_gpus differs from what was asked for (a missing WORLD_SIZE with --gpus N > 1 is
        # handled by launch_ranks before this point)
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}: refusing to run")
    # Rehearsal switch (not used by the driver): AMDREC_BENCH_REHEARSE=1 runs all ranks on cuda:0 over
    # gloo, so the N>1 code path can be exercised on a one-GPU box.  The numbers it prints are meaningless.
    rehearse = os.environ.get("AMDREC_BENCH_REHEARSE") == "1"
    backend = "gloo" if (rehearse or args.dry_run) else "nccl"
    if args.dry_run:
        # launch-path check only: rendezvous + world-size agreement, no GPU (tests/test_bench_launch.py)
        if os.environ.get("AMDREC_BENCH_FAIL_RANK") == str(rank):       # test hook: a rank that dies before the rendezvous
            raise SystemExit(f"bench.py: rank {rank} failing on request (AMDREC_BENCH_FAIL_RANK)")
        if world > 1:
            dist.init_process_group(backend, rank=rank, world_size=world, **_rendezvous())
            seen = torch.ones(1, dtype=torch.int64)
            dist.all_reduce(seen)
            if int(seen.item()) != args.gpus or dist.get_world_size() != args.gpus:
                raise SystemExit(f"bench.py: world formed with {int(seen.item())} ranks, --gpus {args.gpus}")
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"metric": "dry run (launch path only, nothing measured)", "value": None, "n_gpus": world,
                              "dry_run": True, "config": {"backend": backend, "world": world}}), flush=True)
        return
    if rehearse:
        local_rank = 0
    elif torch.cuda.device_count() < world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {torch.cuda.device_count()} GPU(s) are visible "
                         "(AMDREC_BENCH_REHEARSE=1 runs all ranks on cuda:0 over gloo to rehearse the code path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world, **_rendezvous())
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device, **_rendezvous())
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"bench.py: process group has {dist.get_world_size()} ranks, --gpus {args.gpus}")

    from amdrec import _lib, synth
    from amdrec.index import FAISSIndex
    from amdrec.pipeline import AdRecommenderInference
    _lib.load()

    n_ads = args.ads
    tt, rk, (tt_sd, rk_sd), dims = build_models(device)
    user, ad, nnum = dims
    per = (n_ads + world - 1) // world
    row0, rows = rank * per, max(0, min(per, n_ads - rank * per))
    corpus_kind = args.corpus or ("clustered" if args.index == "ivf" else "random")
    shard = device_corpus(n_ads, DIM, device, row0=row0, rows=rows, kind=corpus_kind)
    if args.index == "ivf":
        # ONE coarse quantizer for the whole corpus (nlist lists, nprobe probes), shared by all ranks: rank 0 trains on
        # its shard and broadcasts the centroids, every rank files its rows under them (its slice of every list), so
        # the merged result is exactly the unsharded IVF result (amdrec.sharded)
        from amdrec.sharded import share_ivf_centroids
        index = FAISSIndex(DIM, index_type="IVF", nlist=args.nlist, nprobe=args.nprobe, device=device)
        share_ivf_centroids(index, shard, rank, world)
    else:
        index = FAISSIndex(DIM, index_type="Flat", device=device)
    index.add(shard)                                        # renormalises + stores (faiss_retrieval.py:97-127)
    flat_ref = None
    if args.index == "ivf":
        flat_ref = FAISSIndex(DIM, index_type="Flat", device=device)     # for recall@500 vs Flat (stderr)
        flat_ref.add(shard)
    del shard
    ad_table = torch.from_numpy(synth.ad_features(ad, n_ads, seed=99)).to(device)   # replicated (160 MB / 1M)
    rec = AdRecommenderInference(two_tower_model=tt, transformer_ranker=rk, faiss_index=index,
                                 ad_features=ad_table)
    if args.ranker_engine != "default":
        rk.gemm_engine = args.ranker_engine
    B_global = USERS_PER_GPU * world
    # args.user_batches distinct seeded batches rotate through every loop below; batch 0 (seed 2024) is the one the CPU legs
    # and the parity check use, and the rotation is arranged so that the LAST timed step runs batch 0: the output that is
    # checked against the oracle is the timed region's own.
    NB = max(1, args.user_batches)
    batches_np = [synth.user_batch(user, nnum, B_global, seed=2024 + 7919 * j) for j in range(NB)]
    batches = [(torch.from_numpy(a).to(device), torch.from_numpy(b).to(device)) for a, b in batches_np]
    uc_np, un_np = batches_np[0]
    uc, un = batches[0]

    if world > 1:
        from amdrec.sharded import ShardedRecommender, StageTimer
        runner = ShardedRecommender(rec, rank, world, shard_offset=row0, shard_k=None if args.full_lists else "auto")
        # The drop-in's DEFAULT mode (verify=True): with short per-shard lists every step checks the merge's proof of
        # exactness (one 4-byte all-reduce + a host read) before it returns and repeats an unproven batch with full lists,
        # all inside the timed region.  The unverified mode (the caller checks inexact_count() once per reporting
        # interval) is timed separately below and reported as `no_verify`.
        run = lambda c, n, **kw: runner.recommend_device(c, n, TOP_K, STAGE1_K, **kw)      # noqa: E731
    else:
        run = lambda c, n, **kw: rec.recommend_device(c, n, TOP_K, STAGE1_K, **kw)         # noqa: E731

    def rotating(steps):
        """step i of `steps` -> batch (steps - 1 - i) % NB: distinct users step after step, batch 0 last"""
        return lambda i, **kw: run(*batches[(steps - 1 - i) % NB], **kw)

    same = lambda i, **kw: run(uc, un, **kw)                                               # noqa: E731

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    n_warm = max(args.warmup, MIN_WARMUP)
    warm = rotating(n_warm)
    for i in range(n_warm):
        warm(i)
    barrier()
    # Per-kernel table: HIP events around EVERY tagged launch, in a pre-pass outside the timed region.  An event pair
    # costs the stream ~10 us of idle GPU per launch (kernel trace: 0.2 us between untimed kernels, 5-10 us around timed
    # ones), ~0.1 ms over the ~10 tagged launches of a step, so the timed region times only the dominant kernel.
    n_pre = max(1, min(5, args.steps))

    def prepass(fn, n):
        _lib.profile_enable(True)
        for i in range(n):
            fn(i)
        torch.cuda.synchronize(device)
        rep = _lib.profile_report()
        _lib.profile_enable(False)
        return rep, (max(rep.items(), key=lambda kv: kv[1]["total_ms"])[0] if rep else "")

    prof_all, dom_tag = prepass(rotating(n_pre), n_pre)
    stats0 = runner.short_list_stats() if world > 1 else None

    def timed_region(fn, steps=None, only=None):
        steps = args.steps if steps is None else steps
        barrier()
        _lib.profile_enable(True, only=dom_tag if only is None else only)  # HIP events around ONE kernel's launches, on the launch stream
        # per-step HIP events on the launch stream (torch's current stream IS the stream every kernel is enqueued on,
        # _lib.stream_ptr): median / p95 of the step time
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        for i in range(steps):
            marks[i].record()
            res = fn(i)
        marks[steps].record()
        barrier()
        dt_ = time.perf_counter() - t0
        ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
        prof_ = _lib.profile_report()
        _lib.profile_enable(False)
        spread = None
        if world > 1:
            cdev = "cpu" if rehearse else device
            tmax = torch.tensor([dt_], dtype=torch.float64, device=cdev)
            tmin = tmax.clone()
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(tmin, op=dist.ReduceOp.MIN)
            spread = (float(tmin.item()), float(tmax.item()))
            dt_ = float(tmax.item())
        return dt_, ms, prof_, res, spread

    dt, step_ms, prof, out, spread = timed_region(rotating(args.steps))
    # the same region replaying ONE batch (rounds 1-3 timed this): every step gathers the same 256 000 rows of the ad
    # projection cache and re-scores the same corpus rows, next to a 256 MB Infinity Cache
    dt_same, _, _, _, _ = timed_region(same)
    same_batch = {"value": round(B_global * args.steps / dt_same, 1), "ms_per_step": round(dt_same / args.steps * 1000, 3),
                  "note": "one user batch replayed every step (what rounds 1-3 reported as `value`)"}
    short_lists, no_verify, exchange, multi = None, None, None, None
    if world > 1:
        st = runner.short_list_stats()
        exchange = dict(runner.last_exchange)
        short_lists = {"list_k": runner.list_k(STAGE1_K), "k": STAGE1_K, "list_k_timed": exchange["list_k"],
                       "verified_every_step": True,
                       "inexact_in_timed_steps": st["unproven_queries"] - stats0["unproven_queries"],
                       "batches_repeated_with_full_lists_in_timed_steps": st["repeated_batches"] - stats0["repeated_batches"],
                       "hit_rate": st["hit_rate"], "switched_off": st["switched_off"]}
        if runner.list_k(STAGE1_K) < STAGE1_K:
            # the same steps without the per-step check; a number is only reported if the proof held for every one of them
            runner.inexact_count()
            rot = rotating(args.steps)
            dt2, ms2, _, _, _ = timed_region(lambda i: rot(i, verify=False))
            bad = runner.inexact_count()                     # collective: the same value on every rank
            no_verify = {"value": round(B_global * args.steps / dt2, 1) if not bad else None,
                         "ms_per_step": round(dt2 / args.steps * 1000, 3), "unproven_queries": bad,
                         "note": "recommend_device(verify=False): the proof counter is read once after the region, not per step"}
        # Self-diagnosis of the multi-GPU step (the driver's 8-GPU run is the first one on hardware): rank 0's per-stage
        # times from a few EXTRA steps with HIP events at the stage boundaries (outside every timed region: an event
        # pair idles the stream ~10 us), the slowest / fastest rank's wall time over the timed region, and what
        # torch.distributed says the world is.
        n_diag = max(1, min(5, args.steps))
        runner.timer = StageTimer(device)
        rot = rotating(n_diag)
        for i in range(n_diag):
            rot(i)
        stages = runner.timer.report()
        runner.timer = None
        barrier()
        multi = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                 "step_ms_max_over_ranks": round(spread[1] / args.steps * 1000, 3),
                 "step_ms_min_over_ranks": round(spread[0] / args.steps * 1000, 3),
                 "stages_ms_rank0": stages, "stages_steps": n_diag,
                 "stages_note": "mean ms per step on rank 0, HIP events on the launch stream at the stage boundaries of "
                                "ShardedRecommender._step (tower, search, pack, exchange = the one collective, merge, "
                                "ranker, select); proof_wait = host wall time blocked on the proof's all-reduce while the "
                                "ranker runs; measured in extra steps after the timed region",
                 "verified": {"value": round(B_global * args.steps / dt, 1), "ms_per_step": round(dt / args.steps * 1000, 3)},
                 "unverified": no_verify}
    assert out["ad_ids"].shape[-1] == TOP_K

    # The same step with the ranker on the STRICT fp32-MFMA engine (v_mfma_f32_32x32x2_f32: bitwise an fp32 fma chain, the
    # reference's arithmetic, transformer_ranker.py:355-378) - a second, labelled figure so that the headline's emulated
    # fp32 (engine f16x3) is transparent: args.steps timed steps (rotating batches) after its own warm-up, with its own
    # roofline block (the fp32 engine's dominant kernel against the 157.3 TF fp32-MFMA peak) and per-kernel table.
    strict = None
    if not args.no_strict_fp32 and rk.gemm_engine != "fp32":
        eng0 = rk.gemm_engine
        rows_step = USERS_PER_GPU * STAGE1_K
        rk.gemm_engine = "fp32"
        try:
            for i in range(3):
                warm(i)
            n_pre_s = max(1, min(3, args.steps))
            prof_s, dom_s = prepass(rotating(n_pre_s), n_pre_s)
            dts, mss, prof_dom_s, _, _ = timed_region(rotating(args.steps), only=dom_s)
            strict = {"value": round(B_global * args.steps / dts, 1), "unit": "recs/s", "ms_per_step": round(dts / args.steps * 1000, 3),
                      "step_ms_median": round(mss[len(mss) // 2], 3),
                      "steps": args.steps, "engine": rk.gemm_engine_for(rows_step), "dtype": "f32",
                      "roofline": roofline_block(dom_s, prof_dom_s.get(dom_s), list(prof_s), args.steps) if dom_s in prof_dom_s else None,
                      "ranker_kernels": ranker_family(prof_s, n_pre_s),
                      "note": "same step, TransformerRanker.gemm_engine = 'fp32': every ranker GEMM on the fp32 MFMA "
                              "(exact fp32 products, fp32 accumulate); search and towers unchanged"}
        finally:
            rk.gemm_engine = eng0
        for i in range(3):                              # re-pack the default engine (and its caches) before anything else
            warm(i)
        torch.cuda.synchronize(device)

    if rank == 0:
        ms_step = dt / args.steps * 1000
        value = B_global * args.steps / dt
        # dominant kernel = the tag with the largest total time in the pre-pass; its launches were timed in the region
        roofline = roofline_block(dom_tag, prof.get(dom_tag), list(prof_all), args.steps) if prof.get(dom_tag) else None
        # the other kernels: from the n_pre fully timed steps that ran before the timed region
        kernels = {k: {"ms_per_step": round(v["total_ms"] / n_pre, 3),
                       "tflops": round(v["flops"] / (v["total_ms"] * 1e-3) / 1e12, 2) if v["total_ms"] else None,
                       "launches_per_step": v["launches"] / n_pre} for k, v in sorted(prof_all.items())}
        sf = next((v for k, v in prof_all.items() if k.startswith("search_filter")), None)
        search = None
        if args.index == "ivf":
            search = ivf_search_stats(index, flat_ref, tt, uc, un, prof_all, n_pre)
        elif sf:
            ms = sf["total_ms"] / sf["launches"]
            mixed = getattr(index, "_mixed", False)      # bf16 shadow corpus read by the filter pass
            eb = 2 if mixed else 4
            alg_bytes = rows * DIM * eb + B_global * DIM * eb       # this rank's shard, one corpus pass
            search = {"engine": "bf16 MFMA filter + fp32 re-score + certificate" if mixed else "fp32 MFMA filter",
                      "filter_pass_ms": round(ms, 3), "alg_GBps": round(alg_bytes / (ms * 1e-3) / 1e9, 1),
                      "hbm_frac": round(alg_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                      "tflops": round(sf["flops"] / sf["launches"] / (ms * 1e-3) / 1e12, 2),
                      "note": ("filter = one pass over the bf16 corpus; B=512 queries per pass, intensity 512 FLOP/B vs "
                               "bf16 ridge 312: compute-side bound (vector issue + bf16 MFMA), see DESIGN.md section 6" if mixed else
                               "B=512 per corpus pass is fp32-FLOP bound (intensity 255 FLOP/B vs ridge 19.7)")}
        if search is not None and not args.no_search_sweep and world == 1 and args.index == "flat":
            search["sweep"] = search_sweep(index, device)
        # one recommend_device call by batch size - before the CPU legs: ~15 s of idle GPU let the clocks drop, and a
        # 0.45 ms call measured right after them read 1.5 ms
        latency = latency_sweep(rec, uc, un, device) if world == 1 and not args.no_latency_sweep else None
        cpu, parity = None, None
        if not args.no_cpu_baseline and world == 1:
            corpus_cpu = index._xb[:index._n].cpu().numpy()
            n_cpu = min(args.cpu_users, B_global)
            cpu, ref = cpu_baseline(tt_sd, rk_sd, dims, corpus_cpu, ad_table.cpu().numpy(), uc_np, un_np, n_cpu)
            if args.index == "flat":            # the oracle index is exact: only the exact engine is comparable
                parity = parity_check(ref, out, n_cpu)
        default_cfg = n_ads == N_ADS and args.index == "flat"
        eng_step = rk.gemm_engine_for(USERS_PER_GPU * STAGE1_K)
        arithmetic = ("fp32 in / fp32 out everywhere; ranker (gemm_engine = " + rk.gemm_engine + f", x3_min_rows = {rk.x3_min_rows}: "
                      f"the {USERS_PER_GPU * STAGE1_K}-row pass of a step runs engine {eng_step}, a single request's "
                      f"{STAGE1_K}-row pass engine {rk.gemm_engine_for(STAGE1_K)}): "
                      + ("operands scaled by powers of two and split into 2 fp16 planes, 3 fp16-MFMA products per MAC, fp32 "
                         "accumulate - an EMULATION of fp32: each operand keeps 22 significand bits (fp32: 24); error vs float64 "
                         "= 2-3x the fp32-MFMA engine's on the LOGITS and up to 5-6x its worst element mid-chain "
                         "(profiles/r04_accuracy.json, tests/test_x3_gpu.py); the strict fp32-MFMA figure is `strict_fp32`); "
                         if eng_step == "f16x3" else
                         ("3 bf16 planes, 6 bf16-MFMA products per MAC, fp32 accumulate; " if eng_step == "bf16x6" else
                          "fp32 MFMA; "))
                      + "search: bf16-MFMA prefilter, fp32 re-score, certified exact; everything else fp32 MFMA / fp32 VALU")
        line = {"metric": "end-to-end recs/sec (1M ads d=256, top-500->10)" if default_cfg else
                          f"end-to-end recs/sec ({n_ads} ads d=256, {args.index}, top-500->10)", "value": round(value, 1),
                "unit": "recs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "warmup_steps_run": n_warm + n_pre, "user_batches": NB, "same_batch": same_batch,
                "ms_per_step": round(ms_step, 3),
                "step_ms_median": round(step_ms[len(step_ms) // 2], 3),
                "step_ms_p95": round(step_ms[min(len(step_ms) - 1, int(np.ceil(0.95 * len(step_ms))) - 1)], 3),
                "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None,
                # what the dominant stage computes in: fp32 operands and results, products on the fp16 / bf16 matrix pipe
                "dtype": {"f16x3": "f32 (f16x3 emulation: 2 fp16 planes per operand, 3 fp16-MFMA products per MAC, fp32 accumulate)",
                          "bf16x6": "f32 (bf16x6 emulation: 3 bf16 planes per operand, 6 bf16-MFMA products per MAC, fp32 accumulate)",
                          }.get(eng_step, "f32"),
                "compute_dtype": {"f16x3": "f16x3", "bf16x6": "bf16x6"}.get(eng_step, "f32"),
                "data": "synthetic", "arithmetic": arithmetic,
                "config": {"workload": ("configs[2]: 1M synthetic ads d=256, UserTower batch=512/GPU, "
                                        "exact IP top-500, TransformerRanker(256,8 heads,3 layers) on 500 cands, top-10")
                           if default_cfg else f"{n_ads} ads ({corpus_kind} corpus), index={args.index}"
                           + (f" nlist={args.nlist} nprobe={args.nprobe}" if args.index == "ivf" else ""),
                           "n_ads": n_ads, "dim": DIM, "users_per_step": B_global, "stage1_k": STAGE1_K,
                           "top_k": TOP_K, "corpus_rows_per_gpu": rows,
                           "parallelism": f"corpus row-sharded x{world}, ranker data-parallel over users",
                           "backend": None if world == 1 else ("gloo" if rehearse else "nccl"),   # torch "nccl" == RCCL on ROCm
                           "rehearsal_all_ranks_on_one_gpu": bool(rehearse and world > 1),
                           "world": dist.get_world_size() if world > 1 else 1,
                           "exchange": exchange["kind"] if exchange else None,
                           "bytes_per_rank": exchange["bytes_per_rank"] if exchange else None,
                           "shard_lists": short_lists, "no_verify": no_verify},
                "strict_fp32": strict, "roofline": roofline, "cpu_baseline": cpu, "parity_check": parity, "kernels": kernels,
                "kernels_note": (f"per-kernel table from {n_pre} steps with HIP events around every tagged launch, run before "
                                 "the timed region; the timed region records events around the dominant kernel only "
                                 "(roofline.avg_launch_ms): an event pair costs the stream ~10 us of idle GPU per launch"),
                "search": search}
        if latency is not None:
            line["e2e_latency_by_batch"] = latency
        if multi is not None:                   # N > 1 only: the N = 1 line carries no such key
            line["multi_gpu"] = multi
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def ivf_search_stats(index, flat_ref, tt, uc, un, prof, steps):
    """IVF arm: recall@500 against the exact Flat result on this rank's rows, and the scan priced against HBM with the
    bytes it must read: every probed list once per 64-query group (1 KiB per row), from the probes of the timed batch."""
    from amdrec.index import flat_search
    emb = tt.get_user_embeddings(uc, un)
    qn = emb / emb.norm(dim=1, keepdim=True).clamp_min(1e-30)
    st = index._ivf
    nq, nprobe = qn.shape[0], index.nprobe
    cs = torch.empty((nq, nprobe), dtype=torch.float32, device=qn.device)
    probes = torch.empty((nq, nprobe), dtype=torch.int64, device=qn.device)
    flat_search(st.centroids, st.nlist, qn.contiguous(), nprobe, cs, probes)
    _, _, _, lens, max_len, _ = st._build_lists(index._xb, index._n)
    cnt = torch.bincount(probes[probes >= 0].reshape(-1), minlength=st.nlist)
    scan_rows = int(((cnt + 63) // 64 * lens).sum().item())
    per_query = float(lens[probes.clamp_min(0)].sum(1).float().mean().item())
    out = {"engine": f"IVF-Flat nlist={index.nlist} nprobe={nprobe} (coarse fp32-MFMA top-{nprobe} over the centroids, "
                     "grouped fp32-MFMA list scan, exact k-select)",
           "rows_scanned_per_query": round(per_query, 1), "scan_alg_bytes_per_step": scan_rows * DIM * 4,
           "list_len_min_mean_max": [int(lens.min().item()), round(float(lens.float().mean().item()), 1), int(max_len)]}
    scans = {k: v for k, v in prof.items() if k.startswith("ivf_scan") and v["total_ms"]}
    if scans:
        # every scan launch of the step (the two-phase scan has two; its second runs on the bf16 shadow of the lists when the
        # first phase is selective enough - half the bytes per row - and re-scores the nominated rows from the fp32 lists)
        ms = sum(v["total_ms"] for v in scans.values()) / steps
        from amdrec.ivf import first_phase_probes
        n_first = first_phase_probes(nprobe)
        two_phase = len(scans) > 1
        alg = scan_rows * DIM * 4
        if two_phase:
            def tiled_rows(pr, qt):
                c = torch.bincount(pr[pr >= 0].reshape(-1), minlength=st.nlist)
                return int(((c + qt - 1) // qt * lens).sum().item())
            mixed = any("bf16" in k for k in scans)
            alg = tiled_rows(probes[:, :n_first], 64) * DIM * 4 + tiled_rows(probes[:, n_first:], 64) * DIM * (2 if mixed else 4)
            out["scan_alg_bytes_per_step"] = alg
            out["scan_phases"] = {k: round(v["total_ms"] / steps, 4) for k, v in sorted(scans.items())}
        out["scan_ms"] = round(ms, 3)
        out["scan_alg_GBps"] = round(alg / ms / 1e6, 1)
        out["scan_hbm_frac"] = round(alg / ms / 1e6 / HBM_PEAK_GBS, 4)
    if flat_ref is not None:
        a, _ = index.search_device(emb, STAGE1_K)
        b, _ = flat_ref.search_device(emb, STAGE1_K)
        a, b = a.cpu().numpy(), b.cpu().numpy()
        out["recall_at_500_vs_flat"] = round(float(np.mean([len(set(x) & set(y)) / STAGE1_K for x, y in zip(a, b)])), 4)
    return out


def latency_sweep(rec, uc, un, device, reps=30):
    """End-to-end latency of one recommend_device call (device-resident inputs) by batch size; B = 1 is what
    the reference's recommend_ads serves (it claims "<100 ms", README.md:193)."""
    rows = []
    for B in (1, 8, 64, 512):
        for _ in range(10):
            rec.recommend_device(uc[:B], un[:B], TOP_K, STAGE1_K)
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            rec.recommend_device(uc[:B], un[:B], TOP_K, STAGE1_K)
        e1.record()
        torch.cuda.synchronize(device)
        ms = e0.elapsed_time(e1) / reps
        row = {"B": B, "ms_per_call": round(ms, 3), "recs_per_s": round(B / ms * 1e3, 1)}
        if B <= 64:                                  # hipGraph replay of the same call
            g = rec.capture(B, TOP_K, STAGE1_K)
            for _ in range(3):
                g(uc[:B], un[:B])
            torch.cuda.synchronize(device)
            e0.record()
            for _ in range(reps):
                g(uc[:B], un[:B])
            e1.record()
            torch.cuda.synchronize(device)
            row["ms_per_call_hipgraph"] = round(e0.elapsed_time(e1) / reps, 3)
        rows.append(row)
    return rows


def search_sweep(index, device, reps=20):
    """Search-only sweep over queries per corpus pass: achieved algorithmic HBM GB/s and
    fp32-equivalent TFLOP/s of the WHOLE search call (sample + threshold + corpus pass + finalize), HIP events on the
    launch stream; plus the corpus pass alone from the library's per-launch events (`pass_ms`).  B <= 32 is where HBM
    binds (bf16 shadow: ridge ~300 queries per pass), B = 512 is the benchmark's batch."""
    from amdrec import _lib
    n = index._n
    g = torch.Generator(device=device)
    g.manual_seed(5)
    q = torch.randn((512, DIM), generator=g, device=device)
    q = q / q.norm(dim=1, keepdim=True)
    eb = 2 if getattr(index, "_mixed", False) else 4
    rows = []
    for B in (1, 8, 32, 128, 512):
        qq = q[:B].contiguous()
        for _ in range(3):
            index.search_device(qq, STAGE1_K, normalize=False)
        torch.cuda.synchronize(device)
        # whole call: no per-launch events inside (an event pair idles the stream ~10 us: five tags would add a quarter
        # to a 0.16 ms call); the per-kernel breakdown comes from a second, event-instrumented loop
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            index.search_device(qq, STAGE1_K, normalize=False)
        e1.record()
        torch.cuda.synchronize(device)
        ms = e0.elapsed_time(e1) / reps
        _lib.profile_enable(True)
        for _ in range(reps):
            index.search_device(qq, STAGE1_K, normalize=False)
        torch.cuda.synchronize(device)
        prof = _lib.profile_report()
        _lib.profile_enable(False)
        # one corpus pass (bf16 shadow when the index runs the mixed search) + the fp32 rows of the k results
        alg_bytes = n * DIM * eb + B * DIM * 4 + B * STAGE1_K * 12 + (B * STAGE1_K * DIM * 4 if eb == 2 else 0)
        flops = 2.0 * B * n * DIM
        row = {"B": B, "ms": round(ms, 4), "qps": round(B / ms * 1e3, 1),
               "alg_GBps": round(alg_bytes / ms / 1e6, 1), "hbm_frac": round(alg_bytes / ms / 1e6 / HBM_PEAK_GBS, 4),
               "tflops": round(flops / ms / 1e9, 2)}
        for tag, v in prof.items():
            if tag.startswith("search_filter") and v["launches"]:
                pm = v["total_ms"] / v["launches"]
                row["pass_ms"] = round(pm, 4)
                row["pass_hbm_frac"] = round((n * DIM * eb + B * DIM * eb) / pm / 1e6 / HBM_PEAK_GBS, 4)
            elif tag.startswith("search_") and v["launches"]:
                row[tag.replace("search_", "") + "_ms"] = round(v["total_ms"] / v["launches"], 4)
        rows.append(row)
    return {"corpus_rows": n, "k": STAGE1_K, "engine": "bf16 filter + fp32 rescore" if eb == 2 else "fp32",
            "rows": rows}


if __name__ == "__main__":
    main()
