#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs of bench.py into a small JSON (committed under profiles/).

    python tools/pmc_summary.py <dir with *_counter_collection.csv ...> -o profiles/rNN_pmc.json

Units / corrections follow /opt/skills/guides (MI355X_MICROARCH.md §HBM, cdna_hip_programming.md §7):
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly half of the bytes of a wide
coalesced (16 B/lane) streaming read, so read bytes = 2 x FETCH_SIZE x 1024 for these kernels (all their
global loads are 16 B/lane); WRITE_SIZE is exact for 16 B/lane stores.  Counters were collected in
separate passes (FETCH_SIZE and WRITE_SIZE cannot share one).
"""
import argparse
import collections
import csv
import glob
import json
import os
import re


def short(name):
    m = re.search(r"gemm_nt_kernel<amdrec::Shape<(\d+), (\d+), (\d+), (\d+)[, \w]*>", name)
    e = re.search(r"amdrec::(Epi\w+)", name)
    if m and e:
        wp, wq, tp, tq = (int(m.group(i)) for i in range(1, 5))
        epi = {"EpiLinearT": "linear", "EpiRowBiasT": "linear", "EpiResidualLNT": "residual_ln", "EpiCrossT": "cross", "EpiL2NormT": "l2norm",
               "EpiFilter": "search_filter", "EpiStoreScores": "search_sample"}.get(e.group(1), e.group(1))
        gather = "_gather" if "EmbConcatRows" in name else ""
        return f"{epi}{gather}_{32 * wp * tp}x{32 * wq * tq}"
    mx = re.search(r"gemm_x6_kernel<.*?amdrec::(Epi\w+)", name)
    if mx:
        epi = {"EpiLinearT": "linear", "EpiResidualLNT": "residual_ln", "EpiCrossT": "cross", "EpiL2NormT": "l2norm"}.get(mx.group(1), mx.group(1))
        return f"{epi}_256x128_x6"
    if "scan_filter_kernel" in name:
        return "search_filter_stream128x512_bf16"
    if "x3b4::ranker_x3b_kernel" in name:
        return "ranker_rowowner16_64_x3"
    if "ranker_x3b_kernel" in name:
        return "ranker_rowowner16_128_x3"
    if "sample_max_kernel" in name:
        return "search_sample_max128x512_bf16"
    if "tau_from_maxima_kernel" in name:
        return "search_threshold"
    if "fixup_kernel" in name:
        return "search_fixup"
    if "ranker_x3_kernel" in name:
        return "ranker_rowowner_128_x3"
    if "finalize_mixed_kernel" in name:
        return "search_finalize_mixed"
    if "sample_threshold_kernel" in name:
        return "search_threshold"
    return name.split("(")[0].replace("amdrec::", "").replace("void ", "")[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("-o", "--out", required=True)
    ap.add_argument("--bench-json", help="the bench.py line of the profiled configuration: its `kernels` keys are stored as "
                                         "`bench_tags`, and bench.py quotes this file only for runs with the same tag set")
    a = ap.parse_args()
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for d in a.dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[k][r["Counter_Name"]] += 1
    out = {}
    for k, c in agg.items():
        e = {}
        if "FETCH_SIZE" in c:
            n = launches[k]["FETCH_SIZE"]
            e["launches"] = n
            e["read_bytes_per_launch"] = 2.0 * c["FETCH_SIZE"] * 1024 / n
        if "WRITE_SIZE" in c:
            n = launches[k]["WRITE_SIZE"]
            e["write_bytes_per_launch"] = c["WRITE_SIZE"] * 1024 / n
        if "read_bytes_per_launch" in e and "write_bytes_per_launch" in e:
            e["hbm_bytes_per_launch"] = e["read_bytes_per_launch"] + e["write_bytes_per_launch"]
        if "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"] > 0:
            g = c["GRBM_GUI_ACTIVE"] / 8.0                      # sum over the 8 XCDs
            e["mfma_busy_frac"] = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (g * 1024)   # 256 CUs x 4 SIMDs
            if c.get("SQ_WAVE_CYCLES"):
                e["wave_wait_any_frac"] = c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]
        if "TCC_HIT_sum" in c and c["TCC_HIT_sum"] + c.get("TCC_MISS_sum", 0.0) > 0:
            e["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
        for k2 in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS"):
            if k2 in c:
                e[k2.lower() + "_per_launch"] = c[k2] / launches[k][k2]
        if e:
            out[k] = e
    tags = sorted(json.load(open(a.bench_json))["kernels"]) if a.bench_json else []
    with open(a.out, "w") as f:
        json.dump({"note": "per-launch averages over all launches of the run; see tools/pmc_summary.py for units "
                           "and the gfx950 FETCH_SIZE x2 correction", "bench_tags": tags, "kernels": out}, f, indent=1,
                  sort_keys=True)
    for k, e in sorted(out.items()):
        print(k, {kk: (round(v, 3) if isinstance(v, float) and v < 10 else int(v)) for kk, v in e.items()})


if __name__ == "__main__":
    main()
