#!/bin/bash
# One gpurun call -> the round's rocprofv3 evidence for bench.py's default run:
#   $OUT/stats  : --kernel-trace --stats of `python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-search-sweep ...`
#   $OUT/pmc*   : five --pmc passes (one counter group each, kernel trace only) of a 3-step run
# then tools/pmc_summary.py -> $OUT/pmc.json.  Copy what is to be judged into profiles/ (tools/profile_round.sh prints the list).
R=${1:-r04}
OUT=gpurun_out/prof_$R
cd /root/repo
export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
ARGS="--no-cpu-baseline --no-search-sweep --no-strict-fp32 --no-latency-sweep"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || echo "stats pass failed"
# the same command with the ranker on the STRICT fp32-MFMA engine: per-kernel stats of linear_256x128 /
# residual_ln_256x128 / cross_256x128 -> copy to profiles/rNN_bench_kernel_stats_fp32.csv
rocprofv3 --kernel-trace --stats -d $OUT/stats_fp32 -o stats_fp32 --output-format csv -- python3 bench.py --steps 10 --warmup 3 --ranker-engine fp32 $ARGS > $OUT/bench_under_rocprof_fp32.json 2> $OUT/stats_fp32.err || echo "fp32 stats pass failed"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc$i -o pmc$i --output-format csv -- python3 bench.py --steps 3 --warmup 1 $ARGS > $OUT/pmc$i.json 2> $OUT/pmc$i.err || echo "pmc pass $i failed ($grp)"
done
python3 bench.py $ARGS --steps 10 --warmup 3 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
python3 tools/pmc_summary.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 -o $OUT/pmc.json --bench-json $OUT/bench_plain.json > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head -3
grep -i "rowowner\|x3b" $OUT/pmc_summary.txt
# the dominant kernel's launches in the stats run (kernel trace): average over the 10 timed launches
python3 - $OUT <<'PY'
import csv, glob, sys, json
out = sys.argv[1]
f = glob.glob(out + "/stats/**/*kernel_trace.csv", recursive=True)
if f:
    rows = [r for r in csv.DictReader(open(f[0])) if "x3b::ranker_x3b_kernel" in r["Kernel_Name"]]
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows]
    line = json.load(open(out + "/bench_under_rocprof.json"))
    nw, steps = line.get("warmup_steps_run", 0), line["steps"]          # warm-up + pre-pass launches come first, then the timed region
    timed = d[nw:nw + steps] or d[-steps:]
    doc = {"kernel": "amdrec::x3b::ranker_x3b_kernel", "launch_us": [round(x, 1) for x in d], "launches": len(d),
           "avg_all_us": round(sum(d) / len(d), 1), "avg_timed_region_us": round(sum(timed) / len(timed), 1),
           "bench_live_avg_launch_ms": line["roofline"]["avg_launch_ms"]}
    json.dump(doc, open(out + "/ranker_launches_under_rocprof.json", "w"), indent=1)
    print(doc["avg_all_us"], doc["avg_timed_region_us"], doc["bench_live_avg_launch_ms"])
PY
