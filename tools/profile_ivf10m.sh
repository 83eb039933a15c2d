#!/bin/bash
# rocprofv3 evidence for configs[4] on one GPU (10M clustered ads, IVF 4096 x 64, 512 users): kernel stats + the PMC passes of
# tools/profile_round.sh for the list-scan kernels.  Copy $OUT/stats/*kernel_stats.csv and $OUT/pmc.json into profiles/.
OUT=gpurun_out/prof_ivf10m
cd /root/repo
export TMPDIR=/tmp
rm -rf $OUT; mkdir -p $OUT
ARGS="--ads 10000000 --index ivf --nlist 4096 --nprobe 64 --no-cpu-baseline --no-search-sweep --no-strict-fp32 --no-latency-sweep"
rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats --output-format csv -- python3 bench.py --steps 10 --warmup 3 $ARGS > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || echo "stats pass failed"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp -d $OUT/pmc$i -o pmc$i --output-format csv -- python3 bench.py --steps 3 --warmup 1 $ARGS > $OUT/pmc$i.json 2> $OUT/pmc$i.err || echo "pmc pass $i failed ($grp)"
done
python3 bench.py $ARGS --steps 10 --warmup 3 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
python3 tools/pmc_summary.py $OUT/pmc1 $OUT/pmc2 $OUT/pmc3 $OUT/pmc4 $OUT/pmc5 -o $OUT/pmc.json --bench-json $OUT/bench_plain.json > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_stats.csv" | head -3
grep -i "ivf" $OUT/pmc_summary.txt | head
