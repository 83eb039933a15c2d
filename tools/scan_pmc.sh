#!/bin/bash
# SQ counter passes over a search-only loop (tools/search_only.py B reps): where the corpus pass's wave time goes
B=${1:-512}
OUT=${2:-gpurun_out/scan_pmc}
cd /root/repo; export TMPDIR=/tmp; mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/p$i -o p$i --output-format csv -- python3 tools/search_only.py $B 10 > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(out+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0][-40:]
        agg[k][r['Counter_Name']]+=float(r['Counter_Value']); n[k][r['Counter_Name']]+=1
for k in agg:
    if 'scan_filter' not in k and 'finalize_mixed' not in k and 'sample_max' not in k: continue
    print(k)
    for c in sorted(agg[k]): print(f"   {c:32s} {agg[k][c]/max(n[k][c],1):16.0f}")
PY
