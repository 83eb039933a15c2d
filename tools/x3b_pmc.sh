#!/bin/bash
# PMC passes over the row-owner probe binary (one counter group per pass; no tracing options besides kernel-trace)
BIN=${1:-tools/bin/x3b_opt_0}
OUT=${2:-gpurun_out/x3b_pmc}
cd /root/repo
export TMPDIR=/tmp
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_INSTS_SMEM" \
           "SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_FLAT SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $grp -d $OUT/p$i -o p$i --output-format csv -- $BIN > $OUT/p$i.log 2>&1 || echo "pass $i failed: $grp"
done
python3 - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(float); n=collections.defaultdict(int)
for f in glob.glob(out+'/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'x3b' not in r['Kernel_Name']: continue
        agg[r['Counter_Name']]+=float(r['Counter_Value']); n[r['Counter_Name']]+=1
for k in sorted(agg): print(f"{k:36s} per launch {agg[k]/max(n[k],1):16.0f}  (rows {n[k]})")
PY
