#!/bin/bash
# kernel timeline of one single-user request (tools/request_trace.py under rocprofv3 --kernel-trace): launches and gaps
OUT=gpurun_out/prof_request
cd /root/repo; export TMPDIR=/tmp; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace -d $OUT/trace -o trace --output-format csv -- python3 tools/request_trace.py 1 60 > $OUT/run.log 2>&1
cat $OUT/run.log | grep "B="
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "select_topk" in r["Kernel_Name"]]
a, b = idx[40], idx[41]                       # one back-to-back request: from the end of request 40's last kernel
t0 = int(rows[a]["End_Timestamp"])
busy = 0
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'][:100]}")
print(f"request span {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
a, b = idx[-3], idx[-2]                       # one synchronised request
t0 = int(rows[a]["End_Timestamp"])
print("synchronised:")
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'][:60]}")
PY
