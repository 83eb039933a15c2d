#!/bin/bash
# same-box A/B of LLVM AMDGPU scheduler strategies on the row-owner kernel (tools/bin/x3s_* built from tools/x3_probe.hip with
# -mllvm -amdgpu-sched-strategy=... etc.; x3s_base = the product's flags)
for r in 1 2 3; do
  for v in base maxilp maxmem iterilp itermin trackers nounclust; do echo -n "$v "; timeout -k 10 60 tools/bin/x3s_$v 256000 | tail -1 || exit 1; done
done
