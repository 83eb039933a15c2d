#!/bin/bash
# Elimination runs of the ONE-wave-per-SIMD 64-row row-owner shape (x3b4) at the full pass size: what a 64-row workgroup
# pays with nothing to hide behind (round 4: sizing the hybrid kernel).  Binaries: tools/bin/x3b4_dbg<bits> (tools/x3_probe.hip,
# -DAMDREC_X3_VARIANT=16 -DAMDREC_X3_PROBE_NS=x3b4 -DAMDREC_X3_DBG=<bits>: 1 no DMA, 4 no fragment reads, 8 no hidden conversion).
ROWS=${ROWS:-256000}
for r in 1 2; do
  echo -n "x3b(8 waves) "; tools/bin/x3b_dbg0 $ROWS | tail -1
  for d in 0 1 4 5 8 13; do echo -n "x3b4 dbg=$d "; tools/bin/x3b4_dbg$d $ROWS | tail -1; done
done
