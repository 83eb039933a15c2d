#!/bin/bash
# What would a kernel gain whose fragment sets feed FOUR row tiles (the hybrid row-tile-owner / column-split proposal)?
# Upper bound by elimination: the product kernels with fragment reads for ONE group in four (AMDREC_X3_DBG=64), and none (4),
# in the 8-wave 128-row shape (x3b: two waves per SIMD - no real kernel can share fragments across its 8 row tiles, the
# planes of 4 tiles alone are 256 registers) and the 4-wave 64-row shape the proposal needs (x3b4: one wave per SIMD).
ROWS=${ROWS:-256000}
for r in 1 2; do
  for b in x3b_dbg0 x3b_dbg64 x3b_dbg4 x3b4_dbg0 x3b4_dbg64 x3b4_dbg4; do echo -n "$b "; tools/bin/$b $ROWS | tail -1; done
done
