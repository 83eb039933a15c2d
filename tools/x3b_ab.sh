#!/bin/bash
# same-box A/B of the row-owner kernel's AMDREC_X3B_OPT variants (tools/bin/x3b_opt_<bits>, built by hand from
# tools/x3_probe.hip): every variant ROUNDS times, interleaved, after one warm-up pass
ROUNDS=${ROUNDS:-4}
VARS=${VARS:-"0 1 2 4 8 16 3 7 15 24 31"}
for v in $VARS; do tools/bin/x3b_opt_$v > /dev/null; done
for r in $(seq $ROUNDS); do
  for v in $VARS; do echo -n "opt=$v "; tools/bin/x3b_opt_$v | tail -1; done
done
