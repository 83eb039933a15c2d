#!/bin/bash
# Same-box A/B of the corpus pass (csrc/search.hip scan_filter_kernel) variants: AMDREC_SCAN_OPT bits compiled into a copy of
# the library (tools/bin/libamdrec_scan<bits>.so = the product objects + a diagnostic search.hip).
#   tools/scan_ab.sh build   (here, no GPU needed; build the product library first)
#   tools/scan_ab.sh run     (on the GPU box): search-only loop at B = 512 / 128 / 32, variants interleaved, ROUNDS times
set -e
cd "$(dirname "$0")/.."
VARS=${VARS:-"1"}
if [ "$1" = build ]; then
  mkdir -p tools/bin
  for d in $VARS; do
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DAMDREC_SCAN_OPT=$d -x hip \
        -c movie-recommender-demo_amd/csrc/search.hip -o /tmp/search_scan$d.o &&
      /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o tools/bin/libamdrec_scan$d.so /tmp/search_scan$d.o \
        $(ls movie-recommender-demo_amd/lib/*.o | grep -v 'search\.hip' | grep -v '\.san\.') ) &
  done
  wait
else
  for r in $(seq ${ROUNDS:-3}); do
    for B in ${BATCHES:-512 128 32}; do
      python tools/search_only.py $B 20 2>&1 | grep "^B="
      for d in $VARS; do AMDREC_LIB_PATH=tools/bin/libamdrec_scan$d.so python tools/search_only.py $B 20 2>&1 | grep "^B="; done
    done
  done
fi
