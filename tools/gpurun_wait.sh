#!/bin/bash
# gpurun with a wait for a free box: exit code 3 = "no box or slot free right now (nothing charged)" -> sleep and ask again.
# Any other exit code (the command ran, or the call was refused) ends the loop.  Usage: tools/gpurun_wait.sh <timeout> '<command>'
T=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 150
done
exit 3
