#!/bin/bash
# gpurun with retries on exit code 3 ONLY (no box / slot free: nothing ran, nothing charged).  usage: tools/grun.sh TIMEOUT 'command'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
