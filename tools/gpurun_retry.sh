#!/bin/bash
# gpurun with retries while no slot / box is free (exit code 3: nothing charged).  usage: tools/gpurun_retry.sh LOG TIMEOUT 'command'
LOG=$1; T=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout $T -- "$@" > $LOG 2>&1; rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 90
done
