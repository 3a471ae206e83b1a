#!/bin/bash
# gpurun with a wait for a free GPU slot: exit code 3 = "no box or slot free right now (nothing charged)" is retried after a pause;
# every other outcome (ran, refused, timed out) is returned as is -- a command that ran is never run twice.
#   tools/gpu_retry.sh <timeout seconds> '<command>'
t=$1; shift
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
