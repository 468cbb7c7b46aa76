#!/bin/bash
# Runs HERE (the build container): gpurun with retries while every GPU slot of the pod is busy (exit code 3: nothing charged).
# Usage: bash benchmarks/gpurun_retry.sh <timeout seconds> '<command>'
T=$1; shift
for i in $(seq 1 20); do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
