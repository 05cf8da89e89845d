#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (nothing is charged for those attempts):  tools/grun.sh <timeout-s> '<command>'
T=$1; shift
for i in $(seq 1 30); do
  out=$(/usr/local/graft/bin/gpurun --timeout $T -- "$@" 2>&1)
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"; exit 0
done
echo "$out"; exit 3
