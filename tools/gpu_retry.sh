#!/bin/bash
# tools/gpu_retry.sh <timeout s> '<command>': gpurun, tried again while the pod's GPU slots are busy (rc 3)
t=$1; shift
for i in $(seq 1 40); do
  out=$(gpurun --timeout $t -- "$@" 2>&1); rc=$?
  if echo "$out" | grep -q "status=transient"; then sleep 45; continue; fi
  echo "$out"; exit $rc
done
echo "gpu_retry: gave up"; exit 3
