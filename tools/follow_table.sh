#!/bin/bash
# Scheme 1 (latency) against scheme 2 (following) of the shipped library, same box, same call: ms per batch of the drop-in
# shapes.   tools/follow_table.sh [Bs] > gpurun_out/follow_table.txt
Bs=${1:-1,2,4,8,16,32}
for s in 1 2 1 2; do
  echo "== PSOAP_DAG_SCHEME=$s"
  PSOAP_DAG_SCHEME=$s timeout 900 python tools/latency_quick.py 1,2,3,5 $Bs nopredict 2>&1 | grep -v amdgpu.ids
done
