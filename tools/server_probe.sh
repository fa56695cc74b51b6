#!/bin/bash
# Round 5: the reference's worker-per-chunk model through ONE process that owns the GPU (psoap_amd/server.py) against the same
# workers each with its own GPU context (the staged path without the lock beyond 8 of them).
out=gpurun_out/server_probe; mkdir -p $out
python -m pytest tests/test_gpu_server.py -q -m gpu 2>&1 | tail -5 | tee $out/gputest.txt
for W in 8 16 32; do for cfg in 1 3; do
  echo "== $W workers cfg $cfg through the server" | tee -a $out/probe.txt
  PSOAP_GPU_SERVER=auto PSOAP_GPU_SERVER_IDLE_S=5 PSOAP_LOCK_DIR=/tmp/psoap-srv-$W-$cfg timeout 900 python tools/shared_gpu_probe.py $W 600 $cfg 2 2>&1 | grep -v "^$" | cut -c1-700 | tee -a $out/probe.txt
  sleep 7
done; done
