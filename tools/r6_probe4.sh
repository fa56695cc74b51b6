#!/bin/bash
# Round 6, fourth GPU call: the long soaks of the three schemes through resident launches (product library), and persistent
# launches KEPT beyond 8 processes (PSOAP_SHARE_DAG_MAX=64: the regime round 5 fenced off after 3 wrong values in 158,400).
set -u
mkdir -p gpurun_out
O=gpurun_out/r6_probe4.txt
: > $O
run() { echo "== $*" >> $O; ( "$@" ) 2>&1 | grep -v amdgpu.ids >> $O; echo "   rc=${PIPESTATUS[0]}" >> $O; }
sha256sum psoap_amd/csrc/libpsoap_gp.so >> $O
echo "## streams, N = 4096 (cfg 2) and N = 8192 (cfg 5), 32 lanes" >> $O
run timeout 700 python tools/soak_stream.py 2 32 2 600
run timeout 700 python tools/soak_stream.py 2 32 0 540
run timeout 700 python tools/soak_stream.py 2 32 1 540
run timeout 700 python tools/soak_stream.py 5 32 2 500
run timeout 500 python tools/soak_stream.py 5 32 0 300
run timeout 500 python tools/soak_stream.py 5 32 1 300
echo "## 16 worker processes, persistent launches kept (PSOAP_SHARE_DAG_MAX=64), the library's lock on" >> $O
PSOAP_SHARE_DAG_MAX=64 PSOAP_QUIET=1 run timeout 900 python tools/shared_gpu_probe.py 16 12000 1 2
PSOAP_SHARE_DAG_MAX=64 PSOAP_QUIET=1 run timeout 1100 python tools/shared_gpu_probe.py 16 4000 3 2
tail -40 $O
