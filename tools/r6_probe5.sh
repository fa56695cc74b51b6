#!/bin/bash
# Round 6, fifth GPU call: the default bench line in its new shape (--mode dag with the resident launch beside it), the GPU parity
# tests through the CHAOS build (every task kind with one task in sixteen ~100 us late), and persistent launches KEPT at 16 worker
# processes (PSOAP_SHARE_DAG_MAX=64: the regime round 5 fenced off after 3 wrong values in 158,400).
set -u
mkdir -p gpurun_out
O=gpurun_out/r6_probe5.txt
: > $O
run() { echo "== $*" >> $O; ( "$@" ) 2>&1 | grep -v amdgpu.ids >> $O; echo "   rc=${PIPESTATUS[0]}" >> $O; }
sha256sum psoap_amd/csrc/libpsoap_gp.so ab_libs/chaos.so >> $O
run timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strong
PSOAP_GP_LIB=$PWD/ab_libs/chaos.so run timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullshape.py tests/test_gpu_group.py tests/test_gpu_stream.py tests/test_gpu_retrieve.py tests/test_gpu_pipeline.py tests/test_gpu_calibration.py -q -m gpu
echo "## 16 worker processes, persistent launches kept (PSOAP_SHARE_DAG_MAX=64), the library's lock on" >> $O
PSOAP_SHARE_DAG_MAX=64 PSOAP_QUIET=1 run timeout 560 python tools/shared_gpu_probe.py 16 2500 3 2
PSOAP_SHARE_DAG_MAX=64 PSOAP_QUIET=1 run timeout 420 python tools/shared_gpu_probe.py 16 2500 1 2
tail -30 $O | cut -c1-700
