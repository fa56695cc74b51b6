#!/bin/bash
# Round 6, fifth GPU call (short): the default bench line in its new shape (--mode dag with the resident launch beside it), and
# the GPU parity tests through the CHAOS build (every task kind -- predict's augmented launch, group launches, the three
# schemes, streams -- with one task in sixteen ~100 us late: goldens and bit-identity must hold).
set -u
mkdir -p gpurun_out
O=gpurun_out/r6_probe5.txt
: > $O
run() { echo "== $*" >> $O; ( "$@" ) 2>&1 | grep -v amdgpu.ids >> $O; echo "   rc=${PIPESTATUS[0]}" >> $O; }
run timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-strong
PSOAP_GP_LIB=$PWD/ab_libs/chaos.so run timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullshape.py tests/test_gpu_group.py tests/test_gpu_stream.py tests/test_gpu_retrieve.py tests/test_gpu_pipeline.py tests/test_gpu_calibration.py -q -m gpu
tail -30 $O | cut -c1-600
