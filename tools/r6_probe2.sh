#!/bin/bash
# Round 6, second GPU call: litmus tests (complete), the solo kernel (tests + small-matrix table), the GPU test suite,
# then the first long soaks of the accumulator fix.
set -u
mkdir -p gpurun_out
O=gpurun_out/r6_probe2.txt
: > $O
run() { echo "== $*" >> $O; ( "$@" ) 2>&1 | grep -v amdgpu.ids >> $O; echo "   rc=${PIPESTATUS[0]}" >> $O; }
sha256sum psoap_amd/csrc/libpsoap_gp.so >> $O
run timeout 900 python tools/litmus.py 20000
run timeout 600 python -m pytest tests/test_gpu_solo.py -q -m gpu -x
for shape in "2 12 84 8 32" "2 12 84 32 32" "2 10 200 8 32" "2 10 200 1 32" "1 10 200 1 32" "2 12 84 1 32" "2 20 300 8 32"; do
  run timeout 600 python tools/small_bench.py $shape
done
run timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_solo.py
export PSOAP_STREAM_ALLOW_SCHEME2=1
echo "## long soaks, accumulator records" >> $O
run timeout 700 python tools/soak_stream.py 2 8 2 600
run timeout 700 python tools/soak_stream.py 2 32 2 600
run timeout 700 python tools/soak_stream.py 5 8 2 600
tail -60 $O
