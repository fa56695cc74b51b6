#!/bin/bash
# Round 6, first GPU call: the accumulator fix against the round-5 library under the stream soak that found the defect,
# the coherence litmus tests, and the launch-per-step path with a different proposal in every workspace at every launch.
set -u
mkdir -p gpurun_out
O=gpurun_out/r6_probe1.txt
: > $O
run() { echo "== $*" >> $O; ( "$@" ) 2>&1 | grep -v amdgpu.ids >> $O; echo "   rc=$?" >> $O; }
sha256sum psoap_amd/csrc/libpsoap_gp.so ab_libs/*.so >> $O
run timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')"
run timeout 600 python tools/litmus.py 20000
export PSOAP_STREAM_ALLOW_SCHEME2=1
echo "## control: the round-5 library" >> $O
PSOAP_GP_LIB=$PWD/ab_libs/r5.so run timeout 400 python tools/soak_stream.py 5 8 2 150
PSOAP_GP_LIB=$PWD/ab_libs/r5.so run timeout 400 python tools/soak_stream.py 2 8 2 150
echo "## the accumulator records (round 6)" >> $O
run timeout 500 python tools/soak_stream.py 5 8 2 240
run timeout 500 python tools/soak_stream.py 2 8 2 240
echo "## launch per step, permuted proposals" >> $O
PSOAP_DEBUG_POISON=15 run timeout 400 python tools/soak_batch_perm.py 5 8 120
PSOAP_DEBUG_POISON=15 run timeout 400 python tools/soak_batch_perm.py 5 1 90
PSOAP_DEBUG_POISON=15 run timeout 400 python tools/soak_batch_perm.py 2 8 90
PSOAP_GP_LIB=$PWD/ab_libs/r5.so run timeout 400 python tools/soak_batch_perm.py 2 8 120
echo "## returning-atomic polls (-DPSOAP_RMW_POLL)" >> $O
PSOAP_GP_LIB=$PWD/ab_libs/rmw.so run timeout 400 python tools/soak_stream.py 5 8 2 100
PSOAP_GP_LIB=$PWD/ab_libs/rmw.so run timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-strong
run timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-strong
tail -80 $O
