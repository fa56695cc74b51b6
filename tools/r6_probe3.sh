#!/bin/bash
# Round 6, third GPU call: the chaos builds (one task in sixteen sleeps ~100 us at its start or ahead of its publications)
# under the stream soak -- with the in-order publication of potrf_done / rows_done and without it -- then the product
# library's long soak of the shape that showed the last small mismatch.
set -u
mkdir -p gpurun_out
O=gpurun_out/r6_probe3.txt
: > $O
run() { echo "== $*" >> $O; ( "$@" ) 2>&1 | grep -v amdgpu.ids >> $O; echo "   rc=${PIPESTATUS[0]}" >> $O; }
sha256sum psoap_amd/csrc/libpsoap_gp.so ab_libs/chaos.so ab_libs/chaos_noinorder.so >> $O
export PSOAP_STREAM_ALLOW_SCHEME2=1
export PSOAP_SOAK_REF=$PWD/gpurun_out/soakref
rm -f gpurun_out/soakref*.npy
echo "## the product library: reference values, short" >> $O
for a in "2 8 2" "2 8 0" "2 8 1" "5 8 2" "1 8 2"; do run timeout 200 python tools/soak_stream.py $a 20; done
echo "## chaos WITHOUT the in-order publication (rounds 3-5's protocol + the accumulator records)" >> $O
PSOAP_GP_LIB=$PWD/ab_libs/chaos_noinorder.so run timeout 400 python tools/soak_stream.py 2 8 2 150
PSOAP_GP_LIB=$PWD/ab_libs/chaos_noinorder.so run timeout 400 python tools/soak_stream.py 1 8 2 60
echo "## chaos WITH it (the product's protocol)" >> $O
PSOAP_GP_LIB=$PWD/ab_libs/chaos.so run timeout 500 python tools/soak_stream.py 2 8 2 300
PSOAP_GP_LIB=$PWD/ab_libs/chaos.so run timeout 400 python tools/soak_stream.py 5 8 2 150
PSOAP_GP_LIB=$PWD/ab_libs/chaos.so run timeout 400 python tools/soak_stream.py 1 8 2 60
PSOAP_GP_LIB=$PWD/ab_libs/chaos.so run timeout 400 python tools/soak_stream.py 2 8 0 100
PSOAP_GP_LIB=$PWD/ab_libs/chaos.so run timeout 400 python tools/soak_stream.py 2 8 1 100
PSOAP_GP_LIB=$PWD/ab_libs/chaos.so PSOAP_DEBUG_POISON=15 run timeout 400 python tools/soak_batch_perm.py 2 8 100
PSOAP_GP_LIB=$PWD/ab_libs/chaos.so PSOAP_DEBUG_POISON=15 run timeout 400 python tools/soak_batch_perm.py 3 1 60
PSOAP_GP_LIB=$PWD/ab_libs/chaos.so PSOAP_DEBUG_POISON=15 run timeout 400 python tools/soak_batch_perm.py 1 32 60
echo "## the product library, long" >> $O
run timeout 1000 python tools/soak_stream.py 2 8 2 840
tail -70 $O
