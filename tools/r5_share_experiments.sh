#!/bin/bash
# Round 5, item 1: what happens to persistent launches when many processes share one GPU -- and does the library's account of
# it (moved workgroups) cover every wrong value?   Output: gpurun_out/r5_share/*.txt
out=gpurun_out/r5_share; mkdir -p $out
W=${W:-16}; R=${R:-1000}
run() { name=$1; shift; echo "== $name: $*" | tee -a $out/$name.txt; ( time timeout 1500 env "$@" ) >> $out/$name.txt 2>&1; tail -4 $out/$name.txt; }
# 1. detection only: values of disturbed launches are kept -- wrong values against reported disturbances
run detect_only_new PSOAP_GP_LIB=$PWD/ab_libs/new.so PSOAP_SHARE_DETECT_ONLY=1 PSOAP_SHARE_DAG_MAX=64 python tools/shared_gpu_probe.py $W $R 3 2
# 2. the same with operand loads past the vector L1 (sc1): is a moved workgroup's stale L1 what goes wrong?
run detect_only_sc1 PSOAP_GP_LIB=$PWD/ab_libs/sc1.so PSOAP_SHARE_DETECT_ONLY=1 PSOAP_SHARE_DAG_MAX=64 python tools/shared_gpu_probe.py $W $R 3 2
# 3. the product behaviour with the persistent kernel kept beyond 8 processes: retries, then the staged path
run retry_new PSOAP_GP_LIB=$PWD/ab_libs/new.so PSOAP_SHARE_DAG_MAX=64 python tools/shared_gpu_probe.py $W $R 3 2
# 4. the default policy (staged path beyond 8 processes)
run default_new PSOAP_GP_LIB=$PWD/ab_libs/new.so python tools/shared_gpu_probe.py $W $R 3 2
# 5. no lock: the staged path from the start
run nolock_new PSOAP_GP_LIB=$PWD/ab_libs/new.so python tools/shared_gpu_probe.py $W $(($R / 4)) 3 0
