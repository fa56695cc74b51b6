#!/bin/bash
out=gpurun_out/r5_round7; mkdir -p $out
for v in nopool pool; do
  echo "== $v" | tee -a $out/part_wait_share.txt
  PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/part_wait_share.py 3 1 2>&1 | tail -2 | tee -a $out/part_wait_share.txt
  PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/part_wait_share.py 3 8 2>&1 | tail -2 | tee -a $out/part_wait_share.txt
  PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/part_wait_share.py 5 1 2>&1 | tail -2 | tee -a $out/part_wait_share.txt
done
