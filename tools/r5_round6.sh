#!/bin/bash
out=gpurun_out/r5_round6; mkdir -p $out
for rep in 1 2; do
  for v in nopool pool pool16 pool64; do
    echo "== $v (rep $rep)" | tee -a $out/latency_ab.txt
    PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/latency_quick.py "3,5,1" "1,8" 2>/dev/null | tee -a $out/latency_ab.txt | python -c "
import sys, json
rows = list(map(json.loads, sys.stdin))
print(' '.join('N%d/B%d:%.3f' % (d['N'], d['B'], d['ms']) for d in rows if 'N' in d), [round(r['predict_cfg5']['device_ms'], 2) for r in rows if 'predict_cfg5' in r])"
  done
done
for v in pool16 pool64; do
for jit in 0 3 10; do
  echo "== $v PSOAP_DAG_JIT=$jit" | tee -a $out/jit_sweep.txt
  PSOAP_DAG_JIT=$jit PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/latency_quick.py "3,5" "1,8" nopredict 2>/dev/null | tee -a $out/jit_sweep.txt | python -c "
import sys, json
rows = list(map(json.loads, sys.stdin))
print(' '.join('N%d/B%d:%.3f' % (d['N'], d['B'], d['ms']) for d in rows if 'N' in d))"
done; done
PSOAP_GP_LIB=$PWD/ab_libs/pool64.so python tools/wg_occupancy.py 3 1 100 > $out/wg_occupancy_pool64.txt 2>&1
PSOAP_GP_LIB=$PWD/ab_libs/nopool.so python tools/wg_occupancy.py 3 1 100 > $out/wg_occupancy_nopool.txt 2>&1
PSOAP_GP_LIB=$PWD/ab_libs/pool64.so python tools/row_periods.py 3 1 > $out/row_periods_pool64.txt 2>&1
cat $out/row_periods_pool64.txt | head -3 | cut -c1-400
