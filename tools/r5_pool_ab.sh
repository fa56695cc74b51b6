#!/bin/bash
# Round 5, item 4: ready-only hand-out of the PART tasks (ab_libs/pool.so) against list order (ab_libs/nopool.so), same box,
# interleaved: single evaluations and small batches, predict at the retrieve shape, the workgroups' occupancy.
out=gpurun_out/r5_pool; mkdir -p $out
for rep in 1 2; do
  for v in nopool pool; do
    echo "== $v (rep $rep)" | tee -a $out/latency_ab.txt
    PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/latency_quick.py "3" "1,2,4,8" nopredict 2>/dev/null | tee -a $out/latency_ab.txt | python -c "
import sys, json
print(' '.join('N%d/B%d:%.3f' % (d['N'], d['B'], d['ms']) for d in map(json.loads, sys.stdin)))"
    PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/latency_quick.py "1,2" "16,32" nopredict 2>/dev/null | tee -a $out/latency_ab.txt | python -c "
import sys, json
print(' '.join('N%d/B%d:%.3f(%.3f)' % (d['N'], d['B'], d['ms'], d['frac']) for d in map(json.loads, sys.stdin)))"
    PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/latency_quick.py "5,2,1" "1,8" 2>/dev/null | tee -a $out/latency_ab.txt | python -c "
import sys, json
rows = list(map(json.loads, sys.stdin))
print(' '.join('N%d/B%d:%.3f' % (d['N'], d['B'], d['ms']) for d in rows if 'N' in d), [r for r in rows if 'predict_cfg5' in r])"
  done
done
for v in nopool pool; do
  PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/wg_occupancy.py 3 1 100 > $out/wg_occupancy_$v.txt 2>&1
  PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/row_periods.py 3 1 > $out/row_periods_$v.txt 2>&1
  PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/row_periods.py 5 1 >> $out/row_periods_$v.txt 2>&1
done
head -3 $out/wg_occupancy_pool.txt; cat $out/row_periods_pool.txt | head -8
# parity with the pool: the golden / oracle suites
PSOAP_GP_LIB=$PWD/ab_libs/pool.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullshape.py tests/test_gpu_group.py tests/test_gpu_stress.py -x -q -m gpu 2>&1 | tail -5 | tee $out/gputests_pool.txt
# JIT lead of the pool's order (0: readiness order)
for jit in 0 3 6 10; do
  echo "== pool PSOAP_DAG_JIT=$jit" | tee -a $out/jit_sweep.txt
  PSOAP_DAG_JIT=$jit PSOAP_GP_LIB=$PWD/ab_libs/pool.so python tools/latency_quick.py "3,5" "1,8" nopredict 2>/dev/null | tee -a $out/jit_sweep.txt | python -c "
import sys, json
print(' '.join('N%d/B%d:%.3f' % (d['N'], d['B'], d['ms']) for d in map(json.loads, sys.stdin)))"
done
