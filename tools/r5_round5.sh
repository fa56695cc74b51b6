#!/bin/bash
out=gpurun_out/r5_round5; mkdir -p $out
python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node=1 tools/gather_beside_stream.py 12 device,host,per-step 2>&1 | grep RESULT | tee $out/gather_beside_stream.txt
for rep in 1 2; do
  for v in nopool pool; do
    echo "== $v (rep $rep)" | tee -a $out/latency_ab.txt
    PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/latency_quick.py "3,5,1" "1,8" 2>/dev/null | tee -a $out/latency_ab.txt | python -c "
import sys, json
rows = list(map(json.loads, sys.stdin))
print(' '.join('N%d/B%d:%.3f' % (d['N'], d['B'], d['ms']) for d in rows if 'N' in d), [round(r['predict_cfg5']['device_ms'], 2) for r in rows if 'predict_cfg5' in r])"
  done
done
for jit in 0 3 10; do
  echo "== pool PSOAP_DAG_JIT=$jit" | tee -a $out/jit_sweep.txt
  PSOAP_DAG_JIT=$jit PSOAP_GP_LIB=$PWD/ab_libs/pool.so python tools/latency_quick.py "3,5" "1,8" nopredict 2>/dev/null | tee -a $out/jit_sweep.txt | python -c "
import sys, json
rows = list(map(json.loads, sys.stdin))
print(' '.join('N%d/B%d:%.3f' % (d['N'], d['B'], d['ms']) for d in rows if 'N' in d))"
done
PSOAP_GP_LIB=$PWD/ab_libs/pool.so python tools/wg_occupancy.py 3 1 100 > $out/wg_occupancy_pool.txt 2>&1
PSOAP_GP_LIB=$PWD/ab_libs/pool.so python tools/row_periods.py 3 1 > $out/row_periods_pool.txt 2>&1
cat $out/row_periods_pool.txt | head -3 | cut -c1-400
PSOAP_GP_LIB=$PWD/ab_libs/pool.so python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullshape.py tests/test_gpu_group.py -x -q -m gpu 2>&1 | tail -3 | tee $out/gputests_pool.txt
