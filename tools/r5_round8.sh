#!/bin/bash
out=gpurun_out/r5_round8; mkdir -p $out
python -m pytest tests/test_gpu_sharing.py tests/test_gpu_sampler.py tests/test_gpu_workers.py -q -m gpu -k "sharing or sampler or gather_beside or six_worker" 2>&1 | tail -30 > $out/gputests.txt; tail -8 $out/gputests.txt
# item 6a: three workgroups per compute unit for the throughput-scheme batch kernels (168 registers per lane)
for rep in 1 2; do
  for v in new wpe3; do
    echo "== $v (rep $rep)" | tee -a $out/wpe3_ab.txt
    PSOAP_GP_LIB=$PWD/ab_libs/$v.so python bench.py --mode dag --no-cpu-baseline --no-strong --no-extras --steps 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench --mode dag', round(d['value'],1), 'evals/s', round(d['ms_per_step'],2), 'ms')" | tee -a $out/wpe3_ab.txt
    PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/latency_quick.py "3,5,2,1" "32" nopredict 2>/dev/null | python -c "
import sys, json
print(' '.join('N%d/B%d:%.3f(%.3f)' % (d['N'], d['B'], d['ms'], d['frac']) for d in map(json.loads, sys.stdin)))" | tee -a $out/wpe3_ab.txt
  done
done
# item 5: predict at the retrieve shape under the three schemes and one / two workgroups per compute unit
for sch in 0 1 2; do for w in 256 512; do
  echo "== predict PSOAP_DAG_SCHEME=$sch PSOAP_DAG_WORKERS=$w" | tee -a $out/predict_schemes.txt
  PSOAP_DAG_SCHEME=$sch PSOAP_DAG_WORKERS=$w python tools/latency_quick.py "5" "1" 2>/dev/null | tail -1 | tee -a $out/predict_schemes.txt
done; done
echo "== predict default" | tee -a $out/predict_schemes.txt
python tools/latency_quick.py "5" "1" 2>/dev/null | tail -1 | tee -a $out/predict_schemes.txt
# the 986 us block row of profiles/r4_row_periods.txt (N = 8192, one evaluation, row 51): five more runs
for k in 1 2 3 4 5; do python tools/row_periods.py 5 1 2>/dev/null | head -3 | cut -c1-700 >> $out/row_periods_n8192_x5.txt; done
grep "^span\|^period" $out/row_periods_n8192_x5.txt | cut -c1-400
