#!/bin/bash
out=gpurun_out/r5_wt; mkdir -p $out
for rep in 1 2 3; do for v in new nowt; do
  echo -n "$v (rep $rep): " | tee -a $out/wt_ab.txt
  PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/latency_quick.py "3,5,1" "1,8" 2>/dev/null | python -c "
import sys, json
rows = list(map(json.loads, sys.stdin))
print(' '.join('N%d/B%d:%.3f' % (d['N'], d['B'], d['ms']) for d in rows if 'N' in d), 'predict', [round(r['predict_cfg5']['device_ms'], 2) for r in rows if 'predict_cfg5' in r])" | tee -a $out/wt_ab.txt
done; done
PSOAP_GP_LIB=$PWD/ab_libs/nowt.so python tools/predict_timeline.py 2000 2>&1 | grep "PARTs of" | tee -a $out/wt_ab.txt
