#!/bin/bash
out=gpurun_out/r5_predict; mkdir -p $out
for rep in 1 2; do
for pct in 19 21 23 25 27 29 31 33 35; do
  echo -n "rep $rep SPLIT_PCT=$pct: " | tee -a $out/sweep2.txt
  PSOAP_DAG_SPLIT_PCT=$pct python tools/latency_quick.py 5 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())['predict_cfg5']; print('device %.2f ms (%.3f of peak)' % (d['device_ms'], d['tflops'] / 78.6))" | tee -a $out/sweep2.txt
done; done
