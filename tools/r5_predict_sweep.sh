#!/bin/bash
out=gpurun_out/r5_predict; mkdir -p $out
for pct in 4 9 17 25 35 50; do for mn in 2 4 8 16; do
  echo -n "SPLIT_PCT=$pct MIN=$mn: " | tee -a $out/sweep.txt
  PSOAP_DAG_SPLIT_PCT=$pct PSOAP_DAG_SPLIT_MIN=$mn python tools/latency_quick.py 5 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())['predict_cfg5']; print('device %.2f ms (%.3f of peak)' % (d['device_ms'], d['tflops'] / 78.6))" | tee -a $out/sweep.txt
done; done
for len in 4 8 16 32 63; do
  echo -n "SCHUR_LEN=$len: " | tee -a $out/sweep.txt
  PSOAP_SCHUR_LEN=$len python tools/latency_quick.py 5 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())['predict_cfg5']; print('device %.2f ms (%.3f of peak)' % (d['device_ms'], d['tflops'] / 78.6))" | tee -a $out/sweep.txt
done
for jit in 2 4 6 10 16; do
  echo -n "JIT=$jit: " | tee -a $out/sweep.txt
  PSOAP_DAG_JIT=$jit python tools/latency_quick.py 5 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())['predict_cfg5']; print('device %.2f ms (%.3f of peak)' % (d['device_ms'], d['tflops'] / 78.6))" | tee -a $out/sweep.txt
done
