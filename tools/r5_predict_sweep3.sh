#!/bin/bash
# predict's split knobs again, after the partial-tile hand-over became cheaper (late round 5)
out=gpurun_out/r5_predict3; mkdir -p $out; rm -f $out/sweep.txt
one() { python tools/latency_quick.py 5 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())['predict_cfg5']; print('device %.2f ms (%.3f of peak)' % (d['device_ms'], d['tflops'] / 78.6))"; }
for rep in 1 2; do echo -n "defaults: " | tee -a $out/sweep.txt; one | tee -a $out/sweep.txt; done
for pct in 9 13 17 21 25 30 35; do for mn in 2 4 8; do
  echo -n "SPLIT_PCT=$pct MIN=$mn: " | tee -a $out/sweep.txt
  PSOAP_DAG_SPLIT_PCT=$pct PSOAP_DAG_SPLIT_MIN=$mn one | tee -a $out/sweep.txt
done; done
for len in 4 6 8 12 16; do
  echo -n "SCHUR_LEN=$len: " | tee -a $out/sweep.txt
  PSOAP_SCHUR_LEN=$len one | tee -a $out/sweep.txt
done
for jit in 4 6 8 10 14; do
  echo -n "JIT=$jit: " | tee -a $out/sweep.txt
  PSOAP_DAG_JIT=$jit one | tee -a $out/sweep.txt
done
