#!/bin/bash
# Round 5, item 7: split-factor knobs of the latency scheme for big batches of small matrices (N = 2000 / 4096, B = 32)
out=gpurun_out/r5_small; mkdir -p $out
for pct in 10 20 35 50 70 100; do for mn in 2 4 8; do
  echo -n "PCT=$pct MIN=$mn: " | tee -a $out/split_sweep.txt
  PSOAP_DAG_SPLIT_PCT=$pct PSOAP_DAG_SPLIT_MIN=$mn python tools/latency_quick.py 1,2 32 nopredict 2>/dev/null | python -c "
import sys, json
print(' '.join('N%d/B%d:%.3f(%.3f)' % (d['N'], d['B'], d['ms'], d['frac']) for d in map(json.loads, sys.stdin)))" | tee -a $out/split_sweep.txt
done; done
for jit in 0 2 4 6 10; do
  echo -n "JIT=$jit: " | tee -a $out/split_sweep.txt
  PSOAP_DAG_JIT=$jit python tools/latency_quick.py 1,2 32 nopredict 2>/dev/null | python -c "
import sys, json
print(' '.join('N%d/B%d:%.3f(%.3f)' % (d['N'], d['B'], d['ms'], d['frac']) for d in map(json.loads, sys.stdin)))" | tee -a $out/split_sweep.txt
done
for q in 8 4 2 1; do
  echo -n "QUEUES=$q: " | tee -a $out/split_sweep.txt
  PSOAP_DAG_QUEUES=$q python tools/latency_quick.py 1,2 32 nopredict 2>/dev/null | python -c "
import sys, json
print(' '.join('N%d/B%d:%.3f(%.3f)' % (d['N'], d['B'], d['ms'], d['frac']) for d in map(json.loads, sys.stdin)))" | tee -a $out/split_sweep.txt
done
