#!/bin/bash
out=gpurun_out/r5_round9; mkdir -p $out
for rep in 1 2 3; do for v in new expprobe; do
  echo "== $v (rep $rep)" | tee -a $out/exp_probe.txt
  PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/latency_quick.py "3,1" "32" nopredict 2>/dev/null | python -c "
import sys, json
print(' '.join('N%d/B%d:%.3f(%.3f)' % (d['N'], d['B'], d['ms'], d['frac']) for d in map(json.loads, sys.stdin)))" | tee -a $out/exp_probe.txt
done; done
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $out/gputests_all.txt; tail -6 $out/gputests_all.txt
