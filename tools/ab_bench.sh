#!/bin/bash
# A/B two builds of the HIP library on the same box: ab_libs/old.so vs ab_libs/new.so, interleaved.
for rep in 1 2 3; do
  for v in old new; do
    PSOAP_GP_LIB=$PWD/ab_libs/$v.so python bench.py --no-cpu-baseline --steps ${STEPS:-10} $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), 'evals/s', round(d['ms_per_step'],2), 'ms')"
  done
done
