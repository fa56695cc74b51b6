#!/bin/bash
# A/B/C of builds of the HIP library on one box, interleaved:   LIBS="old new nowt" STEPS=10 tools/ab_bench3.sh
for rep in 1 2 3; do
  for v in ${LIBS:-old new}; do
    PSOAP_GP_LIB=$PWD/ab_libs/$v.so python bench.py --no-cpu-baseline --no-strong --no-extras --steps ${STEPS:-10} $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), 'evals/s', round(d['ms_per_step'],2), 'ms', 'per-step path', round(d.get('launch_per_step',{}).get('evals_per_s',0),1))"
  done
done
