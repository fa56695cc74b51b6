#!/bin/bash
# Interleaved bench of ONE library under several values of an environment variable:
#   VAR=PSOAP_DAG_SCHEME VALUES="0 2 4 8" LIB=new bash tools/ab_env.sh
for rep in 1 2; do
  for v in $VALUES; do
    env $VAR=$v PSOAP_GP_LIB=$PWD/ab_libs/${LIB:-new}.so python bench.py --no-cpu-baseline --steps ${STEPS:-10} $BENCH_ARGS 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$VAR=$v', round(d['value'],1), 'evals/s', round(d['ms_per_step'],2), 'ms')"
  done
done
