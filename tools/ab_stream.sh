#!/bin/bash
# A/B two builds of the HIP library on the same box, interleaved: ab_libs/old.so against ab_libs/new.so --
# the streamed headline and the launch-per-step path of bench.py (both in every line).
for rep in 1 2 3; do
  for v in old new; do
    PSOAP_GP_LIB=$PWD/ab_libs/$v.so python bench.py --no-cpu-baseline --no-extras --no-strong --steps ${STEPS:-20} --warmup 5 --allow-fallback 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'streamed', round(d['value'],1), 'evals/s', round(d['ms_per_step'],3), 'ms | per launch', round(d['launch_per_step']['evals_per_s'],1), round(d['launch_per_step']['ms_per_step'],3), 'ms')"
  done
done
