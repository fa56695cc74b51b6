#!/bin/bash
# A/B builds of the HIP library on the same box (ab_libs/<name>.so), interleaved:
#   tools/ab_latency.sh "base new" "3" "1,4,32"
for rep in 1 2; do
  for v in $1; do
    echo "== $v (rep $rep)"
    PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/latency_quick.py "$2" "$3" nopredict 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print('  N=%d B=%d  %.3f ms  frac %.3f' % (d['N'], d['B'], d['ms'], d['frac']))"
  done
done
