#!/bin/bash
# A/B of an environment knob on one box, interleaved:  tools/ab_env.sh "PSOAP_DAG_GEO=0" "PSOAP_DAG_GEO=1" "1,3" "1,4,32"
for rep in 1 2; do
  for v in "$1" "$2"; do
    echo "== $v (rep $rep)"
    env $v python tools/latency_quick.py "$3" "$4" nopredict 2>/dev/null | python -c "
import sys, json
print(' '.join('N%d/B%d:%.2f' % (d['N'], d['B'], d['ms']) for d in map(json.loads, sys.stdin)))"
  done
done
