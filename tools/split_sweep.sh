# sweep of the split-factor knobs of the persistent kernel's task list (experiments): tools/split_sweep.sh
for cfgv in "100 2" "100 4" "70 4" "50 4" "50 3" "35 4"; do set -- $cfgv
echo "== pct $1 minp $2"
PSOAP_DAG_SPLIT_PCT=$1 PSOAP_DAG_SPLIT_MIN=$2 python tools/latency_quick.py ${CFGS:-1,2,3,5} ${BS:-1,2,4,32} nopredict 2>/dev/null | python -c "
import sys, json
print(' '.join('N%d/B%d:%.2f' % (d['N'], d['B'], d['ms']) for d in map(json.loads, sys.stdin)))"
done
