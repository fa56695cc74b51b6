#!/bin/bash
# A/B two builds of the HIP library: dag-mode latency table (B = 1, 4, 32) for every config shape.
for v in old new; do
  echo "== $v"
  PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/latency_table.py 2>/dev/null | grep '"dag"' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(f\"N={d['N']:5d} c={d['c']} B={d['B']:2d}: {d['ms_per_batch']:8.2f} ms  {d['evals_per_s']:8.1f} evals/s  {d['tflops']:5.1f} TF\")"
done
