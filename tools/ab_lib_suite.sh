#!/bin/bash
# the partial-tile hand-over in pieces (tools/predict_timeline.py), and the store routine without loads in its element loop
# against round 5's first evidence library (ab_libs/r5base.so = the sources of commit 7e0f... built by tools/build_variant.py)
out=gpurun_out/r5_slot; mkdir -p $out; rm -f $out/*.txt
cp psoap_amd/csrc/libpsoap_gp.so ab_libs/new.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullshape.py tests/test_gpu_retrieve.py tests/test_gpu_group.py tests/test_gpu_stream.py -q -m gpu -x 2>&1 | tail -3 | tee -a $out/ab.txt
for v in ${LIBS:-r5base new}; do
  echo "== $v" | tee -a $out/timeline.txt
  PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/predict_timeline.py 1000 2>&1 | head -16 | tee -a $out/timeline.txt
done
for rep in 1 2; do for v in ${LIBS:-r5base new}; do
  echo -n "$v predict: " | tee -a $out/ab.txt
  PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/latency_quick.py 5 1 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())['predict_cfg5']; print('device %.2f ms (%.3f of peak)' % (d['device_ms'], d['tflops'] / 78.6))" | tee -a $out/ab.txt
done; done
LIBS="${LIBS:-r5base new}" STEPS=10 tools/ab_bench3.sh 2>&1 | tee -a $out/ab.txt
for v in ${LIBS:-r5base new}; do
  echo "== $v lnlike latencies (cfg 1,2,3,5; B = 1, 4, 8, 32)" | tee -a $out/ab.txt
  PSOAP_GP_LIB=$PWD/ab_libs/$v.so python tools/latency_quick.py 1,2,3,5 1,4,8,32 2>/dev/null | grep '"B"' | python -c "
import sys, json
for ln in sys.stdin:
    d = json.loads(ln); print(d['N'], d['B'], d['ms'], 'ms', round(d['tflops'] / 78.6, 3))" | tee -a $out/ab.txt
done
for v in ${LIBS:-r5base new}; do
  echo -n "$v staged path, N = 6000, 32 matrices: " | tee -a $out/ab.txt
  PSOAP_GP_LIB=$PWD/ab_libs/$v.so python -c "
import sys, time; sys.path.insert(0, '.')
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
ch = syn.make_config_chunk(3); B = 32
gps = syn.make_walkers(ch.n_components, B, seed=1); lw = np.repeat(ch.lwls[None], B, axis=0)
with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
    h.set_mode('staged'); h.upload(lw, gps)
    for _ in range(2): h.eval(); h.fetch()
    t0 = time.perf_counter()
    for _ in range(5): h.eval(); out = h.fetch()
    print('%.2f ms per launch' % (1e3 * (time.perf_counter() - t0) / 5), repr(out[:2]))
" 2>&1 | tail -1 | tee -a $out/ab.txt
done
