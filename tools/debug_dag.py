import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
npx, B, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
ch = syn.make_chunk(2, 1, npx, seed=1)
gps = syn.make_walkers(2, B, seed=2)
lw = np.repeat(ch.lwls[None], B, axis=0)
with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
    h.set_mode(mode)
    P = (ch.N + 127) // 128
    ntask = B * P * (P + 1) // 2
    if os.environ.get('TLOG','1')=='1': h._L.psoap_chunk_dag_tasklog(h._h, None, 0)
    t0 = time.time()
    try:
        out = h.lnlike_batch(lw, gps)
        print(npx, B, mode, out[:3], f"{time.time()-t0:.3f}s", flush=True)
    except Exception as e:
        print("ERR", str(e)[-200:])
    if os.environ.get('TLOG','1')!='1': sys.exit(0)
    log = np.zeros(ntask * 4, dtype=np.uint64)
    h._L.psoap_chunk_dag_tasklog(h._h, log.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), ntask)
    log = log.reshape(ntask, 4).astype(np.int64)
    base = log[log > 0].min()
    for t in range(min(ntask, 24)):
        print(t, [(int(x) - int(base)) / 100.0 if x > 0 else None for x in log[t]])
