import os, sys, subprocess, json
sys.path.insert(0, os.getcwd())
import numpy as np
if len(sys.argv) > 1:
    from psoap_amd import synthetic as syn
    from psoap_amd.chunk import ChunkHandle
    mode, c = int(sys.argv[1]), int(sys.argv[2])
    ch = syn.make_chunk(c, 5, 117, seed=600 + 10 * mode + c)
    M = ch.N if (mode == 1 and c == 3) else 150
    pred = np.linspace(ch.lwls[0].min(), ch.lwls[0].max(), M)
    with ChunkHandle(ch.fl, ch.sigma, max_batch=1) as h:
        mu, S = h.predict(mode, ch.lwls, np.stack([pred] * c), np.full(c, 0.25), syn.GP_BASE[c])
    np.save(sys.argv[3], S)
else:
    for mode, c in ((0, 1), (0, 2), (0, 3), (1, 2), (2, 1)):
        outs = []
        for f in ("0", "1"):
            p = f"/tmp/S_{mode}_{c}_{f}.npy"
            subprocess.check_call([sys.executable, __file__, str(mode), str(c), p], env=dict(os.environ, PSOAP_PREDICT_FUSED=f))
            outs.append(np.load(p))
        d = np.abs(outs[0] - outs[1])
        R = d.shape[0]
        T = (R + 127) // 128
        tile = np.array([[d[128*i:128*(i+1), 128*j:128*(j+1)].max() for j in range(T)] for i in range(T)])
        print("mode", mode, "c", c, "R", R, "max diff", d.max(), "sym", np.array_equal(outs[1], outs[1].T))
        print(np.array2string(tile, precision=1, max_line_width=200))
