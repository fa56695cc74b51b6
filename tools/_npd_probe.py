import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
ch = syn.make_chunk(2, 3, 100, seed=5)
B = 4
gps = np.tile(np.array(syn.GP_BASE[2]), (B, 1))
lw = np.repeat(ch.lwls[None], B, axis=0).copy()
lw[2, :, 1] = lw[2, :, 0]
sigma = ch.sigma.copy(); sigma[:2] = 0.0
for mode in ("dag", "staged"):
    with ChunkHandle(ch.fl, sigma, max_batch=B) as h:
        h.set_mode(mode)
        print(mode, h.lnlike_batch(lw, gps))

from oracle import oracle as orc_mod
orc = orc_mod
try:
    print("oracle", orc.lnlike(lw[2], ch.fl, sigma, gps[2]))
except Exception as e:
    print("oracle raised", e)
