"""The experimental following scheme (library built with -DPSOAP_FOLLOW, PSOAP_DAG_SCHEME=2): results against the staged
path and their stability over repeated launches.   PSOAP_GP_LIB=ab_libs/lat_follow.so PSOAP_DAG_SCHEME=2 python tools/follow_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from psoap_amd import synthetic as syn
from psoap_amd.chunk import ChunkHandle
for cfg, B in ((1, 1), (1, 4), (3, 1), (3, 4), (2, 2)):
    ch = syn.make_config_chunk(cfg)
    c = ch.n_components
    gps = syn.make_walkers(c, B, seed=cfg)
    lw = syn.walker_lwls(ch, syn.make_walker_velocities(ch, B, seed=cfg + 10))
    with ChunkHandle(ch.fl, ch.sigma, max_batch=B) as h:
        h.set_mode("staged"); h.upload(lw, gps); h.eval(); want = h.fetch()
        h.set_mode("dag"); h.upload(lw, gps)
        errs = []
        for rep in range(6):
            h.eval(); got = h.fetch()
            errs.append(np.abs(got - want) / np.abs(want))
        print(cfg, B, "rel err per rep:", [" ".join(f"{e:.1e}" for e in er) for er in errs])
