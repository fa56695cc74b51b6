#!/bin/bash
out=gpurun_out/r5_final; mkdir -p $out
python tools/kfd_probe.py 2>&1 | tail -12 > $out/kfd_probe.txt
python -c "
import sys; sys.path.insert(0, '.')
from psoap_amd import _lib, synthetic as syn, covariance
ch = syn.make_chunk(1, 4, 100, seed=1)
covariance.lnlike_f(None, ch.lwls[0], ch.fl, ch.sigma, 0.2, 5.0)
print('share_stats of a lone process:', _lib.share_stats(0))" 2>&1 | tail -1 | tee -a $out/kfd_probe.txt
python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee $out/gputests.txt
