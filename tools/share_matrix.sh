#!/bin/bash
# The review's sharing matrix once more, with the FINAL library of round 5 (its kernels changed after profiles/r5_share_probes.txt):
# 16 / 32 workers x 1500 calls, N = 2000 and 6000, lock on and off; 8 workers x 2000 under the lock.  SHORT=1: without the two
# 32-worker N = 6000 runs (11 minutes).
out=gpurun_out/share_matrix; mkdir -p $out; rm -f $out/*.txt
sha256sum psoap_amd/csrc/libpsoap_gp.so | tee $out/all.txt
run() { name=$1; shift; echo "== $name: $*" | tee -a $out/all.txt; ( time timeout ${TMO:-1500} env "$@" ) > $out/$name.txt 2>&1; grep -v "^$\|^psoap: " $out/$name.txt | tail -6 | cut -c1-900 | tee -a $out/all.txt; }
R=${R:-1500}
for W in 16 32; do for cfg in 3 1; do for lock in 2 0; do
  if [ "${SHORT:-0}" = 1 ] && [ $W = 32 ] && [ $cfg = 3 ]; then continue; fi
  run probe_w${W}_cfg${cfg}_lock${lock} python tools/shared_gpu_probe.py $W $R $cfg $lock
done; done; done
run probe_w8_cfg3_lock2 python tools/shared_gpu_probe.py 8 2000 3 2
