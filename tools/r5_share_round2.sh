#!/bin/bash
# Round 5, second GPU pass of item 1/2: the policy as shipped (persistent launches under the lock up to 8 processes, the
# staged path without the lock beyond), what the verdict asks for (16 / 32 workers x 1500 calls, N = 2000 and 6000, lock on and
# off), the persistent kernel kept at 16 workers with details of any wrong value, the RCCL gather beside a resident launch.
out=gpurun_out/r5_share2; mkdir -p $out
run() { name=$1; shift; echo "== $name: $*" | tee -a $out/$name.txt; ( time timeout ${TMO:-1500} env "$@" ) >> $out/$name.txt 2>&1; grep -v "^$" $out/$name.txt | tail -6 | cut -c1-900; }
R=${R:-1500}
(LIBS="old new nowt" STEPS=10 bash tools/ab_bench3.sh) > $out/ab_bench.txt 2>&1; cat $out/ab_bench.txt
python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node=1 tools/gather_beside_stream.py 12 0,4,8 > $out/gather_beside_stream.txt 2>&1; grep RESULT $out/gather_beside_stream.txt || tail -20 $out/gather_beside_stream.txt
python -m pytest tests/test_gpu_sharing.py tests/test_gpu_workers.py -q -m gpu -k "sharing or eight_ranks or six_worker or rccl" 2>&1 | tail -40 > $out/gputests.txt; tail -15 $out/gputests.txt
for W in 16 32; do for cfg in 3 1; do for lock in 2 0; do
  run probe_w${W}_cfg${cfg}_lock${lock} python tools/shared_gpu_probe.py $W $R $cfg $lock
done; done; done
run probe_w8_cfg3_lock2 python tools/shared_gpu_probe.py 8 2000 3 2
run probe_w16_cfg3_dag PSOAP_SHARE_DAG_MAX=64 python tools/shared_gpu_probe.py 16 1500 3 2
