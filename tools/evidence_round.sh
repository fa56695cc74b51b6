#!/bin/bash
# The measured evidence of a round in ONE call on ONE box (run through gpurun from the repo root):
#   tools/evidence_round.sh r4 [min_probe_evals_per_s]
# GPU tests, soak, bench line, rocprofv3 passes (tools/profile_round.sh), latency / scheme tables, timelines -- all into
# gpurun_out/; tools/collect_profiles.py TAG then copies the judged ones into profiles/.  Boxes of the pool differ by a few
# per cent (clock 2.30 - 2.35 GHz under this load): with a second argument the call ends early, at the cost of one short
# bench run, when this box's headline rate in that run is below it.
# Exit codes: 0 everything ran and passed; 3 the probe failed; 4 "not this box" (below the asked rate); 5 a step failed
# (which ones: gpurun_out/evidence_status_TAG.txt).
set -u -o pipefail
TAG=${1:-r4}; MINPEAK=${2:-0}
mkdir -p gpurun_out
STATUS=gpurun_out/evidence_status_$TAG.txt
: > $STATUS
FAILED=0
step() {   # step NAME command...: run, record the exit code
    local name=$1; shift
    "$@"
    local rc=$?
    echo "$name rc=$rc" >> $STATUS
    if [ $rc -ne 0 ]; then FAILED=1; echo "STEP FAILED: $name (rc=$rc)"; fi
    return 0
}
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras --no-strong > gpurun_out/probe_$TAG.json 2> gpurun_out/probe_$TAG.err
if [ $? -ne 0 ]; then echo "the probe bench failed:"; tail -5 gpurun_out/probe_$TAG.err; exit 3; fi
PEAK=$(python -c "import json;print(json.load(open('gpurun_out/probe_$TAG.json'))['value'])") || { echo "no value in the probe's line"; exit 3; }
echo "headline rate of this box in a 5-step run: $PEAK evals/s"
if python -c "import sys; sys.exit(0 if float('$PEAK') < float('$MINPEAK') else 1)"; then echo "below $MINPEAK: not this box"; exit 4; fi
sha256sum psoap_amd/csrc/libpsoap_gp.so > gpurun_out/lib_sha256_$TAG.txt
step gputests bash -c "python -m pytest tests -m gpu -q > gpurun_out/gputests_$TAG.txt 2>&1"; grep -v amdgpu.ids gpurun_out/gputests_$TAG.txt | tail -1
step soak bash -c "timeout 500 python tools/soak.py 200 > gpurun_out/soak_$TAG.txt 2>&1"; tail -1 gpurun_out/soak_$TAG.txt
step bench bash -c "python bench.py --steps 20 --warmup 5 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err"; cut -c1-160 gpurun_out/bench_$TAG.json
step profile bash -c "tools/profile_round.sh $TAG > gpurun_out/profile_round_$TAG.log 2>&1"
step latency bash -c "timeout 900 python tools/latency_quick.py 1,2,3,5 1,2,4,8,16,32 > gpurun_out/latency_$TAG.jsonl 2>/dev/null"
step stream_table bash -c "timeout 900 python tools/stream_table.py > gpurun_out/stream_table_$TAG.jsonl 2>/dev/null"
step timeline bash -c "timeout 300 python tools/dag_timeline.py 3 32 2>&1 | grep -v amdgpu.ids > gpurun_out/timeline_launch_per_step_$TAG.txt"
step fill bash -c "timeout 300 python tools/fill_bench.py 2>/dev/null > gpurun_out/fill_$TAG.jsonl"
step row_periods bash -c "{ timeout 120 python tools/row_periods.py 3 1; timeout 120 python tools/row_periods.py 5 1; } 2>&1 | grep -v amdgpu.ids > gpurun_out/row_periods_$TAG.txt"
step predict_timeline bash -c "timeout 300 python tools/predict_timeline.py 1000 2>&1 | grep -v amdgpu.ids > gpurun_out/predict_timeline_$TAG.txt"
step sampler_stream bash -c "timeout 300 python tools/sampler_stream_bench.py 30 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/sampler_stream_$TAG.txt"
step gather bash -c "python -m torch.distributed.run --standalone --local-addr 127.0.0.1 --nnodes=1 --nproc-per-node=1 tools/gather_beside_stream.py 12 device,host,per-step 2>&1 | grep RESULT > gpurun_out/gather_beside_stream_$TAG.txt"
# (round 6) many small matrices in one group launch: the reference's chunk sizes (N = 1008, 2000) x 8 chunks x 32 walkers
step small bash -c "{ timeout 300 python tools/small_bench.py 2 12 84 8 32 0; timeout 300 python tools/small_bench.py 2 10 200 8 32 0; timeout 300 python tools/small_bench.py 2 10 200 1 32 0; } 2>/dev/null | grep '^{' > gpurun_out/small_$TAG.jsonl"
# (round 6) the launch-per-step path with a different proposal in every workspace at every launch, NaN-poisoned workspaces
step soak_perm bash -c "PSOAP_DEBUG_POISON=15 timeout 200 python tools/soak_batch_perm.py 2 8 60 2>&1 | grep -v amdgpu.ids > gpurun_out/soak_perm_$TAG.txt"
# (round 6) the chaos build (ab_libs/chaos.so = tools/build_variant.py chaos -DPSOAP_CHAOS of the same sources): one task in
# sixteen ~100 us late -- the three schemes through streams, bit-identical to the product library's values or the step fails
if [ -f ab_libs/chaos.so ]; then
  step chaos bash -c "rm -f gpurun_out/soakref*.npy; export PSOAP_SOAK_REF=\$PWD/gpurun_out/soakref; { for a in '2 8 2' '2 8 0' '2 8 1'; do timeout 200 python tools/soak_stream.py \$a 10; done; for a in '2 8 2' '2 8 0' '2 8 1'; do PSOAP_GP_LIB=\$PWD/ab_libs/chaos.so timeout 300 python tools/soak_stream.py \$a 60; done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/chaos_$TAG.txt; ! grep -q ' [1-9][0-9]* mismatching' gpurun_out/chaos_$TAG.txt"
  step chaos_tests bash -c "PSOAP_GP_LIB=\$PWD/ab_libs/chaos.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullshape.py tests/test_gpu_group.py tests/test_gpu_stream.py tests/test_gpu_retrieve.py tests/test_gpu_pipeline.py tests/test_gpu_calibration.py -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/chaos_$TAG.txt"
fi
# several processes on this one GPU (no wrong value and no time-out with the library's lock on OR off, 8 and 16 workers; the
# step fails on a single wrong value)
step shared_gpu bash -c "{ for a in '8 300 3 2' '16 300 3 2' '16 200 3 0'; do timeout 600 python tools/shared_gpu_probe.py \$a || exit 1; done; } 2>&1 | grep -v amdgpu.ids > gpurun_out/shared_gpu_probe_$TAG.txt"
cat $STATUS
if [ $FAILED -ne 0 ]; then echo "evidence INCOMPLETE: at least one step failed"; exit 5; fi
echo "evidence complete"
