#!/bin/bash
# The measured evidence of a round in ONE call on ONE box (run through gpurun from the repo root):
#   tools/evidence_round.sh r3 [min_probe_evals_per_s]
# GPU tests, soak, bench line, rocprofv3 passes (tools/profile_round.sh), latency / scheme tables, timelines -- all into
# gpurun_out/; tools/collect_profiles.py TAG then copies the judged ones into profiles/.  Boxes of the pool differ by a few
# per cent (clock 2.30 - 2.35 GHz under this load): with a second argument the call ends early, at the cost of one short
# bench run, when this box's headline rate in that run is below it.
TAG=${1:-r3}; MINPEAK=${2:-0}
mkdir -p gpurun_out
python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras --no-strong > gpurun_out/probe_$TAG.json 2>/dev/null
PEAK=$(python -c "import json;print(json.load(open('gpurun_out/probe_$TAG.json'))['value'])")
echo "headline rate of this box in a 5-step run: $PEAK evals/s"
if python -c "import sys; sys.exit(0 if float('$PEAK') < float('$MINPEAK') else 1)"; then echo "below $MINPEAK: not this box"; exit 0; fi
sha256sum psoap_amd/csrc/libpsoap_gp.so > gpurun_out/lib_sha256_$TAG.txt
python -m pytest tests -m gpu -q > gpurun_out/gputests_$TAG.txt 2>&1; grep -v amdgpu.ids gpurun_out/gputests_$TAG.txt | tail -1
timeout 500 python tools/soak.py 200 > gpurun_out/soak_$TAG.txt 2>&1; tail -1 gpurun_out/soak_$TAG.txt
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err; cut -c1-160 gpurun_out/bench_$TAG.json
tools/profile_round.sh $TAG > gpurun_out/profile_round_$TAG.log 2>&1
timeout 900 python tools/latency_quick.py 1,2,3,5 1,2,4,8,16,32 > gpurun_out/latency_$TAG.jsonl 2>/dev/null
timeout 300 python tools/dag_timeline.py 3 32 2>&1 | grep -v amdgpu.ids > gpurun_out/timeline_$TAG.txt
timeout 300 python tools/fill_bench.py 2>/dev/null > gpurun_out/fill_$TAG.jsonl
tools/follow_table.sh 1,2,3,4,6,8,12,16,24,32 > gpurun_out/follow_table_$TAG.txt 2>&1
{ timeout 120 python tools/row_periods.py 3 1; timeout 120 python tools/row_periods.py 1 1; timeout 120 python tools/row_periods.py 5 1; } 2>&1 | grep -v amdgpu.ids > gpurun_out/row_periods_$TAG.txt
timeout 120 python tools/wg_occupancy.py 3 1 200 2>&1 | grep -v amdgpu.ids > gpurun_out/wg_occupancy_$TAG.txt
echo "evidence complete"
