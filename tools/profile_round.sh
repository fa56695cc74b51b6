#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh r1
# kernel trace + stats, then SQ (MFMA) counters, then HBM read / write counters in separate --pmc passes.
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (round 4: --warmup 5 -- the resident launch of the warm-up and that of the timed region then complete 160 matrices each,
# so the dispatches of k_chol_dag<2, false, false, stream> in one run are alike)
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-extras --no-strong"
FULL="python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1
# (round 5) the DRIVER's exact command under the kernel trace: its timed region is ONE dispatch of the resident kernel that
# completes 20 x 32 = 640 evaluations -- roofline.frac of the bench line can be recomputed from this trace alone
DRIVER="python3 $ROOT/bench.py --steps 20 --warmup 5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_driver -- $DRIVER > $OUT/trace_driver.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
# the side legs (stand-alone fill, predict at the retrieve shape, 8 chunks in one launch): kernel trace only
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_full -- $FULL > $OUT/trace_full.log 2>&1
python3 $ROOT/tools/summarize_profiles.py $OUT $TAG > $OUT/summary_$TAG.md
cat $OUT/summary_$TAG.md
tail -1 $OUT/trace.log | cut -c1-400
