#!/bin/bash
# instruction-cache counters of the resident kernel (are the ~100 KB epilogues an instruction-fetch problem?)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r5_icache; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*" $OUT/avail.txt | sort -u > $OUT/names.txt
cat $OUT/names.txt
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 5 --no-cpu-baseline --no-extras --no-strong"
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/pmc1 -- $BENCH > $OUT/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc2 -- $BENCH > $OUT/pmc2.log 2>&1
tail -3 $OUT/pmc1.log $OUT/pmc2.log
python3 - <<PY
import csv, glob, collections
for d in ("pmc1", "pmc2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        for row in csv.DictReader(open(f)):
            agg[row["Kernel_Name"][:60]][row["Counter_Name"]] += float(row["Counter_Value"])
        for k, v in agg.items():
            if "chol_dag" in k: print(d, k, dict(v))
PY
