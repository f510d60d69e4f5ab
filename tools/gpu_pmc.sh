#!/bin/bash
# generic PMC pass: bash tools/gpu_pmc.sh FORMAT "COUNTER1 COUNTER2 ..." [extra bench args]
set -u
export TMPDIR=/tmp
FMT=$1; CTRS=$2; shift 2
OUT=gpurun_out; mkdir -p $OUT; ROOT=$(pwd); TAG=$(echo $CTRS | tr ' ' '_' | cut -c1-60)
cd /tmp && timeout 600 rocprofv3 --pmc $CTRS --kernel-trace -T -d $ROOT/$OUT/pmc_${FMT}_$TAG -o p --output-format csv -- python $ROOT/bench.py --format $FMT --steps 6 --warmup 2 --no-cpu "$@" > $ROOT/$OUT/pmc_${FMT}_$TAG.log 2>&1
cd $ROOT; f=$(find $OUT/pmc_${FMT}_$TAG -name "*counter_collection.csv" | head -1)
if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "decode_linear" in r["Kernel_Name"]]
d = collections.defaultdict(list)
for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(d.items()): print("  %-28s median %14.0f  (n=%d)" % (k, sorted(v)[len(v)//2], len(v)))
PY
else tail -5 $OUT/pmc_${FMT}_$TAG.log; fi
