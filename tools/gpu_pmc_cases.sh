#!/bin/bash
# Counter passes over a list of decode configurations: bash tools/gpu_pmc_cases.sh OUTDIR "COUNTERS" "case1" "case2" ...
# where a case is the argument string of tools/gpu_run_case.py ("BPTC_FLOAT C 16384 8192").  One rocprofv3 --pmc pass per case
# (with --kernel-trace only, as gpurun requires); prints the median per counter over the profiled launches of the decode / fill kernel.
set -u
export TMPDIR=/tmp
OUT=$1; CTRS=$2; shift 2
ROOT=$(pwd); mkdir -p $OUT
for c in "$@"; do
  tag=$(echo "$c" | tr ' ' '_')
  d=$ROOT/$OUT/pmc_$tag
  (cd /tmp && DETEXHIP_LIB=${DETEXHIP_LIB:-} timeout 600 rocprofv3 --pmc $CTRS --kernel-trace -T -d $d -o p --output-format csv -- python $ROOT/tools/gpu_run_case.py $c > $d.log 2>&1)
  f=$(find $d -name "*counter_collection.csv" 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 - "$f" "$c" "$ROOT/$OUT/counters.jsonl" <<'PY'
import csv, sys, collections, json
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "decode_" in r["Kernel_Name"] or "fill_image" in r["Kernel_Name"]]
d = collections.defaultdict(list)
for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
med = {k: sorted(v)[len(v) // 2] for k, v in d.items()}
kern = sorted({r["Kernel_Name"][:60] for r in rows})
print(sys.argv[2], json.dumps(med))
open(sys.argv[3], "a").write(json.dumps({"case": sys.argv[2], "kernels": kern, "median": med, "launches": len(rows) // max(1, len(d))}) + "\n")
PY
  else echo "$c: no counters"; tail -3 $d.log; fi
  rm -rf $d
done
