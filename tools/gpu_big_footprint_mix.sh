#!/bin/bash
# Second pass on the large-footprint cliff: the mock's read and write footprints decoupled (tools/ubench/big_footprint mix) and the
# L2's memory-side request latencies (LEVEL / REQ = average cycles a request is outstanding) of the whole 32768^2 image against an
# 8192-row image.  bash tools/gpu_big_footprint_mix.sh [OUTDIR]
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/${1:-gpurun_out/r06/footprint}; mkdir -p $OUT
B=$ROOT/tools/ubench/big_footprint
timeout 600 $B mix > $OUT/mix.jsonl 2> $OUT/mix.err
for group in "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum"; do
  tag=$(echo $group | tr ' ' '+')
  for case in "BC1 32768" "BC1 16384" "BC1 8192"; do
    ctag=$(echo $case | tr ' ' '_'); d=$OUT/pmc_${ctag}_$tag
    (cd /tmp && timeout 300 rocprofv3 --pmc $group --kernel-trace -d $d -o p --output-format csv -- $B pmc $case > $d.log 2>&1)
    f=$(find $d -name "*counter_collection.csv" 2>/dev/null | head -1)
    if [ -n "$f" ]; then python3 - "$f" "$case" "$OUT/pmc_latency.jsonl" <<'PY'
import csv, sys, collections, json
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "decode_linear" in r["Kernel_Name"]]
d = collections.defaultdict(list)
for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
med = {k: sorted(v)[len(v) // 2] for k, v in d.items()}
open(sys.argv[3], "a").write(json.dumps({"case": sys.argv[2], "median_per_launch": med, "launches": len(rows) // max(1, len(d))}) + "\n")
print(sys.argv[2], med)
PY
    else echo "$case $group: no counters" >> $OUT/pmc_failed.txt; tail -3 $d.log >> $OUT/pmc_failed.txt; fi
    rm -rf $d $d.log
  done
done
cat $OUT/mix.jsonl
