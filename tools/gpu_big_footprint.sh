#!/bin/bash
# The large-footprint cliff: bash tools/gpu_big_footprint.sh [OUTDIR]
# sweep (heights at width 32768, whole / banded / mock / place) for BC1 and BC6H, the mock kernel's grid-vs-footprint matrix, and
# rocprofv3 --pmc passes (address-translation and traffic counters, one pass per counter group, --kernel-trace only) of the whole
# 32768^2 BC1 image against the same image in 8192-row band launches.
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/${1:-gpurun_out/r06/footprint}; mkdir -p $OUT
B=$ROOT/tools/ubench/big_footprint
timeout 300 $B sweep BC1 > $OUT/sweep_bc1.jsonl 2> $OUT/sweep_bc1.err
timeout 300 $B sweep BPTC_FLOAT > $OUT/sweep_bc6h.jsonl 2> $OUT/sweep_bc6h.err
timeout 300 $B mock > $OUT/mock.jsonl 2> $OUT/mock.err
(cd /tmp && timeout 120 rocprofv3 -L > $OUT/counters_avail.txt 2>&1)
grep -i -o "UTCL[A-Za-z0-9_]*\|TCP_[A-Z0-9_]*TRANSL[A-Z0-9_]*\|[A-Z0-9_]*TLB[A-Z0-9_]*" $OUT/counters_avail.txt | sort -u > $OUT/counters_translation.txt
for group in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_REQUEST_sum TCP_UTCL1_PERMISSION_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WR_UNCACHED_32B_sum" "TCC_TAG_STALL_sum TCC_BUSY_sum" "GRBM_GUI_ACTIVE SQ_WAVES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_IO_CREDIT_STALL_sum"; do
  tag=$(echo $group | tr ' ' '+')
  for case in "BC1 32768" "BC1 32768 8192" "BC1 8192"; do
    ctag=$(echo $case | tr ' ' '_'); d=$OUT/pmc_${ctag}_$tag
    (cd /tmp && timeout 300 rocprofv3 --pmc $group --kernel-trace -d $d -o p --output-format csv -- $B pmc $case > $d.log 2>&1)
    f=$(find $d -name "*counter_collection.csv" 2>/dev/null | head -1)
    if [ -n "$f" ]; then python3 - "$f" "$case" "$OUT/pmc.jsonl" <<'PY'
import csv, sys, collections, json
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "decode_linear" in r["Kernel_Name"]]
d = collections.defaultdict(list)
for r in rows: d[r["Counter_Name"]].append(float(r["Counter_Value"]))
med = {k: sorted(v)[len(v) // 2] for k, v in d.items()}
n = len(rows) // max(1, len(d))
open(sys.argv[3], "a").write(json.dumps({"case": sys.argv[2], "median_per_launch": med, "launches": n}) + "\n")
print(sys.argv[2], med, n)
PY
    else echo "$case $group: no counters" >> $OUT/pmc_failed.txt; tail -3 $d.log >> $OUT/pmc_failed.txt; fi
    rm -rf $d $d.log
  done
done
