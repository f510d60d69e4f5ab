#!/bin/bash
# The large-footprint cliff: bash tools/gpu_big_footprint.sh [OUTDIR]
#   sweep   width 32768, heights 8192 ... 32768, BC1 and BC6H: one launch / the entry as shipped (read-ahead banding) / back-to-back band
#           launches / the BC1-shaped mock kernel / a band at each quarter of the allocation          (tools/ubench/big_footprint sweep)
#   mock    grid and footprint decoupled, tile orders, launch splits                                  (… mock)
#   mix     read and write footprints decoupled, load cache policies, tiles per workgroup, read pass + decode pass, overlapped   (… mix)
#   pmc     rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) over the whole 32768^2 BC1 image with the read-ahead off
#           (ra0) and on (ra1), and over 16384- and 8192-row images: address-translation misses and the L2's memory-side request latencies
#           (LEVEL / REQ = average cycles a request is outstanding)
set -u
export TMPDIR=/tmp
ROOT=$(pwd); OUT=$ROOT/${1:-gpurun_out/r06/footprint}; mkdir -p $OUT
B=$ROOT/tools/ubench/big_footprint
timeout 300 $B sweep BC1 > $OUT/sweep_bc1.jsonl 2> $OUT/sweep_bc1.err
timeout 300 $B sweep BPTC_FLOAT > $OUT/sweep_bc6h.jsonl 2> $OUT/sweep_bc6h.err
timeout 300 $B mock > $OUT/mock.jsonl 2> $OUT/mock.err
timeout 600 $B mix > $OUT/mix.jsonl 2> $OUT/mix.err
rm -f $OUT/pmc.jsonl
for group in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $group | tr ' ' '+')
  for case in "0 BC1 32768" "1 BC1 32768" "0 BC1 16384" "0 BC1 8192"; do
    ra=${case%% *}; args=${case#* }; ctag=ra${ra}_$(echo $args | tr ' ' '_'); d=$OUT/pmc_${ctag}_$tag
    (cd /tmp && DETEXHIP_READ_AHEAD=$ra timeout 300 rocprofv3 --pmc $group --kernel-trace -d $d -o p --output-format csv -- $B pmc $args > $d.log 2>&1)
    f=$(find $d -name "*counter_collection.csv" 2>/dev/null | head -1); t=$(find $d -name "*kernel_trace.csv" 2>/dev/null | head -1)
    if [ -n "$f" ]; then python3 - "$f" "$t" "read_ahead=$ra $args" "$OUT/pmc.jsonl" <<'PY'
import csv, sys, collections, json
d = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "decode_linear" in r["Kernel_Name"] or "read_ahead" in r["Kernel_Name"]: d[r["Kernel_Name"].split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for r in csv.DictReader(open(sys.argv[2])):
    if "decode_linear" in r["Kernel_Name"] or "read_ahead" in r["Kernel_Name"]: dur[r["Kernel_Name"].split("(")[0][-40:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
med = lambda v: sorted(v)[len(v) // 2]
row = {"case": sys.argv[3], "kernels": {k: dict({c: med(x) for c, x in v.items()}, launches=len(dur[k]), median_us=round(med(dur[k]), 2)) for k, v in d.items()}}
open(sys.argv[4], "a").write(json.dumps(row) + "\n"); print(json.dumps(row))
PY
    else echo "$case $group: no counters" >> $OUT/pmc_failed.txt; tail -3 $d.log >> $OUT/pmc_failed.txt; fi
    rm -rf $d $d.log
  done
done
