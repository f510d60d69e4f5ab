#!/bin/bash
# copy the judged summaries of the last tools/gpu_round.sh run from gpurun_out/ into profiles/<tag>/
set -u
TAG=${1:-r01}; SRC=gpurun_out; DST=profiles/$TAG
mkdir -p $DST; find $DST -type f ! -name "*variant5*" -delete
cp $SRC/prof_trace/bc1_kernel_stats.csv $DST/bc1_8192_kernel_stats.csv
cp $SRC/prof_trace/bc1_domain_stats.csv $DST/bc1_8192_domain_stats.csv 2>/dev/null
for d in $SRC/prof_pmc_*/; do t=$(basename $d); t=${t#prof_pmc_}; (head -1 $d/bc1_counter_collection.csv; grep decode_linear $d/bc1_counter_collection.csv | head -12) > $DST/bc1_8192_pmc_$t.csv; done
for fmt in BPTC BPTC_FLOAT; do f=$SRC/prof_sq_$fmt/sq_counter_collection.csv; [ -f $f ] && (head -1 $f; grep decode_linear $f | head -16) > $DST/$(echo $fmt | tr A-Z a-z)_8192_pmc_SQ.csv; done
cp $SRC/formats_8192.json $SRC/bench.json $SRC/bench_v1.json $SRC/bench_v2.json $SRC/bench_16384.json $SRC/bench_bc1_BGRA8.json $SRC/bench_bc1_RGB8.json \
   $SRC/bench_bc6h_v0.json $SRC/bench_bc6h_v2.json $SRC/bench_bc6h_v3.json $SRC/bench_bc6h_32768x4096.json $SRC/bench_bc1_32768x8192.json \
   $SRC/bench_bc7_v0.json $SRC/bench_bc7_v3.json $SRC/bench_bc7_v4.json $SRC/bench_bc7_v5.json $SRC/bench_tiled_BC1.json $SRC/bench_tiled_BPTC.json $SRC/bench_tiled_BPTC_FLOAT.json $SRC/bench_tiled_RGTC2.json \
   $SRC/sustain_windows.txt $SRC/histogram.txt $SRC/hbm_reference.txt $SRC/mips.json $SRC/valu_rates.txt $SRC/pytest_gpu.log $SRC/round.log $DST/ 2>/dev/null
ls $DST
