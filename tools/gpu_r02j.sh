#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02j; mkdir -p $OUT; ROOT=$(pwd)
timeout 300 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "histogram" 2>&1 | tail -2
timeout 200 python tools/bench_histogram.py 2>&1 | tail -8 | cut -c1-140
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -T -d $ROOT/$OUT/prof_hist -o h --output-format csv -- python $ROOT/tools/bench_histogram.py > $ROOT/$OUT/prof_hist.log 2>&1
cd $ROOT; f=$(find $OUT/prof_hist -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -4 "$f" | cut -c1-200
echo "== done"
