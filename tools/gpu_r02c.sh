#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02c; mkdir -p $OUT; ROOT=$(pwd)
for lib in libdetexhip libdetexhip_exp_nonpersistent libdetexhip_exp_plain libdetexhip_exp_plain_nonpersistent libdetexhip_exp_nocompute; do
  DETEXHIP_LIB=$ROOT/detex_amd/lib/$lib.so timeout 300 python tools/gpu_time.py BPTC U,C 2>>$OUT/err.log | tee -a $OUT/times.jsonl
done
DETEXHIP_LIB=$ROOT/detex_amd/lib/libdetexhip_exp_nocompute.so timeout 300 python tools/gpu_time.py BC3,BC1 U 2>>$OUT/err.log | tee -a $OUT/times.jsonl
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*" | sort -u | tr '\n' ' ' | head -c 6000 > $OUT/sq_counters.txt
echo "== done"
