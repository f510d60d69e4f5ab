#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02e; mkdir -p $OUT; ROOT=$(pwd)
for rep in 1 2; do for g in 12 16 20 24 40 73; do
DETEXHIP_EXP_GRID=$g timeout 300 python tools/gpu_time.py BPTC U,C linear 8192 grid$g 2>>$OUT/err.log | tee -a $OUT/times.jsonl
done; done
echo "== done"
