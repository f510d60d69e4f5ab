#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02f; mkdir -p $OUT; ROOT=$(pwd)
FM=RGTC1,SIGNED_RGTC1,EAC_R11,EAC_SIGNED_R11,RGTC2,BC1,BC3,ETC2,ETC2_EAC,BPTC_FLOAT,BPTC_SIGNED_FLOAT
timeout 300 python tools/gpu_time.py $FM U linear 8192 base 2>>$OUT/err.log | tee -a $OUT/times.jsonl
for g in 8 16 32; do
DETEXHIP_LIB=$ROOT/detex_amd/lib/libdetexhip_exp_persistall.so DETEXHIP_EXP_GRID=$g timeout 300 python tools/gpu_time.py $FM U linear 8192 persist$g 2>>$OUT/err.log | tee -a $OUT/times.jsonl
done
timeout 300 python tools/gpu_time.py $FM U linear 8192 base 2>>$OUT/err.log | tee -a $OUT/times.jsonl
echo "== done"
