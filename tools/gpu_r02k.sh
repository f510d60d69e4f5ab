#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02k; mkdir -p $OUT; ROOT=$(pwd)
timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "FLOAT" 2>&1 | tail -2
timeout 300 python tools/gpu_time.py BPTC_SIGNED_FLOAT,BPTC_FLOAT U,M,C 2>>$OUT/err.log | tee -a $OUT/times.jsonl | cut -c1-150
timeout 300 python tools/gpu_time.py BPTC_SIGNED_FLOAT,BPTC_FLOAT U tiled 2>>$OUT/err.log | tee -a $OUT/times.jsonl | cut -c1-150
echo "== done"
