#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02d; mkdir -p $OUT; ROOT=$(pwd)
timeout 300 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "BPTC and not FLOAT" 2>&1 | tail -2
for lib in libdetexhip libdetexhip_exp_waves7 libdetexhip_exp_waves6 libdetexhip_exp_waves1 libdetexhip_exp_nocompute libdetexhip; do
  DETEXHIP_LIB=$ROOT/detex_amd/lib/$lib.so timeout 300 python tools/gpu_time.py BPTC U,C 2>>$OUT/err.log | tee -a $OUT/times.jsonl
done
echo "== done"
