#!/bin/bash
# Round 3, GPU call C: parity after the multi-device rewrite, misaligned-store fill sweep
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r03d
rm -rf $OUT; mkdir -p $OUT
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest_gpu.log
echo "== clipped geometry"; timeout 300 python tools/gpu_clipped_timing.py 2>&1 | tee $OUT/clipped.txt
echo "== done"
