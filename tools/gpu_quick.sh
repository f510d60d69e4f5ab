#!/bin/bash
# quick GPU check: parity tests + per-format launch times (no profiling)
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
timeout 900 python bench.py --steps 80 --no-cpu --formats-json $OUT/formats_8192.json 2> $OUT/formats.err | cut -c1-200
grep "launch_us" $OUT/formats.err
