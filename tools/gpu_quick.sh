#!/bin/bash
# quick GPU check: parity tests + per-format launch times + epilogue targets (no profiling)
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
for t in BGRA8 RGB8; do timeout 200 python bench.py --no-cpu --target $t 2>>$OUT/bench.err | tee $OUT/bench_bc1_$t.json | cut -c1-120; python -c "import json;d=json.load(open('$OUT/bench_bc1_$t.json'));print('$t', d['roofline'], d.get('verified_bit_exact_rows'))"; done
timeout 200 python bench.py --no-cpu --format BPTC_FLOAT --target FLOAT_BGRX16 --steps 50 2>>$OUT/bench.err | tee $OUT/bench_bc6h_bgrx16.json | cut -c1-100
timeout 900 python bench.py --steps 80 --no-cpu --formats-json $OUT/formats_8192.json 2> $OUT/formats.err | cut -c1-200
grep "launch_us" $OUT/formats.err
