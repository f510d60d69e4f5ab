#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02h; mkdir -p $OUT; ROOT=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -15
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-1500 $OUT/bench.json; tail -3 $OUT/bench.err
for t in BGRX8 RGB8; do timeout 200 python bench.py --format BPTC_FLOAT --target $t --no-cpu --no-extras 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('BPTC_FLOAT -> $t', d['roofline']['launch_us'], d['verified_bit_exact_rows'])"; done
timeout 200 python bench.py --format RGTC1 --target BGRX8 --no-cpu --no-extras 2>>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('RGTC1 -> BGRX8', d['roofline']['launch_us'], d['verified_bit_exact_rows'])"
echo "== done"
