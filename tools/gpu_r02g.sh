#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02g; mkdir -p $OUT; ROOT=$(pwd)
timeout 900 python -m pytest tests/test_gpu_host_multi.py -m gpu -q -x -p no:cacheprovider 2>&1 | tail -25
timeout 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -4
timeout 300 python bench.py --no-cpu --no-extras 2>$OUT/bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host_tier', d.get('host_tier'), 'value', d['value'])"
echo "== done"
