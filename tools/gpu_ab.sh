#!/bin/bash
# A/B round: parity tests, then launch times of the decoder variants named on the command line
# usage: tools/gpu_ab.sh "<pytest -k expr>" FORMAT:variant ...
set -u
OUT=gpurun_out; mkdir -p $OUT
K="$1"; shift
timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "$K" 2>&1 | tail -6 | tee $OUT/pytest_ab.log
for fv in "$@"; do
  f=${fv%%:*}; v=${fv##*:}
  timeout 300 python bench.py --no-cpu --format $f --variant $v --steps 200 --warmup 400 2>>$OUT/ab.err > $OUT/ab_${f}_v$v.json
  python - <<PY
import json
d=json.load(open("$OUT/ab_${f}_v$v.json"))
print("$f v$v launch_us", d["roofline"].get("launch_us"), "frac", d["roofline"]["frac"], "exact", d.get("verified_bit_exact_rows"))
PY
done
# tiled (block-major) layout of the formats named in $TILED
for f in ${TILED:-}; do
  timeout 300 python bench.py --no-cpu --format $f --layout tiled --steps 200 --warmup 400 2>>$OUT/ab.err > $OUT/ab_${f}_tiled.json
  python - <<PY
import json
d=json.load(open("$OUT/ab_${f}_tiled.json"))
print("$f tiled launch_us", d["roofline"].get("launch_us"), "frac", d["roofline"]["frac"], "exact", d.get("verified_bit_exact_rows"))
PY
done
