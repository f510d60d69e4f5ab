#!/bin/bash
# round-2 measurement pass A: parity of the new BC7 decoder + persistent tile loop, new bench line, BC7 old-vs-new A/B
set -u
export TMPDIR=/tmp
OUT=gpurun_out/r02a
mkdir -p $OUT
ROOT=$(pwd)
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q -x -p no:cacheprovider 2>&1 | tail -8 | tee $OUT/pytest_gpu.log
echo "== A/B kernels parity (measurement build)"; DETEXHIP_LIB=$ROOT/detex_amd/lib/libdetexhip_ab.so timeout 300 python -m pytest tests/test_ab_variants.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3 | tee $OUT/pytest_ab.log
echo "== bench (driver line)"; timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-2500 $OUT/bench.json; tail -3 $OUT/bench.err
echo "== BC7: round-1 decoder (A/B build, variant 4) vs new, streams U M C"
for s in U M C; do
  DETEXHIP_LIB=$ROOT/detex_amd/lib/libdetexhip_ab.so timeout 200 python bench.py --format BPTC --variant 4 --stream $s --steps 200 --warmup 400 --no-cpu --no-extras > $OUT/bc7_r01_$s.json 2>>$OUT/bench.err
  timeout 200 python bench.py --format BPTC --stream $s --steps 200 --warmup 400 --no-cpu --no-extras > $OUT/bc7_new_$s.json 2>>$OUT/bench.err
  python - <<PY
import json
for t in ("r01", "new"):
    d = json.load(open("$OUT/bc7_%s_$s.json" % t)); print("BPTC stream $s", t, d["roofline"]["launch_us"], "us", d["roofline"]["frac"], d.get("verified_bit_exact_rows"))
PY
done
echo "== per-format table (all formats, U/M/C, steady state)"; timeout 900 python bench.py --no-cpu --no-extras --formats-json $OUT/formats_8192.json > /dev/null 2> $OUT/formats.err; grep launch_us $OUT/formats.err
echo "== SQ counters BPTC"
cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS --kernel-trace -T -d $ROOT/$OUT/prof_sq_BPTC -o sq --output-format csv -- python $ROOT/bench.py --format BPTC --steps 10 --warmup 2 --no-cpu --no-extras > $ROOT/$OUT/prof_sq_BPTC.log 2>&1
cd $ROOT; f=$(find $OUT/prof_sq_BPTC -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep decode_linear "$f" | head -8 | cut -d, -f9,16,17
echo "== done"
