#!/bin/bash
# Runs on the GPU box (via gpurun): parity tests, headline bench, per-format table, variant A/B,
# rocprofv3 kernel-trace stats and separate PMC passes.  Everything lands in gpurun_out/.
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
ROOT=$(pwd)
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -15 | tee $OUT/pytest_gpu.log
echo "== bench (headline)"; timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cat $OUT/bench.json
echo "== bench variants (1 = north_star 4x4 LDS tile, 2 = cached stores)"; for v in 1 2; do timeout 200 python bench.py --variant $v --no-cpu > $OUT/bench_v$v.json 2>> $OUT/bench.err; cat $OUT/bench_v$v.json; done
echo "== bench 16384 (beyond the 256 MiB Infinity Cache)"; timeout 300 python bench.py --size 16384 --steps 50 --no-cpu > $OUT/bench_16384.json 2>> $OUT/bench.err; cat $OUT/bench_16384.json
echo "== per-format table"; timeout 900 python bench.py --steps 80 --no-cpu --formats-json $OUT/formats_8192.json > /dev/null 2> $OUT/formats.err; tail -25 $OUT/formats.err
echo "== rocprofv3 kernel trace"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -T -d $ROOT/$OUT/prof_trace -o bc1 --output-format csv -- python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu > $ROOT/$OUT/prof_trace.log 2>&1
cd $ROOT; find $OUT/prof_trace -name "*stats*" | head; f=$(find $OUT/prof_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f"
echo "== rocprofv3 PMC passes (separate runs)"
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -T -d $ROOT/$OUT/prof_pmc_$tag -o bc1 --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu > $ROOT/$OUT/prof_pmc_$tag.log 2>&1
  cd $ROOT; f=$(find $OUT/prof_pmc_$tag -name "*counter_collection.csv" | head -1); echo "$c -> $f"; [ -n "$f" ] && (head -1 "$f"; grep decode_linear "$f" | head -3)
done
echo "== SQ counters, BPTC (VALU-bound kernel)"
cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -T -d $ROOT/$OUT/prof_sq_bptc -o bptc --output-format csv -- python $ROOT/bench.py --format BPTC --steps 10 --warmup 2 --no-cpu > $ROOT/$OUT/prof_sq_bptc.log 2>&1
cd $ROOT; f=$(find $OUT/prof_sq_bptc -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep decode_linear "$f" | head -8 | cut -d, -f9,16,17
echo "== done"
