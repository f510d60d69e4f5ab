#!/bin/bash
# Full measurement round on the GPU box (via gpurun): parity tests, headline bench, per-format table,
# variant A/B, mip-chain and instruction-rate micro-benchmarks, rocprofv3 kernel-trace stats and
# separate PMC passes.  Everything lands in gpurun_out/; tools/save_profiles.sh copies the summaries.
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
ROOT=$(pwd)
echo "== pytest -m gpu"; timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest_gpu.log
echo "== bench (headline)"; timeout 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-330 $OUT/bench.json
echo "== bench variants (1 = north_star 4x4 LDS tile, 2 = cached stores)"; for v in 1 2; do timeout 200 python bench.py --variant $v --no-cpu > $OUT/bench_v$v.json 2>> $OUT/bench.err; python -c "import json;d=json.load(open('$OUT/bench_v$v.json'));print('variant $v', d['roofline']['launch_us'], 'us', d['roofline']['frac'])"; done
echo "== bench 16384 (beyond the 256 MiB Infinity Cache)"; timeout 300 python bench.py --size 16384 --steps 50 --no-cpu > $OUT/bench_16384.json 2>> $OUT/bench.err; python -c "import json;d=json.load(open('$OUT/bench_16384.json'));print(d['value'], d['roofline'])"
echo "== epilogue targets"; for t in BGRA8 RGB8; do timeout 200 python bench.py --no-cpu --target $t > $OUT/bench_bc1_$t.json 2>>$OUT/bench.err; python -c "import json;d=json.load(open('$OUT/bench_bc1_$t.json'));print('$t', d['roofline']['launch_us'], 'us', d['roofline']['frac'], d.get('verified_bit_exact_rows'))"; done
echo "== BC6H A/B: switch scatter (variant 3), cached stores (variant 2)"; for v in 0 2 3; do timeout 200 python bench.py --format BPTC_FLOAT --variant $v --steps 200 --warmup 400 --no-cpu > $OUT/bench_bc6h_v$v.json 2>>$OUT/bench.err; python -c "import json;d=json.load(open('$OUT/bench_bc6h_v$v.json'));print('BPTC_FLOAT variant $v', d['roofline']['launch_us'], 'us', d['roofline']['frac'])"; done
echo "== BC7 A/B: 3 = register-select texel stage, 4 = register field extraction, 5 = mode-sorted waves"; for v in 0 3 4 5; do timeout 200 python bench.py --format BPTC --variant $v --steps 200 --warmup 400 --no-cpu > $OUT/bench_bc7_v$v.json 2>>$OUT/bench.err; python -c "import json;d=json.load(open('$OUT/bench_bc7_v$v.json'));print('BPTC variant $v', d['roofline']['launch_us'], 'us', d['roofline']['frac'], d.get('verified_bit_exact_rows'))"; done
echo "== block-major (tiled) layout"; for f in BC1 BPTC BPTC_FLOAT RGTC2; do timeout 200 python bench.py --format $f --layout tiled --steps 200 --warmup 400 --no-cpu > $OUT/bench_tiled_$f.json 2>>$OUT/bench.err; python -c "import json;d=json.load(open('$OUT/bench_tiled_$f.json'));print('$f tiled', d['roofline']['launch_us'], 'us', d['roofline']['frac'], d.get('verified_bit_exact_rows'))"; done
echo "== 32768-wide bands (the sharded 32768^2 configs: one GPU's band)"; timeout 300 python bench.py --format BPTC_FLOAT --size 32768 --band-height 4096 --steps 100 --warmup 300 --no-cpu > $OUT/bench_bc6h_32768x4096.json 2>>$OUT/bench.err; timeout 300 python bench.py --size 32768 --band-height 8192 --steps 20 --no-cpu > $OUT/bench_bc1_32768x8192.json 2>>$OUT/bench.err; for f in bench_bc6h_32768x4096 bench_bc1_32768x8192; do python -c "import json;d=json.load(open('$OUT/$f.json'));print('$f', d['value'], 'Gpixel/s', d['roofline']['launch_us'], 'us', d['roofline']['frac'])"; done
echo "== per-format table"; timeout 900 python bench.py --steps 80 --no-cpu --formats-json $OUT/formats_8192.json > /dev/null 2> $OUT/formats.err; grep launch_us $OUT/formats.err
echo "== launch time per 25-launch window (power-management transient of VALU-heavy kernels)"; for f in BC1 ETC2 BPTC BPTC_FLOAT; do timeout 120 python tools/gpu_sustain.py $f 32 2>&1 | tail -2; done | tee $OUT/sustain_windows.txt
echo "== mip chains: per-level launches vs one launch"; timeout 300 python tools/bench_mips.py 2>/dev/null | tail -1 > $OUT/mips.json; cat $OUT/mips.json
echo "== HBM reference (plain fill / copy)"; timeout 200 python tools/gpu_hbm_ref.py 2>/dev/null | tee $OUT/hbm_reference.txt
echo "== mode histograms"; timeout 300 python tools/bench_histogram.py 2>/dev/null | tee $OUT/histogram.txt
echo "== VALU instruction rates"; [ -x tools/ubench/valu_rates ] && ./tools/ubench/valu_rates > $OUT/valu_rates.txt 2>&1; cat $OUT/valu_rates.txt
echo "== rocprofv3 kernel trace"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -T -d $ROOT/$OUT/prof_trace -o bc1 --output-format csv -- python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu > $ROOT/$OUT/prof_trace.log 2>&1
cd $ROOT; f=$(find $OUT/prof_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f"
echo "== rocprofv3 PMC passes (separate runs)"
for c in FETCH_SIZE WRITE_SIZE "TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $c | tr ' ' '_')
  cd /tmp && timeout 600 rocprofv3 --pmc $c --kernel-trace -T -d $ROOT/$OUT/prof_pmc_$tag -o bc1 --output-format csv -- python $ROOT/bench.py --steps 10 --warmup 2 --no-cpu > $ROOT/$OUT/prof_pmc_$tag.log 2>&1
  cd $ROOT; f=$(find $OUT/prof_pmc_$tag -name "*counter_collection.csv" | head -1); echo "$c -> $f"; [ -n "$f" ] && (grep decode_linear "$f" | head -2 | cut -d, -f9,16,17)
done
for fmt in BPTC BPTC_FLOAT; do
  echo "== SQ counters, $fmt"
  cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -T -d $ROOT/$OUT/prof_sq_$fmt -o sq --output-format csv -- python $ROOT/bench.py --format $fmt --steps 10 --warmup 2 --no-cpu > $ROOT/$OUT/prof_sq_$fmt.log 2>&1
  cd $ROOT; f=$(find $OUT/prof_sq_$fmt -name "*counter_collection.csv" | head -1); [ -n "$f" ] && grep decode_linear "$f" | head -8 | cut -d, -f9,16,17
done
echo "== done"
