#!/bin/bash
# Round-2 measurement round on the GPU box (via gpurun): everything that is committed under profiles/r02/ comes from here.
set -u
export TMPDIR=/tmp
OUT=gpurun_out/round2
rm -rf $OUT; mkdir -p $OUT
ROOT=$(pwd)
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest_gpu.log
echo "== bench (driver line, N=1)"; timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-900 $OUT/bench.json
echo "== per-format table, all formats, streams U / M / C, linear and tiled"
timeout 900 python bench.py --no-cpu --no-extras --formats-json $OUT/formats_8192.json > /dev/null 2> $OUT/formats.err; grep -c launch_us $OUT/formats.err
timeout 900 python bench.py --no-cpu --no-extras --layout tiled --formats-json $OUT/formats_8192_tiled.json > /dev/null 2> $OUT/formats_tiled.err; grep -c launch_us $OUT/formats_tiled.err
echo "== 16384^2 (beyond the 256 MiB Infinity Cache) and the 32768-wide bands of the sharded configs"
timeout 300 python bench.py --size 16384 --steps 50 --no-cpu --no-extras > $OUT/bench_16384.json 2>> $OUT/bench.err
timeout 300 python bench.py --format BPTC_FLOAT --size 32768 --band-height 4096 --steps 100 --warmup 300 --no-cpu --no-extras > $OUT/bench_bc6h_32768x4096.json 2>>$OUT/bench.err
timeout 300 python bench.py --size 32768 --band-height 8192 --steps 50 --no-cpu --no-extras > $OUT/bench_bc1_32768x8192.json 2>>$OUT/bench.err
for f in bench_16384 bench_bc6h_32768x4096 bench_bc1_32768x8192; do python -c "import json;d=json.load(open('$OUT/$f.json'));print('$f', d['value'], 'Gpixel/s', d['roofline']['launch_us'], 'us', d['roofline']['frac'])"; done
echo "== small-pixel formats at 16384^2 (their 8192^2 launches are 15-30 us: ramp and tail are a visible share)"
timeout 300 python tools/gpu_time.py RGTC1,SIGNED_RGTC1,RGTC2,EAC_R11,EAC_SIGNED_R11 U linear 16384 2>>$OUT/bench.err | tee $OUT/small_formats_16384.jsonl | cut -c1-140
echo "== epilogue targets"; for spec in BC1:BGRA8 BC1:RGB8 RGTC1:BGRX8 EAC_RG11:RGB8 BPTC_FLOAT:FLOAT_BGRX16 BPTC_FLOAT:BGRX8 BPTC_FLOAT:RGB8; do f=${spec%%:*}; t=${spec##*:}; timeout 200 python bench.py --format $f --target $t --no-cpu --no-extras > $OUT/bench_${f}_$t.json 2>>$OUT/bench.err; python -c "import json;d=json.load(open('$OUT/bench_${f}_$t.json'));print('$f -> $t', d['roofline']['launch_us'], 'us', d['roofline']['frac'], d.get('verified_bit_exact_rows'))"; done
echo "== N=2 code path over gloo on one GPU (plumbing, not a measurement)"
DETEX_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 10 --warmup 3 > $OUT/bench_n2_gloo_one_gpu.json 2>> $OUT/bench.err; cut -c1-300 $OUT/bench_n2_gloo_one_gpu.json
echo "== BC7 / BC6H old-vs-new (A/B build)"
DETEXHIP_LIB=$ROOT/build/explib/libdetexhip_ab.so DETEXHIP_VARIANT=4 timeout 300 python tools/gpu_time.py BPTC U,M,C linear 8192 r01_decoder_variant4 2>>$OUT/bench.err | tee -a $OUT/bc7_ab.jsonl | cut -c1-140
timeout 300 python tools/gpu_time.py BPTC U,M,C linear 8192 r02_decoder 2>>$OUT/bench.err | tee -a $OUT/bc7_ab.jsonl | cut -c1-140
DETEXHIP_LIB=$ROOT/build/explib/libdetexhip_ab.so timeout 300 python -m pytest tests/test_ab_variants.py -m gpu -q -p no:cacheprovider 2>&1 | tail -2 | tee $OUT/pytest_ab.log
echo "== round-1 library (commit 0ebf32e built into build/explib/libdetexhip_r01.so) vs this one, same run"
[ -f build/explib/libdetexhip_r01.so ] && bash tools/gpu_cmp_r01.sh 2>&1 | tail -14 | tee $OUT/r01_vs_r02_same_run.txt
echo "== decode without stores / stores without decode (measurement builds)"
for lib in libdetexhip libdetexhip_exp_nostore libdetexhip_exp_nocompute; do DETEXHIP_LIB=$ROOT/detex_amd/lib/$lib.so timeout 300 python tools/gpu_time.py BPTC,BPTC_SIGNED_FLOAT,BPTC_FLOAT,ETC2_EAC,RGTC1,BC3,BC1 U 2>>$OUT/bench.err | tee -a $OUT/compute_vs_memory.jsonl | cut -c1-130; done
echo "== mode histograms (4 Mi and 16 Mi blocks)"; (timeout 300 python tools/bench_histogram.py 2>/dev/null; timeout 300 python tools/bench_histogram.py 4096 2>/dev/null) | tee $OUT/histogram.txt | cut -c1-120
echo "== VALU issue rates"; timeout 120 ./tools/ubench/valu_rates > $OUT/valu_rates.txt 2>&1; tail -12 $OUT/valu_rates.txt
echo "== mip chains"; timeout 300 python tools/bench_mips.py 2>/dev/null | tail -1 > $OUT/mips.json; cut -c1-300 $OUT/mips.json
echo "== host transfer paths"; timeout 120 ./tools/ubench/host_paths 2>&1 | tee $OUT/host_paths.txt | head -8
echo "== launch time per 25-launch window + clocks (power-management transient)"; for f in BC1 BPTC BPTC_SIGNED_FLOAT; do timeout 120 python tools/gpu_sustain.py $f 32 2>&1 | tail -2 | cut -c1-900; done | tee $OUT/sustain_windows.txt
echo "== rocprofv3 kernel trace of the bench command"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -T -d $ROOT/$OUT/prof_trace -o bc1 --output-format csv -- python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu --no-extras > $ROOT/$OUT/prof_trace.log 2>&1
cd $ROOT; f=$(find $OUT/prof_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/bc1_8192_kernel_stats.csv && head -4 "$f" | cut -c1-160
echo "== PMC traffic (separate passes)"
timeout 1500 python tools/pmc_traffic.py $OUT BC1:linear BC3:linear BPTC:linear BPTC_FLOAT:linear BPTC_SIGNED_FLOAT:linear ETC2_EAC:linear RGTC1:linear BC1:tiled BPTC_FLOAT:tiled BPTC:tiled 2>&1 | tail -12
echo "== SQ counters"
for fmt in BPTC BPTC_SIGNED_FLOAT BPTC_FLOAT; do
  cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --kernel-trace -T -d $ROOT/$OUT/prof_sq_$fmt -o sq --output-format csv -- python $ROOT/bench.py --format $fmt --steps 10 --warmup 2 --no-cpu --no-extras > $ROOT/$OUT/prof_sq_$fmt.log 2>&1
  cd $ROOT; f=$(find $OUT/prof_sq_$fmt -name "*counter_collection.csv" | head -1); [ -n "$f" ] && (head -1 "$f"; grep decode_linear "$f" | head -16) > $OUT/$(echo $fmt | tr A-Z a-z)_8192_pmc_SQ.csv && grep decode_linear "$f" | head -8 | cut -d, -f9,16,17
  cd /tmp && timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -T -d $ROOT/$OUT/prof_lds_$fmt -o sq --output-format csv -- python $ROOT/bench.py --format $fmt --steps 10 --warmup 2 --no-cpu --no-extras > $ROOT/$OUT/prof_lds_$fmt.log 2>&1
  cd $ROOT; f=$(find $OUT/prof_lds_$fmt -name "*counter_collection.csv" | head -1); [ -n "$f" ] && (head -1 "$f"; grep decode_linear "$f" | head -10) > $OUT/$(echo $fmt | tr A-Z a-z)_8192_pmc_LDS.csv
done
rm -rf $OUT/prof_* $OUT/pmc_*_*_* 2>/dev/null
echo "== done"
