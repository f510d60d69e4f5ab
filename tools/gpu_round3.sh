#!/bin/bash
# Round-3 measurement round on the GPU box (via gpurun): what is committed under profiles/r03/ from the FINAL library comes from here
# (the exploratory passes of the round -- tools/gpu_round3_a.sh ... -- left their own files there too).
set -u
export TMPDIR=/tmp
OUT=gpurun_out/round3
rm -rf $OUT; mkdir -p $OUT
ROOT=$(pwd)
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6 | tee $OUT/pytest_gpu.log
echo "== bench (driver line, N=1)"; timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-700 $OUT/bench.json
echo "== per-format table, all formats, streams U / M / C, linear and tiled"
timeout 900 python bench.py --no-cpu --no-extras --formats-json $OUT/formats_8192.json > /dev/null 2> $OUT/formats.err; grep -c launch_us $OUT/formats.err
timeout 900 python bench.py --no-cpu --no-extras --layout tiled --formats-json $OUT/formats_8192_tiled.json > /dev/null 2> $OUT/formats_tiled.err; grep -c launch_us $OUT/formats_tiled.err
echo "== large sizes"
timeout 300 python bench.py --size 16384 --steps 50 --no-cpu --no-extras > $OUT/bench_16384.json 2>> $OUT/bench.err
timeout 300 python bench.py --format BPTC_FLOAT --size 32768 --band-height 4096 --steps 100 --warmup 300 --no-cpu --no-extras > $OUT/bench_bc6h_32768x4096.json 2>>$OUT/bench.err
timeout 300 python bench.py --size 32768 --band-height 8192 --steps 50 --no-cpu --no-extras > $OUT/bench_bc1_32768x8192.json 2>>$OUT/bench.err
for f in bench_16384 bench_bc6h_32768x4096 bench_bc1_32768x8192; do python -c "import json;d=json.load(open('$OUT/$f.json'));print('$f', d['value'], 'Gpixel/s', d['roofline']['launch_us'], 'us', d['roofline']['frac'])"; done
echo "== rocprofv3 kernel trace of the bench command (headline) and of the headline formats (last 200 of 1000 launches)"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -T -d $ROOT/$OUT/prof_trace -o bc1 --output-format csv -- python $ROOT/bench.py --steps 100 --warmup 10 --no-cpu --no-extras > $ROOT/$OUT/prof_trace.log 2>&1
cd $ROOT; f=$(find $OUT/prof_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/bc1_8192_kernel_stats.csv && head -4 "$f" | cut -c1-160
bash tools/gpu_rocprof_formats.sh 2>&1 | grep last200 | cut -c1-200; mkdir -p $OUT/rocprof_formats; cp gpurun_out/rocprof_formats/*.json gpurun_out/rocprof_formats/*kernel_stats.csv $OUT/rocprof_formats/ 2>/dev/null
echo "== PMC traffic (separate passes)"
timeout 1500 python tools/pmc_traffic.py $OUT BC1:linear BC3:linear BPTC:linear BPTC_FLOAT:linear BPTC_SIGNED_FLOAT:linear ETC2_EAC:linear RGTC1:linear BC1:tiled BPTC_FLOAT:tiled BPTC:tiled 2>&1 | tail -12
echo "== clipped geometry / pitches"; timeout 300 python tools/gpu_pitch_sweep.py detex_amd/lib/libdetexhip.so BC1 2>/dev/null | tee $OUT/pitch_bc1.jsonl | cut -c1-140
timeout 300 python tools/gpu_clipped_timing.py 2>/dev/null | tee $OUT/clipped.txt
echo "== fill / copy references"; timeout 600 python tools/gpu_hbm_ref.py $OUT/hbm_reference.jsonl 2>/dev/null | grep -c op
echo "== small calls"; timeout 300 python tools/gpu_small_latency.py detex_amd/lib/libdetexhip.so 2>/dev/null | tee $OUT/small_latency.jsonl | cut -c1-200
echo "== mode histograms / mip chains"; (timeout 300 python tools/bench_histogram.py 2>/dev/null; timeout 300 python tools/bench_histogram.py 4096 2>/dev/null) | tee $OUT/histogram.txt | cut -c1-120; timeout 300 python tools/bench_mips.py 2>/dev/null | tail -1 > $OUT/mips.json; cut -c1-200 $OUT/mips.json
echo "== fuzz 60 s"; timeout 300 python tools/gpu_fuzz.py 60 20000 2>&1 | tail -1 | tee $OUT/fuzz.log
rm -rf $OUT/prof_trace $OUT/pmc_*_*_* 2>/dev/null
echo "== done"
